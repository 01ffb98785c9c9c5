#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT 'command'   - retries while no GPU slot / box is free (exit code 3: nothing charged)
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gpurun_last.txt 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.txt || [ $rc -eq 3 ]; then sleep 45; continue; fi
  break
done
tail -${3:-100} /tmp/gpurun_last.txt
exit $rc

#!/bin/bash
# Idle time between consecutive kernels of the control loops (tools/kernel_gaps.py over rocprofv3 kernel traces), with the launch
# tape and with the hipGraph replay it replaces -> gpurun_out/r04_kernel_gaps.txt
OUT=gpurun_out/r04_kernel_gaps.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
: > $OUT
for c in "" "--controller cem --particles 4096" "--controller cem --particles 16384" "--particles 16384" "--workload cartpole" "--workload half_cheetah" "--noise mt19937"; do
  for t in "" "--no-tape"; do
    [ -z "$c" ] && [ -n "$t" ] && continue          # (the headline launches its two kernels directly either way)
    rm -rf gpurun_out/kt
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o t -- python3 bench.py $c $t --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/kt_line.json 2>/dev/null
    echo "== bench.py $c $t" >> $OUT
    python3 -c "import json; j=json.loads(open('gpurun_out/kt_line.json').read().strip().splitlines()[-1]); print('   ms_per_step (under rocprofv3)', round(j['ms_per_step'], 4), '|', j['config']['launch'])" >> $OUT
    python3 tools/kernel_gaps.py gpurun_out/kt | head -8 >> $OUT
  done
done
rm -rf gpurun_out/kt gpurun_out/kt_line.json
cat $OUT

#!/bin/bash
# usage: ab_bench.sh out variant...   (bench.py closed-loop ms per step of tree workloads)
out=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = product ]; then unset MJMPC_AMD_LIB; else export MJMPC_AMD_LIB=$PWD/tools/_build/libmjmpc_amd_$v.so; fi
  for WL in half_cheetah swimmer tray door gripper; do
    echo -n "$v $WL: " >> gpurun_out/$out.txt
    python3 bench.py --workload $WL --steps 30 --warmup 5 --process-warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],4), round(j['roofline']['kernel_ms'],4))" >> gpurun_out/$out.txt
  done
done; done
unset MJMPC_AMD_LIB
cat gpurun_out/$out.txt

#!/bin/bash
# A/B of tree-kernel build variants (tools/_build/libmjmpc_amd_<name>.so, built with mjmpc_amd.build.build(extra_flags=..., lib=...))
# on one box:  bash tools/ab_tree.sh <out> <variant> [<variant> ...]     ("product" = the in-tree library)
out=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = product ]; then unset MJMPC_AMD_LIB; else export MJMPC_AMD_LIB=$PWD/tools/_build/libmjmpc_amd_$v.so; fi
  for cfg in "4096 32 f64 cheetah" "4096 32 f64 swimmer" "4096 32 f64 hand" "4096 32 f64 pen" "4096 32 f64 tray" "4096 32 f64 door" "4096 32 f64 cartpole" "32768 32 f32 cheetah" "4096 32 f32 pen"; do
    echo -n "$v: " >> gpurun_out/$out.txt
    timeout 300 python tools/tree_time.py $cfg 2>&1 | tail -1 >> gpurun_out/$out.txt
  done
done
done
unset MJMPC_AMD_LIB
cat gpurun_out/$out.txt

#!/bin/bash
# GPU box: parity of the flag-synchronised launches at the baseline sizes, then timings of the shapes (developer tool, round 6)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_baseline_sizes_gpu.py -x -q -k "rollout_matches or minimum_slice" 2>&1 | tail -5
for F in -1 0 2; do echo MJMPC_ARM_FLAGS=$F; for P in 512 1024 2048 4096; do if [ $F = -1 ]; then unset MJMPC_ARM_FLAGS; else export MJMPC_ARM_FLAGS=$F; fi; timeout 120 python tools/mono_time.py $P 32 f64 2>&1 | grep -e fused -e failures | tr '\n' ' '; echo " P=$P"; done; done
unset MJMPC_ARM_FLAGS
for P in 1024 4096; do STAMPS_LIB=tools/_build/libmjmpc_stampsf.so timeout 120 python tools/stamps.py $P f64 2>&1 | grep -v -e Warn -e amdgpu.ids | tee gpurun_out/stamps_flags_$P.txt; done

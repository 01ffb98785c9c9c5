python -m pytest tests/test_tree_rollout_gpu.py tests/test_pen_hand_gpu.py tests/test_stress_parity_gpu.py tests/test_locomotion_gpu.py -x -q 2>&1 | tail -3
for v in tools/_build/libhead.so ""; do export MJMPC_AMD_LIB=$v; [ -z "$v" ] && unset MJMPC_AMD_LIB; python tools/_pen_time.py 2>&1 | grep -v amdgpu | tail -1; done
unset MJMPC_AMD_LIB
python tools/tree_time.py 65536 64 f64 hand 2>&1 | grep -v amdgpu | tail -1
python tools/tree_time.py 65536 64 f32 hand 2>&1 | grep -v amdgpu | tail -1
python tools/tree_time.py 4096 32 f64 cheetah 2>&1 | grep -v amdgpu | tail -1

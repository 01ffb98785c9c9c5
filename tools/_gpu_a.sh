python -m pytest tests/test_tree_rollout_gpu.py -x -q 2>&1 | tail -2
python tools/tree_time.py 65536 64 f64 hand 2>&1 | grep -v amdgpu | tail -1
python tools/tree_time.py 65536 64 f32 hand 2>&1 | grep -v amdgpu | tail -1
python tools/tree_time.py 4096 32 f64 hand 2>&1 | grep -v amdgpu | tail -1

python -m pytest tests/test_tree_rollout_gpu.py tests/test_pen_hand_gpu.py tests/test_stress_parity_gpu.py -x -q 2>&1 | tail -3
python tools/tree_time.py 65536 64 f64 hand 2>&1 | grep -v amdgpu | tail -1
python tools/tree_time.py 65536 64 f32 hand 2>&1 | grep -v amdgpu | tail -1
python bench.py --workload pen_hand --steps 5 --warmup 2 --no-cpu-baseline --process-warmup 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pen mppi 4096x32', j['ms_per_step'], j['roofline']['kernel_ms'], j['solver_failures'])"

python -m pytest tests/test_tree_rollout_gpu.py tests/test_pen_hand_gpu.py -x -q 2>&1 | tail -5
python tools/tree_time.py 65536 64 f64 hand 2>&1 | grep -v amdgpu | tail -1
python tools/tree_time.py 65536 64 f32 hand 2>&1 | grep -v amdgpu | tail -1
python bench.py --workload pen_hand --controller dmd --particles 65536 --horizon 64 --steps 3 --warmup 1 --no-cpu-baseline --process-warmup 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pen dmd 65536x64', j['ms_per_step'], j['roofline']['kernel_ms'], j['solver_failures'])"

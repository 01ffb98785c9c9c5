TREE_STATS_LIB=tools/_build/libws_stats.so python tools/tree_stats.py cheetah f64 4096 32 2>&1 | grep -v amdgpu | grep "iterations\|total"
python tools/tree_time.py 4096 32 f64 cheetah 2>&1 | grep -v amdgpu | tail -1
MJMPC_AMD_LIB=tools/_build/libws.so python tools/tree_time.py 4096 32 f64 cheetah 2>&1 | grep -v amdgpu | tail -1
MJMPC_AMD_LIB=tools/_build/libws.so python -m pytest tests/test_locomotion_gpu.py -x -q -k "f64 or cheetah_mppi" 2>&1 | tail -2
MJMPC_AMD_LIB=tools/_build/libws.so python bench.py --workload pen_hand --steps 4 --warmup 1 --no-cpu-baseline --process-warmup 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pen ws', j['ms_per_step'], j['solver_failures'])"
python bench.py --workload pen_hand --steps 4 --warmup 1 --no-cpu-baseline --process-warmup 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pen base', j['ms_per_step'], j['solver_failures'])"

python tools/bench_configs.py --steps 60 --pen > gpurun_out/r03_other_configs_f64.jsonl 2> gpurun_out/r03_cfg.err
python tools/bench_configs.py --steps 60 --dtype f32 > gpurun_out/r03_other_configs_f32.jsonl 2>> gpurun_out/r03_cfg.err
for wl in half_cheetah swimmer hand24 pen_hand; do python bench.py --workload $wl --steps 10 --warmup 3 2>/dev/null | tail -1; done > gpurun_out/r03_tree_bench_lines.jsonl
for a in "4096 32 f64 cheetah" "4096 32 f32 cheetah" "32768 32 f64 cheetah" "32768 32 f32 cheetah" "4096 32 f64 swimmer" "65536 64 f64 hand" "65536 64 f32 hand"; do python tools/tree_time.py $a 2>&1 | grep -v amdgpu | tail -1; done > gpurun_out/r03_tree_time.txt
python tools/tree_stats.py cheetah f64 4096 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_tree_stats.txt
python tools/tree_stats.py hand f64 4096 32 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_tree_stats.txt
python tools/tree_stats.py pen f64 4096 32 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_tree_stats.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_f64_line.json 2>/dev/null
cat gpurun_out/r03_tree_time.txt; grep "per Newton\|total" gpurun_out/r03_tree_stats.txt

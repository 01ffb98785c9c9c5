bash tools/profile_round3.sh r03 > gpurun_out/r03_profile.log 2>&1
python tools/bench_configs.py --steps 60 --pen > gpurun_out/r03_other_configs_f64.jsonl 2> gpurun_out/r03_cfg.err
python tools/bench_configs.py --steps 60 --dtype f32 > gpurun_out/r03_other_configs_f32.jsonl 2>> gpurun_out/r03_cfg.err
for wl in half_cheetah swimmer hand24 pen_hand; do python bench.py --workload $wl --steps 10 --warmup 3 2>/dev/null | tail -1; done > gpurun_out/r03_tree_bench_lines.jsonl
tail -2 gpurun_out/r03_profile.log

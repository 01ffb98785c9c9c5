for w in reacher half_cheetah swimmer hand24 pen_hand cartpole tray door; do
  for dt in f64 f32; do
    timeout 600 python bench.py --workload $w --dtype $dt --steps 400 --warmup 5 --process-warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w $dt', j['ms_per_step'], 'fail', j.get('solver_failures'), 'div', j.get('diverged_particle_substeps'), 'dist', j.get('final_distance_to_target'))" >> gpurun_out/r4_soak.txt 2>&1
  done
done
MJMPC_FUZZ_SEEDS=2000:2500 timeout 1500 python -m pytest tests/test_random_models_gpu.py -q -x 2>&1 | tail -5 >> gpurun_out/r4_soak.txt
for w in half_cheetah cartpole swimmer; do for c in cem dmd; do
  timeout 600 python bench.py --workload $w --controller $c --steps 400 --warmup 5 --process-warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w $c', j['ms_per_step'], 'fail', j.get('solver_failures'), 'div', j.get('diverged_particle_substeps'), j['config']['launch'])" >> gpurun_out/r4_soak.txt 2>&1
done; done
cat gpurun_out/r4_soak.txt

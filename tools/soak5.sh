# Round-5 soak on the GPU box: long closed loops on every workload (iteration-cap hits, resets), and the random-model
# generator over seeds beyond the suite's (new colliders, cones, margin / ref / gap).  -> gpurun_out/r05_soak.txt
O=gpurun_out/r05_soak.txt
: > $O
for w in reacher half_cheetah swimmer hand24 pen_hand cartpole tray door gripper; do
  for dt in f64 f32; do
    timeout 600 python bench.py --workload $w --dtype $dt --steps 400 --warmup 5 --process-warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w $dt', round(j['ms_per_step'], 4), 'ms/step  cap hits', j.get('solver_failures'), ' resets', j.get('diverged_particle_substeps'), ' dist', j.get('final_distance_to_target'))" >> $O 2>&1
  done
done
for r in ${SOAK_SEEDS:-5000:5300 5300:5600}; do
  echo "MJMPC_FUZZ_SEEDS=$r tests/test_random_models_gpu.py:" >> $O
  MJMPC_FUZZ_SEEDS=$r timeout 2400 python -m pytest tests/test_random_models_gpu.py -q 2>&1 | tail -6 >> $O
done
cat $O

# Round-6 soak after the solver's control-flow changes (per-particle decisions, residual reuse, Illinois line search, earlier
# search with friction-loss rows): fresh random-model seeds with bookkeeping, then 400 closed-loop control steps of every
# bench workload in f64 and f32 (iteration-cap hits, diverged substeps, env resets).  -> gpurun_out/r06_soak_solver.txt
O=gpurun_out/r06_soak_solver.txt
S=gpurun_out/fuzz_stats2.jsonl
: > $O; : > $S
for r in ${SOAK_SEEDS:-80000:82000 82000:84000 84000:86000}; do
  echo "MJMPC_FUZZ_SEEDS=$r tests/test_random_models_gpu.py::test_random_model_matches_oracle:" >> $O
  MJMPC_FUZZ_STATS=$S MJMPC_FUZZ_SEEDS=$r timeout 2400 python -m pytest tests/test_random_models_gpu.py -q -k matches_oracle 2>&1 | tail -4 >> $O
done
python tools/soak_summary.py $S >> $O
for w in half_cheetah swimmer hand24 pen_hand tray door gripper; do
  for dt in f64 f32; do
    timeout 900 python bench.py --workload $w --dtype $dt --steps 400 --warmup 5 --process-warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']; print('$w $dt ms/step', round(j['ms_per_step'],4), {k:v for k,v in list(j.items())+list(c.items()) if any(t in k for t in ('fail','diverg','reset','cap'))})" >> $O 2>&1
  done
done
cat $O

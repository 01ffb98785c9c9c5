# Round-6 soak on the GPU box: the random-model generator over seeds beyond the suite's WITH bookkeeping (re-draws, refusals,
# scaled-tolerance hits, non-finite oracle rollouts per 1000 seeds).  -> gpurun_out/r06_soak_bookkeeping.txt
O=gpurun_out/r06_soak_bookkeeping.txt
S=gpurun_out/fuzz_stats.jsonl
: > $O; : > $S
for r in ${SOAK_SEEDS:-70000:70500 70500:71000}; do
  echo "MJMPC_FUZZ_SEEDS=$r tests/test_random_models_gpu.py::test_random_model_matches_oracle:" >> $O
  MJMPC_FUZZ_STATS=$S MJMPC_FUZZ_SEEDS=$r timeout 2400 python -m pytest tests/test_random_models_gpu.py -q -k matches_oracle 2>&1 | tail -4 >> $O
done
python tools/soak_summary.py $S >> $O
cat $O

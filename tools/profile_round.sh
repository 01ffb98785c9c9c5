#!/bin/bash
# Round profile on the GPU box:  bash tools/profile_round.sh r02      (outputs under gpurun_out/<tag>_*)
# 1. rocprofv3 --kernel-trace --stats of exactly the bench command (f64 and f32)
# 2. PMC passes over tools/pmc_run.py 4096 f64: two SQ passes, FETCH_SIZE, WRITE_SIZE (separate runs, kernel-trace only)
# 3. tools/pmc_summarize.py turns them into profiles-ready JSON (traffic per launch, VALU issue fraction)
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for DT in f64 f32; do
  python3 bench.py --steps 200 --warmup 20 --dtype $DT $([ $DT = f32 ] && echo --no-cpu-baseline) > $OUT/${TAG}_bench_${DT}_line.json 2> $OUT/${TAG}_bench_${DT}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$DT -o ${TAG}_$DT -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype $DT > $OUT/${TAG}_bench_${DT}_line_under_rocprof.json 2> $OUT/${TAG}_prof_${DT}.err
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_pmcS1 -o s1 -- python3 tools/pmc_run.py 4096 f64 > $OUT/${TAG}_pmcS1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_pmcS2 -o s2 -- python3 tools/pmc_run.py 4096 f64 > $OUT/${TAG}_pmcS2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmcF -o f -- python3 tools/pmc_run.py 4096 f64 > $OUT/${TAG}_pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmcW -o w -- python3 tools/pmc_run.py 4096 f64 > $OUT/${TAG}_pmcW.log 2>&1
python3 tools/pmc_summarize.py $TAG $OUT > $OUT/${TAG}_pmc_summary.txt 2>&1
# 4. the other configurations (arm: cfg1 / cfg3 / cfg4*, tree engine: locomotion + the 24-dof hand), and the tree kernel under rocprof
for DT in f64 f32; do
  python3 tools/bench_configs.py --steps 60 --dtype $DT > $OUT/${TAG}_other_configs_$DT.jsonl 2> $OUT/${TAG}_other_configs_$DT.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_tree -o ${TAG}_tree -- python3 tools/bench_configs.py --only-tree --dtype f64 > $OUT/${TAG}_tree_under_rocprof.jsonl 2> $OUT/${TAG}_prof_tree.err
find $OUT -name "*.csv" -size +20M -delete
ls $OUT | grep $TAG

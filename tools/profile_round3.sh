#!/bin/bash
# Round-3 profiles on the GPU box:  bash tools/profile_round3.sh [tag]      (outputs under gpurun_out/<tag>_*)
# 1. rocprofv3 --kernel-trace --stats of exactly the bench command (f64)
# 2. PMC passes (each its own run, kernel-trace only) over tools/pmc_run.py for the reacher and the cheetah:
#    SQ occupancy / issue counters (two passes), FP64 instruction mix (one pass), FETCH_SIZE, WRITE_SIZE
# 3. tools/pmc_summarize.py turns them into the JSON figures bench.py quotes
TAG=${1:-r03}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_f64_line.json 2> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_f64_200_line.json 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lookahead > $OUT/${TAG}_bench_f64_200_lookahead_line.json 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-mono > $OUT/${TAG}_bench_f64_200_nomono_line.json 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --dtype f32 > $OUT/${TAG}_bench_f32_200_line.json 2>> $OUT/${TAG}_bench_f64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_f64 -o ${TAG}_f64 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_f64_line_under_rocprof.json 2> $OUT/${TAG}_prof_f64.err
rocprofv3 --list-avail > $OUT/${TAG}_counters_avail.txt 2>&1
for WL in reacher half_cheetah; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_${WL}_pmcS1 -o s1 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_${WL}_pmcS2 -o s2 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/${TAG}_${WL}_pmcS3 -o s3 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS3.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/${TAG}_${WL}_pmcS4 -o s4 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS4.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${WL}_pmcF -o f -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcF.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${WL}_pmcW -o w -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcW.log 2>&1
done
python3 tools/pmc_summarize.py $TAG $OUT > $OUT/${TAG}_pmc_summary.txt 2>&1
# keep the merged artefacts small: per-pass CSVs are copied next to the summaries, the trace directories dropped
for WL in reacher half_cheetah; do for p in S1 S2 S3 S4 F W; do
  f=$(find $OUT/${TAG}_${WL}_pmc$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${WL}_pmc_${p}_counter_collection.csv
done; done
f=$(find $OUT/${TAG}_prof_f64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_bench_f64_kernel_stats.csv
rm -rf $OUT/${TAG}_*_pmcS1 $OUT/${TAG}_*_pmcS2 $OUT/${TAG}_*_pmcS3 $OUT/${TAG}_*_pmcS4 $OUT/${TAG}_*_pmcF $OUT/${TAG}_*_pmcW $OUT/${TAG}_prof_f64
ls $OUT | grep $TAG

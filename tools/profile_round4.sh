#!/bin/bash
# Round-4 profiles on the GPU box:  bash tools/profile_round4.sh [tag]      (outputs under gpurun_out/<tag>_*)
# 1. the bench lines (driver's command, 200 steps, lookahead, f32, 1024 / 512 particles, the other workloads and controllers)
# 2. rocprofv3 --kernel-trace --stats of the bench command (f64), of the CEM configuration and of the tree workloads
# 3. PMC passes (each its own run, kernel-trace only) over tools/pmc_run.py for the reacher: SQ occupancy / issue counters,
#    FP64 instruction mix, FETCH_SIZE, WRITE_SIZE; tools/pmc_summarize.py turns them into the figures bench.py quotes
TAG=${1:-r04}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_f64_line.json 2> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_f64_200_line.json 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lookahead > $OUT/${TAG}_bench_f64_200_lookahead_line.json 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --dtype f32 > $OUT/${TAG}_bench_f32_200_line.json 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --noise mt19937 > $OUT/${TAG}_bench_f64_mt19937_line.json 2>> $OUT/${TAG}_bench_f64.err
: > $OUT/${TAG}_sweep.jsonl
for P in 512 1024 4096 8192 16384 65536; do
  python3 bench.py --particles $P --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_sweep.jsonl 2>> $OUT/${TAG}_bench_f64.err
done
: > $OUT/${TAG}_workloads.jsonl
for WL in half_cheetah swimmer hand24 pen_hand cartpole tray door; do
  python3 bench.py --workload $WL --steps 30 --warmup 5 --process-warmup 10 --cpu-seconds 4 >> $OUT/${TAG}_workloads.jsonl 2>> $OUT/${TAG}_bench_f64.err
done
: > $OUT/${TAG}_controllers.jsonl
python3 bench.py --controller cem --particles 16384 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --controller cem --particles 1024 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --controller cem --particles 4096 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $OUT/${TAG}_bench_f64.err
python3 bench.py --controller dmd --particles 4096 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $OUT/${TAG}_bench_f64.err
python3 tools/bench_configs.py --steps 40 --pen > $OUT/${TAG}_other_configs_f64.jsonl 2>> $OUT/${TAG}_bench_f64.err
python3 tools/bench_configs.py --steps 40 --pen --dtype f32 > $OUT/${TAG}_other_configs_f32.jsonl 2>> $OUT/${TAG}_bench_f64.err
# tree kernel: launch times from a fixed start state (tools/tree_time.py; MJMPC_TREE_SPARSE=1: the tree-sparse factorisation
# where the dense 32-lane one is the default) and, when the instrumented library is there, its phase clocks
: > $OUT/${TAG}_tree_time.txt
for c in "4096 32 f64 cheetah" "4096 32 f64 swimmer" "4096 32 f64 hand" "4096 32 f64 handf" "4096 32 f64 pen" "4096 32 f32 pen" "4096 32 f64 tray" "4096 32 f64 door" "4096 32 f64 cartpole" "65536 16 f64 hand"; do
  python3 tools/tree_time.py $c 2>/dev/null | tail -1 >> $OUT/${TAG}_tree_time.txt
done
for c in "4096 32 f64 pen" "4096 32 f32 pen" "4096 32 f64 handf"; do
  echo -n "MJMPC_TREE_SPARSE=1: " >> $OUT/${TAG}_tree_time.txt
  MJMPC_TREE_SPARSE=1 python3 tools/tree_time.py $c 2>/dev/null | tail -1 >> $OUT/${TAG}_tree_time.txt
done
if [ -f tools/_build/libmjmpc_amd_treestats.so ]; then
  : > $OUT/${TAG}_tree_stats.txt
  for m in cheetah swimmer hand pen; do python3 tools/tree_stats.py $m f64 4096 32 2>/dev/null | tail -11 >> $OUT/${TAG}_tree_stats.txt; done
  python3 tools/tree_stats.py pen f32 4096 32 2>/dev/null | tail -11 >> $OUT/${TAG}_tree_stats.txt
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_pen -o ${TAG}_pen -- python3 bench.py --workload pen_hand --controller dmd --particles 65536 --horizon 64 --steps 4 --warmup 1 --process-warmup 0 --no-cpu-baseline > $OUT/${TAG}_pen_65536x64_line_under_rocprof.json 2> $OUT/${TAG}_prof_pen.err
f=$(find $OUT/${TAG}_prof_pen -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_pen_65536x64_kernel_stats.csv
rm -rf $OUT/${TAG}_prof_pen
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_f64 -o ${TAG}_f64 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_f64_line_under_rocprof.json 2> $OUT/${TAG}_prof_f64.err
f=$(find $OUT/${TAG}_prof_f64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_bench_f64_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_cem -o ${TAG}_cem -- python3 bench.py --controller cem --particles 16384 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_cem_line_under_rocprof.json 2> $OUT/${TAG}_prof_cem.err
f=$(find $OUT/${TAG}_prof_cem -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cem_16384_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_tree -o ${TAG}_tree -- python3 bench.py --workload tray --steps 30 --warmup 5 --process-warmup 5 --no-cpu-baseline > $OUT/${TAG}_tray_line_under_rocprof.json 2> $OUT/${TAG}_prof_tree.err
f=$(find $OUT/${TAG}_prof_tree -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_tray_kernel_stats.csv
for WL in reacher half_cheetah; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_${WL}_pmcS1 -o s1 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_${WL}_pmcS2 -o s2 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/${TAG}_${WL}_pmcS3 -o s3 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS3.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/${TAG}_${WL}_pmcS4 -o s4 -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcS4.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${WL}_pmcF -o f -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcF.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${WL}_pmcW -o w -- python3 tools/pmc_run.py $WL 4096 f64 > $OUT/${TAG}_${WL}_pmcW.log 2>&1
done
python3 tools/pmc_summarize.py $TAG $OUT > $OUT/${TAG}_pmc_summary.txt 2>&1
for WL in reacher half_cheetah; do for p in S1 S2 S3 S4 F W; do
  f=$(find $OUT/${TAG}_${WL}_pmc$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${WL}_pmc_${p}_counter_collection.csv
done; done
rm -rf $OUT/${TAG}_*_pmcS1 $OUT/${TAG}_*_pmcS2 $OUT/${TAG}_*_pmcS3 $OUT/${TAG}_*_pmcS4 $OUT/${TAG}_*_pmcF $OUT/${TAG}_*_pmcW $OUT/${TAG}_prof_f64 $OUT/${TAG}_prof_cem $OUT/${TAG}_prof_tree
[ "$TAG" = r04 ] && bash tools/profile_cem.sh > /dev/null 2>&1      # the CEM lines, kernel stats at 16384 and 4096, selection phases
ls $OUT | grep $TAG

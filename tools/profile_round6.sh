#!/bin/bash
# Round-6 profiles on the GPU box:  bash tools/profile_round6.sh PART [tag]      (outputs under gpurun_out/<tag>_*)
#   PART a: the bench lines (driver's command, 200 steps, lookahead, f32, the particle sweep, the other workloads and controllers,
#           BASELINE's other configs) and rocprofv3 --kernel-trace --stats of the bench command, pen-in-hand at 65536 x 64, tray, gripper
#   PART b: PMC passes (each its own run, kernel-trace only) over tools/pmc_run.py - the reacher at 4096 / 16384 / 65536 particles
#           (the one-wave, two-wave and throughput instantiations), HalfCheetah and the general instantiation's models (cart-pole,
#           door, tray, gripper): SQ occupancy / issue counters, FP64 + FP32 instruction mix, FETCH_SIZE, WRITE_SIZE;
#           tools/pmc_summarize.py turns them into the figures bench.py quotes; then launch times and phase clocks of the tree kernel
#   PART c: idle time between kernels - the sharded iteration with the library's all-gather and with torch's, the one-GPU loops
PART=${1:-a}
TAG=${2:-r06}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ERR=$OUT/${TAG}_bench.err
if [ "$PART" = a ]; then
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_f64_line.json 2> $ERR
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_f64_200_line.json 2>> $ERR
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lookahead > $OUT/${TAG}_bench_f64_200_lookahead_line.json 2>> $ERR
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --dtype f32 > $OUT/${TAG}_bench_f32_200_line.json 2>> $ERR
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --noise mt19937 > $OUT/${TAG}_bench_f64_mt19937_line.json 2>> $ERR
: > $OUT/${TAG}_sweep.jsonl
for P in 512 1024 2048 4096 8192 16384 65536; do
  python3 bench.py --particles $P --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_sweep.jsonl 2>> $ERR
done
: > $OUT/${TAG}_workloads.jsonl
for WL in half_cheetah swimmer hand24 pen_hand cartpole tray door gripper; do
  python3 bench.py --workload $WL --steps 30 --warmup 5 --process-warmup 10 --cpu-seconds 4 >> $OUT/${TAG}_workloads.jsonl 2>> $ERR
done
: > $OUT/${TAG}_controllers.jsonl
python3 bench.py --controller cem --particles 16384 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $ERR
python3 bench.py --controller cem --particles 4096 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $ERR
python3 bench.py --controller dmd --particles 4096 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_controllers.jsonl 2>> $ERR
python3 tools/bench_configs.py --steps 40 --pen > $OUT/${TAG}_other_configs_f64.jsonl 2>> $ERR
python3 tools/bench_configs.py --steps 40 --pen --dtype f32 > $OUT/${TAG}_other_configs_f32.jsonl 2>> $ERR
prof() {    # name, then the bench arguments
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$n -o ${TAG}_$n -- python3 bench.py "$@" --no-cpu-baseline > $OUT/${TAG}_${n}_line_under_rocprof.json 2> $OUT/${TAG}_prof_$n.err
  f=$(find $OUT/${TAG}_prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${n}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_$n
}
prof bench_f64 --steps 100 --warmup 10
prof pen_65536x64 --workload pen_hand --controller dmd --particles 65536 --horizon 64 --steps 4 --warmup 1 --process-warmup 0
prof tray --workload tray --steps 30 --warmup 5 --process-warmup 5
prof gripper --workload gripper --steps 30 --warmup 5 --process-warmup 5
prof door --workload door --steps 30 --warmup 5 --process-warmup 5
prof cartpole --workload cartpole --steps 30 --warmup 5 --process-warmup 5
fi
if [ "$PART" = b ]; then
pmc() {     # directory name, workload, particles
  d=$1; wl=$2; p=$3
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_${d}_pmcS1 -o s1 -- python3 tools/pmc_run.py $wl $p f64 > $OUT/${TAG}_${d}_pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_${d}_pmcS2 -o s2 -- python3 tools/pmc_run.py $wl $p f64 >> $OUT/${TAG}_${d}_pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/${TAG}_${d}_pmcS3 -o s3 -- python3 tools/pmc_run.py $wl $p f64 >> $OUT/${TAG}_${d}_pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/${TAG}_${d}_pmcS4 -o s4 -- python3 tools/pmc_run.py $wl $p f64 >> $OUT/${TAG}_${d}_pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${d}_pmcF -o f -- python3 tools/pmc_run.py $wl $p f64 >> $OUT/${TAG}_${d}_pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${d}_pmcW -o w -- python3 tools/pmc_run.py $wl $p f64 >> $OUT/${TAG}_${d}_pmc.log 2>&1
}
pmc reacher reacher 4096
pmc reacher1024 reacher 1024
pmc reacher16384 reacher 16384
pmc reacher65536 reacher 65536
for WL in half_cheetah cartpole door tray gripper; do pmc $WL $WL 4096; done
python3 tools/pmc_summarize.py $TAG $OUT reacher reacher:1024 reacher:16384 reacher:65536 half_cheetah cartpole door tray gripper > $OUT/${TAG}_pmc_summary.txt 2>&1
for d in reacher reacher1024 reacher16384 reacher65536 half_cheetah cartpole door tray gripper; do for p in S1 S2 S3 S4 F W; do
  f=$(find $OUT/${TAG}_${d}_pmc$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${d}_pmc_${p}_counter_collection.csv
done; done
rm -rf $OUT/${TAG}_*_pmcS1 $OUT/${TAG}_*_pmcS2 $OUT/${TAG}_*_pmcS3 $OUT/${TAG}_*_pmcS4 $OUT/${TAG}_*_pmcF $OUT/${TAG}_*_pmcW
: > $OUT/${TAG}_tree_time.txt
for c in "4096 32 f64 cheetah" "4096 32 f64 swimmer" "4096 32 f64 hand" "4096 32 f64 handf" "4096 32 f64 pen" "4096 32 f32 pen" "4096 32 f64 tray" "4096 32 f64 door" "4096 32 f64 cartpole" "4096 32 f64 gripper" "65536 16 f64 hand"; do
  python3 tools/tree_time.py $c 2>/dev/null | tail -1 >> $OUT/${TAG}_tree_time.txt
done
if [ -f tools/_build/libmjmpc_stampsf.so ]; then
  for P in 1024 4096; do STAMPS_LIB=tools/_build/libmjmpc_stampsf.so python3 tools/stamps.py $P f64 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_arm_phase_clocks_$P.txt; done
fi
if [ -f tools/_build/libmjmpc_amd_treestats.so ]; then
  : > $OUT/${TAG}_tree_stats.txt
  for m in cheetah swimmer hand pen cartpole door tray gripper; do python3 tools/tree_stats.py $m f64 4096 32 2>/dev/null | tail -14 >> $OUT/${TAG}_tree_stats.txt; done
fi
fi
if [ "$PART" = c ]; then
# the sharded iteration on one GPU (tools/rccl_world1.py --time: a world-size-1 RCCL group whose communicator claims two ranks):
# ms per step with the library's all-gather (direct launches / launch tape) and with torch.distributed's (hipGraph replay), and
# the idle time around the collective from a kernel trace; then the gaps of the one-GPU loops as in round 4
G=$OUT/${TAG}_kernel_gaps.txt
: > $G
echo "== tools/rccl_world1.py --time (library all-gather: mjmpc_comm_all_gather_f64)" >> $G
python3 tools/rccl_world1.py --time 2>/dev/null | grep "ms per step\|sharded iteration" >> $G
echo "== MJMPC_TORCH_COLLECTIVES=1 tools/rccl_world1.py --time (torch.distributed's all-gather inside a replayed hipGraph)" >> $G
MJMPC_TORCH_COLLECTIVES=1 python3 tools/rccl_world1.py --time 2>/dev/null | grep "ms per step\|sharded iteration" >> $G
for v in "" 1; do
  rm -rf $OUT/kt
  MJMPC_TORCH_COLLECTIVES=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python3 tools/rccl_world1.py --time > /dev/null 2>&1
  echo "== kernel trace of the same run, MJMPC_TORCH_COLLECTIVES=$v (second half of the trace = the separate-launch and three-launch timing loops)" >> $G
  python3 tools/kernel_gaps.py $OUT/kt | head -10 >> $G
done
for c in "" "--controller cem --particles 4096" "--particles 16384" "--workload cartpole"; do
  for t in "" "--no-tape"; do
    [ -z "$c" ] && [ -n "$t" ] && continue
    rm -rf $OUT/kt
    rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python3 bench.py $c $t --steps 100 --warmup 10 --no-cpu-baseline > $OUT/kt_line.json 2>/dev/null
    echo "== bench.py $c $t" >> $G
    python3 -c "import json; j=json.loads(open('$OUT/kt_line.json').read().strip().splitlines()[-1]); print('   ms_per_step (under rocprofv3)', round(j['ms_per_step'], 4), '|', j['config']['launch'])" >> $G
    python3 tools/kernel_gaps.py $OUT/kt | head -8 >> $G
  done
done
rm -rf $OUT/kt $OUT/kt_line.json
fi
ls $OUT | grep $TAG

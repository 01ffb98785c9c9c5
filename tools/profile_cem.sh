OUT=gpurun_out; TAG=r04
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
: > $OUT/${TAG}_controllers.jsonl
for P in 16384 1024 4096; do python3 bench.py --controller cem --particles $P --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | tee $OUT/${TAG}_cem_${P}_line.json >> $OUT/${TAG}_controllers.jsonl; done
python3 bench.py --controller dmd --particles 4096 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_controllers.jsonl
for P in 16384 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_cem -o ${TAG}_cem -- python3 bench.py --controller cem --particles $P --steps 100 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_cem_${P}_line_under_rocprof.json 2> $OUT/${TAG}_prof_cem.err
  f=$(find $OUT/${TAG}_prof_cem -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_cem_${P}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_cem
done
cp $OUT/${TAG}_cem_16384_line_under_rocprof.json $OUT/${TAG}_cem_line_under_rocprof.json
bash tools/cem_phases.sh 16384 > $OUT/${TAG}_cem_select_phases.txt 2>&1; bash tools/cem_phases.sh 4096 >> $OUT/${TAG}_cem_select_phases.txt 2>&1
python3 tools/cem_time.py 16384 2>&1 | tail -6 > $OUT/${TAG}_cem_time.txt; python3 tools/cem_time.py 4096 2>&1 | tail -6 >> $OUT/${TAG}_cem_time.txt
cat $OUT/${TAG}_cem_time.txt; head -6 $OUT/${TAG}_cem_16384_kernel_stats.csv | cut -c1-60,200-400; head -6 $OUT/${TAG}_cem_4096_kernel_stats.csv | cut -c1-60,200-400

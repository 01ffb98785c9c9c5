export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cem_prof -o cem -- python3 bench.py --controller cem --particles 16384 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/cem_line.json 2> gpurun_out/cem.err
f=$(find gpurun_out/cem_prof -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(r['Name'][:90].replace('void mjmpc::(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1000,1), r['Percentage'])
PY
rm -rf gpurun_out/cem_prof

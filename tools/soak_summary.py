"""Summary of a random-model soak (tests/test_random_models_gpu.py run with MJMPC_FUZZ_STATS=file):
    python tools/soak_summary.py gpurun_out/fuzz_stats.jsonl
refusals / re-draws / scaled-tolerance hits / non-finite oracle rollouts per 1000 seeds, and what was refused."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
n = len(rows)
if not n:
    sys.exit("no seeds in " + sys.argv[1])
per = 1000.0 / n
red = sum(r["redraws"] for r in rows)
print("seeds %d (%d ... %d)" % (n, min(r["seed"] for r in rows), max(r["seed"] for r in rows)))
print("re-drawn models            %5d  = %.1f per 1000 seeds; seeds with any re-draw %d, most re-draws for one seed %d"
      % (red, red * per, sum(1 for r in rows if r["redraws"]), max(r["redraws"] for r in rows)))
hits = sum(r["scaled_tolerance_hits"] for r in rows)
print("rollouts beyond the one-step tolerance (1e-9 scaled), held to 1e-7 scaled: %d = %.1f per 1000 seeds (40 rollouts per seed), in %d seeds"
      % (hits, hits * per, sum(1 for r in rows if r["scaled_tolerance_hits"])))
nf = sum(r["nonfinite_oracle"] for r in rows)
print("non-finite oracle rollouts %5d  = %.1f per 1000 seeds" % (nf, nf * per))
why = collections.Counter(x.split(":")[1].strip()[:60] if ":" in x else x for r in rows for x in r["refusals"])
for k, c in why.most_common(12):
    print("   refused %4d x  %s" % (c, k))

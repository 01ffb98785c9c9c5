"""Per-step timeline from a rocprofv3 kernel trace (CSV): kernel durations and the gaps between consecutive kernels.
    python tools/timeline.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[skip:]
dur, gap_after = defaultdict(list), defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    n = n.split("(")[0][-60:]
    dur[n].append(e - s)
    if i + 1 < len(rows):
        gap_after[n].append(rows[i + 1][0] - e)
print("%-62s %6s %10s %12s" % ("kernel", "n", "avg us", "gap after us"))
for n in dur:
    g = gap_after.get(n, [0])
    print("%-62s %6d %10.2f %12.2f" % (n, len(dur[n]), sum(dur[n]) / len(dur[n]) / 1e3, sum(g) / max(1, len(g)) / 1e3))
span = rows[-1][1] - rows[0][0]
print("span %.1f us over %d kernels" % (span / 1e3, len(rows)))

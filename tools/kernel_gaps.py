"""Developer tool: idle time between consecutive kernels of a control loop, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py ... ; python tools/kernel_gaps.py DIR
prints, for the steady state (second half of the trace), the average duration of each kernel and the average gap to the next."""
import csv, glob, os, sys
from collections import defaultdict

def short(n):
    n = n.replace("void ", "").replace("mjmpc::(anonymous namespace)::", "").replace("mjmpc::", "")
    return n.split("(")[0][:52]


f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for a, b in zip(rows[:-1], rows[1:]):
    name = short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])
    dur[name] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    gap[name] += int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    cnt[name] += 1
for k in sorted(cnt, key=lambda k: -cnt[k])[:12]:
    print("%4d x  kernel %8.1f us, gap to next %6.1f us   %s" % (cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3, k))

"""Turn the PMC passes of tools/profile_round.sh into the per-launch figures bench.py reports:
    python tools/pmc_summarize.py r02 gpurun_out
writes <out>/<tag>_traffic_f64_4096x32.json and <out>/<tag>_valu_issue_f64_4096x32.json (copy them to profiles/)."""
import csv
import glob
import json
import os
import sys

tag, out = sys.argv[1], sys.argv[2]
KERNEL = "arm_rollout_kernel<double, false, false"


def rows(d, pat="*counter_collection.csv"):
    for f in glob.glob(os.path.join(out, tag + "_" + d, "**", pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def per_kernel(d, name_part):
    """counter -> list of per-dispatch values, and the dispatch durations (ns) of the matching kernel"""
    vals, dur = {}, {}
    for r in rows(d):
        if name_part in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return vals, list(dur.values())


def mean(x):
    return sum(x) / len(x) if x else float("nan")


# ---- HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are reported in KB; calibrated on the 64 MiB copy) ----
fk, _ = per_kernel("pmcF", KERNEL)
wk, _ = per_kernel("pmcW", KERNEL)
fc, _ = per_kernel("pmcF", "copyBuffer")        # dst.copy_(src): 65536 KiB read + written (the largest copies)
wc, _ = per_kernel("pmcW", "copyBuffer")
f_raw, w_raw = mean(fk.get("FETCH_SIZE", [])), mean(wk.get("WRITE_SIZE", []))
fcal = max(fc.get("FETCH_SIZE", [float("nan")]))
wcal = max(wc.get("WRITE_SIZE", [float("nan")]))
f_corr, w_corr = 65536.0 / fcal, 65536.0 / wcal
traffic = (f_raw * f_corr + w_raw * w_corr) * 1024.0
tj = {"kernel": KERNEL + ", 1, true> (two wavefronts per particle group)", "dtype": "f64", "particles": 4096, "horizon": 32,
      "FETCH_SIZE_raw_KB": f_raw, "WRITE_SIZE_raw_KB": w_raw,
      "calibration": {"what": "64 MiB contiguous device copy in the same process (65536 KiB read, 65536 KiB written)",
                      "FETCH_SIZE_raw_KB": fcal, "WRITE_SIZE_raw_KB": wcal, "fetch_correction": f_corr,
                      "write_correction": w_corr},
      "traffic_bytes_per_launch": traffic, "algorithmic_bytes_rollout_only": 4096 * 32 * 15 * 8,
      "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_run.py; "
                "corrections from the calibration copy (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md HBM section)"}
with open(os.path.join(out, "%s_traffic_f64_4096x32.json" % tag), "w") as f:
    json.dump(tj, f, indent=1)
print(json.dumps(tj, indent=1))

# ---- SQ: waves, instructions, busy fractions ----
s1, d1 = per_kernel("pmcS1", KERNEL)
s2, d2 = per_kernel("pmcS2", KERNEL)
m = {k: mean(v) for k, v in list(s1.items()) + list(s2.items())}
dur_ns = mean(d1)
simds, clock = 1024, 2.4e9
issue = m.get("SQ_INSTS_VALU", float("nan")) * 4.0 / (simds * dur_ns * 1e-9 * clock)
ij = {"kernel": KERNEL + ", 1, true>", "dtype": "f64", "particles": 4096, "horizon": 32, "counters_per_launch": m,
      "duration_ns_in_pmc_pass": dur_ns,
      "valu_issue_frac": issue,
      "formula": "SQ_INSTS_VALU x 4 cycles per wave64 instruction / (1024 SIMDs x kernel duration x 2.4 GHz)",
      "valu_busy_of_wave_cycles": m.get("SQ_ACTIVE_INST_VALU", float("nan")) / m.get("SQ_WAVE_CYCLES", float("nan")),
      "wait_any_of_wave_cycles": m.get("SQ_WAIT_ANY", float("nan")) / m.get("SQ_WAVE_CYCLES", float("nan")),
      "method": "rocprofv3 --kernel-trace --pmc <SQ counters> in two passes over tools/pmc_run.py 4096 f64 (tools/profile_round.sh)"}
with open(os.path.join(out, "%s_valu_issue_f64_4096x32.json" % tag), "w") as f:
    json.dump(ij, f, indent=1)
print(json.dumps(ij, indent=1))

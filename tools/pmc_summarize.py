"""Turn the PMC passes of tools/profile_round3.sh into the per-launch figures bench.py reports:
    python tools/pmc_summarize.py r03 gpurun_out
writes, per workload, <out>/<tag>_[<workload>_]traffic_f64_4096x32.json and <tag>_[<workload>_]valu_issue_f64_4096x32.json
(copy them to profiles/): HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, calibrated on a known copy), the share of the
chip's VALU issue slots the kernel fills, and the floating-point operations the kernel ITSELF executes per particle-step
(SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 + _F32, wave instructions x 64 lanes, an FMA = 2)."""
import csv
import glob
import json
import os
import sys

tag, out = sys.argv[1], sys.argv[2]
P, H = 4096, 32
WORKLOADS = {
    # workload -> (file prefix, dominant kernel, other kernels of the step reported beside it)
    "reacher": ("", "arm_rollout_kernel<double, false, false, 1, true, true>",
                ["arm_mppi_finish_kernel<double>", "arm_rollout_kernel<double, false, false, 1, true, false>"]),
    "half_cheetah": ("half_cheetah_", "tree_rollout_kernel<double, 8, 16, true, 16, 12", []),
}
# round 5: further entries on the command line, "workload[:P]" - the directory of their passes is <tag>_<workload>[<P>]_pmc*,
# the dominant kernel the instantiation of the workload's rollout kernel that ran longest in the S1 pass
EXTRA = []
for a in sys.argv[3:]:           # workload[:P[:H]]
    parts = a.split(":")
    EXTRA.append((parts[0], int(parts[1]) if len(parts) > 1 and parts[1] else 4096, int(parts[2]) if len(parts) > 2 else 32))


def rows(d, pat="*counter_collection.csv"):
    for f in glob.glob(os.path.join(out, tag + "_" + d, "**", pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def per_kernel(d, name_part):
    """counter -> list of per-dispatch values, and the dispatch durations (ns) of the matching kernel"""
    vals, dur = {}, {}
    match = [r for r in rows(d) if name_part in r["Kernel_Name"]]
    # (the full-size launches only: since round 5 an engine makes its reset record with a one-particle launch of the same
    # instantiation, which must not enter the per-launch means)
    big = max((int(r.get("Grid_Size", 0) or 0) for r in match), default=0)
    for r in match:
        if int(r.get("Grid_Size", 0) or 0) == big:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return vals, list(dur.values())


def mean(x):
    return sum(x) / len(x) if x else float("nan")


def flops(wl, kernel):
    v64, _ = per_kernel(wl + "_pmcS3", kernel)
    v32, _ = per_kernel(wl + "_pmcS4", kernel)
    m = {k: mean(v) for k, v in list(v64.items()) + list(v32.items())}
    g = lambda k: m.get(k, 0.0)        # noqa: E731
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + 2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")
    f32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_TRANS_F32")
    return m, 64.0 * f64, 64.0 * f32


def longest_kernel(d, base):
    tot = {}
    for r in rows(d):
        if base in r["Kernel_Name"]:
            tot.setdefault(r["Kernel_Name"], {})[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return max(tot, key=lambda k: sum(tot[k].values())) if tot else base


JOBS = [(wl, prefix, KERNEL, others, 4096, 32) for wl, (prefix, KERNEL, others) in WORKLOADS.items()] if not EXTRA else []
for wl_, p_, h_ in EXTRA:
    d_ = wl_ + (str(p_) if p_ != 4096 else "") + ("x%d" % h_ if h_ != 32 else "")
    base = "arm_rollout_kernel<double" if wl_ in ("reacher", "cartpole") else "tree_rollout_kernel<double"
    if wl_ == "reacher" and p_ <= 2048:         # (round 6: the four-wave flag shape takes these launches)
        base = "arm_rollout_flags_kernel<double"
    # (reacher: the fused iteration's rollout kernel - the instantiations whose last argument, MONO, is true)
    k_ = longest_kernel(d_ + "_pmcS1", base)
    if wl_ == "reacher" and 2048 < p_ <= 4096:  # the headline's kernel by name: the fused iteration's rollout launch (DUO, MONO)
        k_ = WORKLOADS["reacher"][1]
    JOBS.append((d_, ("" if wl_ == "reacher" else wl_ + "_"), k_, WORKLOADS["reacher"][2] if wl_ == "reacher" else [], p_, h_))
for wl, prefix, KERNEL, others, P, H in JOBS:
    # ---- HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are reported in KB; calibrated on the 64 MiB copy) ----
    fk, _ = per_kernel(wl + "_pmcF", KERNEL)
    wk, _ = per_kernel(wl + "_pmcW", KERNEL)
    fc, _ = per_kernel(wl + "_pmcF", "copyBuffer")        # dst.copy_(src): 65536 KiB read + written (the largest copies)
    wc, _ = per_kernel(wl + "_pmcW", "copyBuffer")
    if not fk:
        print("no PMC rows for", wl, KERNEL)
        continue
    f_raw, w_raw = mean(fk.get("FETCH_SIZE", [])), mean(wk.get("WRITE_SIZE", []))
    fcal = max(fc.get("FETCH_SIZE", [float("nan")]))
    wcal = max(wc.get("WRITE_SIZE", [float("nan")]))
    f_corr, w_corr = 65536.0 / fcal, 65536.0 / wcal
    traffic = (f_raw * f_corr + w_raw * w_corr) * 1024.0
    tj = {"kernel": KERNEL, "dtype": "f64", "particles": P, "horizon": H, "workload": wl,
          "FETCH_SIZE_raw_KB": f_raw, "WRITE_SIZE_raw_KB": w_raw,
          "calibration": {"what": "64 MiB contiguous device copy in the same process (65536 KiB read, 65536 KiB written)",
                          "FETCH_SIZE_raw_KB": fcal, "WRITE_SIZE_raw_KB": wcal, "fetch_correction": f_corr,
                          "write_correction": w_corr},
          "traffic_bytes_per_launch": traffic,
          "note": ("the fused iteration's rollout kernel draws its samples itself, keeps the actions in LDS and writes one "
                   "226-double record per workgroup: its HBM traffic is model + records, far below the ALGORITHMIC bytes "
                   "(SURVEY 8d: delta in, action + cost out, update re-reads), which is what roofline.achieved is defined on")
          if wl == "reacher" else "",
          "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_run.py; "
                    "corrections from the calibration copy (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md HBM section)"}
    with open(os.path.join(out, "%s_%straffic_f64_%dx%d.json" % (tag, prefix, P, H)), "w") as f:
        json.dump(tj, f, indent=1)
    print(json.dumps(tj, indent=1))

    # ---- SQ: waves, instructions, busy fractions, the kernel's own FLOPs ----
    s1, d1 = per_kernel(wl + "_pmcS1", KERNEL)
    s2, d2 = per_kernel(wl + "_pmcS2", KERNEL)
    m = {k: mean(v) for k, v in list(s1.items()) + list(s2.items())}
    fm, fl64, fl32 = flops(wl, KERNEL)
    m.update(fm)
    dur_ns = mean(d1)
    simds, clock = 1024, 2.4e9
    issue = m.get("SQ_INSTS_VALU", float("nan")) * 4.0 / (simds * dur_ns * 1e-9 * clock)
    ij = {"kernel": KERNEL, "dtype": "f64", "particles": P, "horizon": H, "workload": wl, "counters_per_launch": m,
          "duration_ns_in_pmc_pass": dur_ns,
          "valu_issue_frac": issue,
          "formula": "SQ_INSTS_VALU x 4 cycles per wave64 instruction / (1024 SIMDs x kernel duration x 2.4 GHz)",
          "valu_busy_of_wave_cycles": m.get("SQ_ACTIVE_INST_VALU", float("nan")) / m.get("SQ_WAVE_CYCLES", float("nan")),
          "wait_any_of_wave_cycles": m.get("SQ_WAIT_ANY", float("nan")) / m.get("SQ_WAVE_CYCLES", float("nan")),
          "kernel_flops_per_launch": {"f64": fl64, "f32": fl32},
          "kernel_flops_per_particle_step": (fl64 + fl32) / (P * H),
          "kernel_flops_formula": "(SQ_INSTS_VALU_ADD + MUL + TRANS + 2 x FMA, _F64 and _F32) x 64 lanes / (P x H): what the kernel's "
                                  "own instruction stream executes, spare lanes and the work both wavefronts of a particle "
                                  "group duplicate included",
          "other_kernels": {},
          "method": "rocprofv3 --kernel-trace --pmc <SQ counters> in four passes over tools/pmc_run.py %s 4096 f64 "
                    "(tools/profile_round3.sh)" % wl}
    for ok in others:
        o1, od = per_kernel(wl + "_pmcS1", ok)
        _, of64, of32 = flops(wl, ok)
        if od:
            ij["other_kernels"][ok] = {"duration_ns": mean(od), "SQ_INSTS_VALU": mean(o1.get("SQ_INSTS_VALU", [])),
                                       "SQ_WAVES": mean(o1.get("SQ_WAVES", [])),
                                       "kernel_flops_per_particle_step": (of64 + of32) / (P * H)}
    with open(os.path.join(out, "%s_%svalu_issue_f64_%dx%d.json" % (tag, prefix, P, H)), "w") as f:
        json.dump(ij, f, indent=1)
    print(json.dumps(ij, indent=1))

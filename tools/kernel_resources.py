#!/usr/bin/env python
"""Rebuild the HIP library and print the register / LDS / scratch figures of the kernels whose name contains argv[1]."""
import re
import subprocess
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "rollout_kernel"
out = subprocess.run([sys.executable, "-m", "mjmpc_amd.build", "--force"], capture_output=True, text=True)
txt = out.stdout + out.stderr
if out.returncode != 0:
    print(txt[-3000:])
    sys.exit(1)
cur, rows = None, {}
for ln in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur and pat in cur:
        rows.setdefault(cur, {})[m.group(1).split(" [")[0]] = int(m.group(2))
for k, v in rows.items():
    name = re.sub(r"^_ZN5mjmpc12_GLOBAL__N_1\d+", "", k)[:46]
    print("%-46s vgpr %3d agpr %3d scratch %3d occ %d vspill %2d sspill %3d lds %6d" % (
        name, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1),
        v.get("VGPRs Spill", -1), v.get("SGPRs Spill", -1), v.get("LDS Size", -1)))

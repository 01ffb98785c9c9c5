import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
H, P = 32, 4096
for dt in ("f64",):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    tdt = torch.float64
    g = torch.Generator(device="cuda").manual_seed(0)
    noise = torch.randn(P, H, 7, device="cuda", dtype=tdt, generator=g)
    mean = torch.zeros(H, 7, device="cuda", dtype=torch.float64)
    coeffs = torch.tensor([0.25, 0.8, 0.0], dtype=torch.float64, device="cuda")
    gseq = torch.ones(H, dtype=torch.float64, device="cuda")
    for _ in range(3):
        eng.rollout_fused(P, H, mean, noise, coeffs, gseq)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): eng.rollout_fused(P, H, mean, noise, coeffs, gseq)
    e1.record(); torch.cuda.synchronize()
    print(dt, "fused %.4f ms" % (e0.elapsed_time(e1) / 10))

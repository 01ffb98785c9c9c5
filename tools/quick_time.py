"""Quick kernel timing on the GPU box: python tools/quick_time.py [P] [H]"""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw

H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
Ps = [int(sys.argv[1])] if len(sys.argv) > 1 else [1024, 4096, 16384, 65536]
for dt in ("f32", "f64"):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    for P in Ps:
        g = torch.Generator(device="cuda").manual_seed(0)
        noise = torch.randn(P, H, 7, device="cuda", dtype=torch.float32 if dt == "f32" else torch.float64, generator=g)
        mean = torch.zeros(H, 7, device="cuda", dtype=torch.float64)
        for want_obs in (False, True):
            for _ in range(3):
                eng.rollout_device(P, H, mean, noise, want_obs=want_obs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                eng.rollout_device(P, H, mean, noise, want_obs=want_obs)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print("%s P=%6d H=%d obs=%d  %.3f ms/rollout  %.1f M particle-steps/s  fails=%d"
                  % (dt, P, H, want_obs, ms, P * H / ms / 1e3, eng.solver_failures()), flush=True)

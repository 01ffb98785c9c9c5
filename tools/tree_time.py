"""Rollout-kernel time of the tree engine on the 24-dof hand: python tools/tree_time.py [P] [H] [dtype]
(MJMPC_AMD_LIB selects an alternative build of the library, e.g. one compiled with -DTREE_SKIP=...)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
from mjmpc_amd.models.hand24 import hand24_raw
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt = sys.argv[3] if len(sys.argv) > 3 else "f64"
eng = TreeRolloutEngine(hand24_raw(), dtype=dt)
g = torch.Generator(device="cuda").manual_seed(0)
noise = 0.5 * torch.randn(P, H, 24, device="cuda", dtype=torch.float32 if dt == "f32" else torch.float64, generator=g)
mean = torch.zeros(H, 24, device="cuda", dtype=torch.float64)
eng.rollout_device(P, H, mean, noise)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    eng.rollout_device(P, H, mean, noise)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%s P=%d H=%d: %.2f ms/rollout, %.2f us per wave-substep at 2048 resident waves, fails=%d"
      % (dt, P, H, ms, ms * 1e3 * 2048 / (P * H * 2 / 2), eng.solver_failures()), flush=True)

"""experiment: read the diag counter after one bench-like rollout"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from mjmpc_amd.control.control_utils import generate_noise
P, H = 4096, 32
for dt in ("f64", "f32"):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    noise = torch.from_numpy(generate_noise(np.eye(7), [0.25, 0.8, 0.0], (P, H), 123)).cuda()
    if dt == "f32": noise = noise.float()
    mean = torch.zeros(H, 7, dtype=torch.float64, device="cuda")
    eng.rollout_device(P, H, mean, noise)
    torch.cuda.synchronize()
    print(dt, "counter", eng.solver_failures(), "wave-substeps", P // 8 * H * 2)

"""Rollout launches + a calibration copy for PMC collection: python tools/pmc_run.py [P] [dtype]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from mjmpc_amd.control.control_utils import generate_noise
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = sys.argv[2] if len(sys.argv) > 2 else "f64"
H = 32
eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
noise = torch.from_numpy(generate_noise(np.eye(7), [0.25, 0.8, 0.0], (P, H), 123)).cuda()
if dt == "f32":
    noise = noise.float()
mean = torch.zeros(H, 7, dtype=torch.float64, device="cuda")
for _ in range(3):
    eng.rollout_device(P, H, mean, noise)
torch.cuda.synchronize()
# calibration: a contiguous device-to-device copy of known size (64 MiB read + 64 MiB written, far beyond L2)
src = torch.empty(8 * 1024 * 1024, dtype=torch.float64, device="cuda").normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()

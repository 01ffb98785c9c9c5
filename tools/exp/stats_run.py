import sys, numpy as np, torch
sys.path.insert(0, ".")
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from oracle import controllers_ref as cr
eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
P, H = 4096, 32
noise = cr.generate_noise(np.eye(7), [0.25, 0.8, 0.0], (P, H), 123)
eng.rollout_device(P, H, np.zeros((H, 7)), noise)
print("fails", eng.solver_failures())

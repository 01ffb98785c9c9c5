"""Back-to-back timing of the one-launch iteration (mjmpc_arm_mppi_step) and of the plain fused rollout, for A/B builds
(MJMPC_AMD_LIB=tools/_build/<lib>.so).   python tools/mono_time.py [P] [H] [dtype]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.control import MPPI
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dtype)
eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, init_cov=1.0, base_action="null", lam=0.01,
         num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
         action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=123, noise_mode="device", noise_dtype=dtype)
chol, coeffs, _ = c.dev.prepare_noise(c.cov_action, c.filter_coeffs)
step = torch.zeros(1, dtype=torch.int64, device="cuda")
act = torch.zeros(7, dtype=torch.float64, device="cuda")
slots = torch.zeros(16, dtype=torch.float64).pin_memory()
noise = c.dev.sample_noise(P, c.cov_action, c.filter_coeffs, 123, 0, dtype=dtype, filtered=False)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def mono(env=True):
    eng.mppi_step(P, H, c.dev.mean, c.dev.mean_alt, c.dev.gseq, coeffs, chol, 123, 0, 0, step, 0.01, 1.0, 0, action_out=act,
                  action_slots=slots, env_step=env)


print("lib", os.environ.get("MJMPC_AMD_LIB", "default"), "P", P, "H", H, dtype)
print("  fused rollout        %.1f us" % timeit(lambda: eng.rollout_fused(P, H, c.dev.mean, noise, coeffs, c.dev.gseq)))
print("  one-launch iteration %.1f us (with env step)" % timeit(lambda: mono(True)))
print("  one-launch iteration %.1f us (without env step)" % timeit(lambda: mono(False)))
print("  solver failures", eng.solver_failures())

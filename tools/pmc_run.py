"""Launches for PMC collection (rocprofv3 --pmc passes, tools/profile_round3.sh):
    python tools/pmc_run.py [reacher|half_cheetah|swimmer|hand24|pen_hand|cartpole|door|tray|gripper] [P] [dtype] [H]
reacher: three fused control iterations (mjmpc_arm_mppi_step: rollout kernel + finish kernel, the default loop) and three
plain rollouts - above 4096 particles four launches of mjmpc_arm_rollout_fused, the captured iteration's rollout; the tree models: three rollouts from their bench start state.  Then a calibration copy (64 MiB)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "reacher"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dt = sys.argv[3] if len(sys.argv) > 3 else "f64"
H = int(sys.argv[4]) if len(sys.argv) > 4 else 32
tdt = torch.float32 if dt == "f32" else torch.float64
if wl == "reacher":
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.control.control_utils import generate_noise
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
    c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, init_cov=1.0, base_action="null", lam=0.01,
             num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
             action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=123, noise_mode="device", noise_dtype=dt)
    chol, coeffs, _ = c.dev.prepare_noise(c.cov_action, c.filter_coeffs)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    noise = torch.from_numpy(generate_noise(np.eye(7), [0.25, 0.8, 0.0], (P, H), 123)).cuda().to(tdt)
    mean = torch.zeros(H, 7, dtype=torch.float64, device="cuda")
    if P <= 4096:           # the fused two-launch iteration (what the control loop runs up to 4096 particles)
        for _ in range(3):
            eng.mppi_step(P, H, c.dev.mean, c.dev.mean_alt, c.dev.gseq, coeffs, chol, 123, 0, 0, step, 0.01, 1.0, 0, env_step=False)
        for _ in range(3):
            eng.rollout_device(P, H, mean, noise)
    else:                   # above: the captured iteration's rollout launch (filter + rollout + cost-to-go on raw samples)
        raw = torch.randn(P, H, 7, device="cuda", dtype=tdt)
        for _ in range(4):
            eng.rollout_fused(P, H, mean, raw, coeffs, c.dev.gseq)
else:
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    if wl == "hand24":
        from mjmpc_amd.models.hand24 import hand24_raw
        raw, st, scale = hand24_raw(), None, 0.55
    elif wl == "pen_hand":
        from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
        raw, st, scale = pen_hand_raw(), holding_state(), 0.1
    elif wl in ("cartpole", "door", "tray", "gripper"):     # the synthetic MJCF models, from their bench start state
        from mjmpc_amd.models.synthetic import start_state, synthetic_raw
        raw = synthetic_raw(wl)
        st, scale = start_state(wl, raw), 0.3
    else:
        from mjmpc_amd.envs import locomotion_env
        from mjmpc_amd.models.half_cheetah import half_cheetah_raw
        from mjmpc_amd.models.swimmer import swimmer_raw
        raw = dict(half_cheetah=half_cheetah_raw, swimmer=swimmer_raw)[wl]()
        env = dict(half_cheetah=locomotion_env.HalfCheetahEnv, swimmer=locomotion_env.SwimmerEnv)[wl](dtype=dt)
        env.reset(seed=123)
        st, scale = env.get_env_state(), 0.55
    if wl in ("cartpole", "door", "tray", "gripper"):      # (round 6: the arm kernels where the model fits them - the cart-pole)
        from mjmpc_amd.envs import make_engine
        eng = make_engine(raw, dtype=dt)
    else:
        eng = TreeRolloutEngine(raw, dtype=dt)
    if st is not None:
        eng.set_env_state(dict(st, target_pos=np.asarray(raw.target_pos, float)) if ("qp" in st and "target_pos" not in st) else st)
    A = eng.d_action
    g = torch.Generator(device="cuda").manual_seed(0)
    noise = scale * torch.randn(P, H, A, device="cuda", dtype=tdt, generator=g)
    if wl in ("cartpole", "door", "tray", "gripper"):
        noise = noise * torch.from_numpy((eng.action_highs - eng.action_lows) / 2).to(noise)
    mean = torch.zeros(H, A, dtype=torch.float64, device="cuda")
    if wl == "pen_hand":
        mean += torch.from_numpy(st["qp"][6:]).cuda()
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

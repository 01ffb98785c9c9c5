"""Long randomized parity sweep, HIP kernels against the FP64 oracle (a checker, not collected by pytest; the bounded versions live in
tests/test_stress_parity_gpu.py and tests/test_stress_locomotion_gpu.py).

    python tests/soak_parity.py [trials] [seed]

Per trial: a random model-appropriate start state, mean and filtered noise; arm 4096 x 32 (every cost), 24-dof hand
512 x 16, pen-in-hand 256 x 8 from the settled pose, one env step of the cheetah / swimmer from 64 random states.  Prints the
worst relative cost error per model and the solver-failure counters."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
from mjmpc_amd.models.half_cheetah import half_cheetah_raw
from mjmpc_amd.models.hand24 import hand24_raw
from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from mjmpc_amd.models.swimmer import swimmer_raw
from oracle.physics_ref import RefArm, threads

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
threads(0)


def filt(eps):
    for t in range(2, eps.shape[1]):
        eps[:, t] = 0.25 * eps[:, t] + 0.8 * eps[:, t - 1]
    return eps


def rel(c, rew):
    return float((np.abs(c + rew) / np.maximum(1.0, np.abs(rew))).max())


worst = {}
t0 = time.time()
# ---- arm
raw = reacher7dof_raw()
eng, ref = ArmRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
lo = np.array([-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5])
hi = np.array([1.714602, 1.3963, 1.7, 0.0, 1.5, 0.0, 1.5])
for k in range(trials):
    q = lo + (hi - lo) * rs.rand(7)
    if k % 3 == 0:
        q[1], q[3] = 0.9 + 0.4 * rs.rand(), -0.1 * rs.rand()
    v = rs.randn(7) * rs.choice([0.3, 1.0, 4.0])
    tgt = np.array([rs.uniform(-.3, .3), rs.uniform(-.2, .2), rs.uniform(-.25, .25)])
    noise = filt(rs.choice([0.3, 1.0, 3.0]) * rs.standard_normal((4096, 32, 7)))
    mean = 0.5 * rs.standard_normal((32, 7))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    c = eng.rollout_device(4096, 32, mean, noise, want_actions=False)[0].cpu().numpy()
    rew = ref.rollout(q, v, tgt, mean, noise, want_obs=False)[1]
    worst["arm 4096x32"] = max(worst.get("arm 4096x32", 0.0), rel(c, rew))
print("arm: %d trials, worst %.2e, failures kernel %d oracle %d  (%.0f s)"
      % (trials, worst["arm 4096x32"], eng.solver_failures(), ref.newton_stats()["fails"], time.time() - t0), flush=True)
# ---- hand
raw = hand24_raw()
eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
joints = [b.joint for b in raw.bodies if b.joint is not None]
lo, hi = np.array([j.range[0] for j in joints]), np.array([j.range[1] for j in joints])
for k in range(trials):
    q = lo + (hi - lo) * (0.05 + 0.9 * rs.rand(24))
    if k % 2 == 0:
        q[:4] = [0.1, 0.5 + 0.15 * rs.rand(), -0.2, 0.3]
    v = rs.randn(24) * rs.choice([0.3, 2.0])
    tgt = np.array(raw.target_pos) + 0.1 * rs.randn(3)
    noise = filt(rs.choice([0.2, 0.7]) * rs.standard_normal((512, 16, 24)))
    mean = 0.2 * rs.standard_normal((16, 24))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    c = eng.rollout_device(512, 16, mean, noise, want_actions=False)[0].cpu().numpy()
    rew = ref.rollout(q, v, tgt, mean, noise, want_obs=False)[1]
    worst["hand 512x16"] = max(worst.get("hand 512x16", 0.0), rel(c, rew))
print("hand: worst %.2e, failures kernel %d oracle %d  (%.0f s)"
      % (worst["hand 512x16"], eng.solver_failures(), ref.newton_stats()["fails"], time.time() - t0), flush=True)
# ---- pen in hand, from the settled pose
raw = pen_hand_raw()
eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
st = holding_state()
q, v = st["qp"].copy(), st["qv"].copy()
u = q[6:].copy()
for _ in range(400):
    q, v = ref.step(q, v, u)[:2]
tgt = np.asarray(raw.target_pos, float)
unstable = 0
for k in range(trials):
    qq = q + np.concatenate([0.002 * rs.randn(3), 0.05 * rs.randn(3), 0.03 * rs.randn(24)])
    vv = 0.2 * rs.randn(30)
    noise = filt(rs.choice([0.02, 0.1, 0.3]) * rs.standard_normal((256, 8, 24)))
    mean = np.tile(u, (8, 1))
    eng.set_env_state(dict(qp=qq, qv=vv, target_pos=tgt))
    c = eng.rollout_device(256, 8, mean, noise, want_actions=False)[0].cpu().numpy()
    o = ref.rollout(qq, vv, tgt, mean, noise)
    rew = o[1]
    # A rollout that goes numerically unstable (stiff servos on gram-sized links under 0.3 rad of set-point noise: joint
    # speeds beyond 10^3 rad/s, then 10^20 - on BOTH sides alike; MuJoCo would reset the simulation there)
    # amplifies a rounding difference without bound: compared up to the step before the oracle's speeds leave 500 rad/s.
    fast = np.abs(o[4][..., 30:60]).max(axis=2) > 500.0                  # (256, 8)
    ok = np.cumsum(fast, axis=1) == 0
    unstable += int((~ok[:, -1]).sum())
    err = np.abs(c + rew) / np.maximum(1.0, np.abs(rew))
    worst["pen 256x8"] = max(worst.get("pen 256x8", 0.0), float(np.where(ok, err, 0.0).max()))
print("pen: worst %.2e (%d of %d rollouts go unstable on both sides and are compared up to there), failures kernel %d oracle %d  (%.0f s)"
      % (worst["pen 256x8"], unstable, trials * 256, eng.solver_failures(), ref.newton_stats()["fails"], time.time() - t0), flush=True)
# ---- locomotion: one env step from many states
for name, raw_fn in (("cheetah", half_cheetah_raw), ("swimmer", swimmer_raw)):
    raw = raw_fn()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    nv, A = ref.nv, eng.d_action
    w = 0.0
    for k in range(4 * trials):
        q0, v0 = 0.3 * rs.standard_normal(nv), rs.choice([0.5, 3.0]) * rs.standard_normal(nv)
        if name == "cheetah":
            q0[1] = rs.uniform(-0.25, 0.1)
            q0[2] = rs.uniform(-1.5, 1.5)
        else:
            q0[3:] = rs.choice([-1.0, 1.0]) * rs.uniform(0.0, 1.5, 4) if k % 2 else q0[3:]
        noise = 2.0 * rs.standard_normal((64, 1, A))
        eng.set_env_state(dict(qpos=q0, qvel=v0))
        out = eng.rollout_device(64, 1, np.zeros((1, A)), noise, want_obs=True)
        o = ref.rollout(q0, v0, np.zeros(3), np.zeros((1, A)), noise)
        w = max(w, rel(out[0].cpu().numpy(), o[1]), float(np.abs(out[3].cpu().numpy() - o[4]).max()))
    print("%s one step from %d states: worst %.2e, failures kernel %d oracle %d  (%.0f s)"
          % (name, 4 * trials, w, eng.solver_failures(), ref.newton_stats()["fails"], time.time() - t0), flush=True)

import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle.physics_ref import RefArm
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
from mjmpc_amd.models.swimmer import swimmer_raw
from mjmpc_amd.models.half_cheetah import half_cheetah_raw
for name, fn, nu in (("cheetah", half_cheetah_raw, 6),):
    raw = fn(); ref = RefArm(raw.to_flat())
    rs = np.random.RandomState(1)
    nv = ref.nv
    P, H = 256, 32
    q0 = 0.1 * rs.randn(nv); v0 = 0.3 * rs.randn(nv)
    q0[1] = -0.05
    mean = 0.3 * rs.randn(H, nu); noise = 0.7 * rs.randn(P, H, nu)
    o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
    print(name, "oracle newton", ref.newton_stats())
    for dt in ("f64", "f32"):
        eng = TreeRolloutEngine(raw, dtype=dt)
        eng.set_env_state(dict(qpos=q0, qvel=v0))
        obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise)
        e = np.abs(nobs - o[4]).max(axis=2)           # (P, H)
        print(dt, "per-step max err", " ".join("%.1e" % x for x in e.max(axis=0)))
        print(dt, "per-step median err", " ".join("%.1e" % x for x in np.median(e, axis=0)))
        print(dt, "frac particles with final err > 1e-2:", (e[:, -1] > 1e-2).mean(), "fails", eng.solver_failures())

"""One-off randomized sweep at a larger scale than the test suite's (developer tool): python tests/big_sweep.py [trials]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from mjmpc_amd.models.half_cheetah import half_cheetah_raw
from mjmpc_amd.models.hand24 import hand24_raw
from oracle.physics_ref import RefArm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rs = np.random.RandomState(777)
raw = reacher7dof_raw(); ref = RefArm(raw.to_flat()); eng = ArmRolloutEngine(raw, dtype="f64")
lo = np.array([-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5]); hi = np.array([1.714602, 1.3963, 1.7, 0.0, 1.5, 0.0, 1.5])
worst, t0 = 0.0, time.time()
for trial in range(n):
    q = lo + (hi - lo) * rs.rand(7)
    if trial % 3 == 0: q[1] = 0.9 + 0.4 * rs.rand(); q[3] = -0.1 * rs.rand()
    if trial % 4 == 1: q[rs.randint(7)] = hi[rs.randint(7)] + 0.03            # start past a limit
    v = rs.randn(7) * [0.5, 3.0, 6.0][trial % 3]
    tgt = np.array([rs.uniform(-.3, .3), rs.uniform(-.2, .2), rs.uniform(-.25, .25)])
    P, H = 4096, 32
    noise = [0.3, 1.0, 3.0][trial % 3] * rs.standard_normal((P, H, 7))
    for t in range(2, H): noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
    mean = 0.5 * rs.standard_normal((H, 7))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    c, _, _, _ = eng.rollout_device(P, H, mean, noise, want_actions=False)
    _, rew, _, _, _ = ref.rollout(q, v, tgt, mean, noise, want_obs=False)
    err = np.abs(c.cpu().numpy() + rew) / np.maximum(1.0, np.abs(rew))
    worst = max(worst, float(err.max()))
    if err.max() > 1e-9: print("ARM trial", trial, "err %.2e" % err.max(), flush=True)
print("arm: %d x 4096 x 32, worst rel err %.2e, kernel fails %d oracle fails %d, %.0f s" % (n, worst, eng.solver_failures(), ref.newton_stats()["fails"], time.time() - t0), flush=True)
for name, rawf, nv, nu in (("hand", hand24_raw, 24, 24), ("cheetah", half_cheetah_raw, 9, 6)):
    raw = rawf(); ref = RefArm(raw.to_flat()); eng = TreeRolloutEngine(raw, dtype="f64")
    joints = [b.joint for b in raw.bodies if b.joint is not None]
    worst, t0 = 0.0, time.time()
    for trial in range(n):
        if name == "hand":
            lo = np.array([j.range[0] for j in joints]); hi = np.array([j.range[1] for j in joints])
            q = lo + (hi - lo) * (-0.05 + 1.1 * rs.rand(24))
            if trial % 2 == 0: q[:4] = [0.1, 0.5 + 0.2 * rs.rand(), -0.2, 0.3]
            v = rs.randn(24) * [0.3, 2.0, 5.0][trial % 3]
            st = dict(qp=q, qv=v, target_pos=np.array(raw.target_pos)); tg = np.array(raw.target_pos)
            P, H = 512, 8
        else:
            q = 0.5 * rs.randn(9); q[1] = rs.uniform(-0.5, 0.2); q[2] = rs.uniform(-np.pi, np.pi); v = 5.0 * rs.randn(9)
            st = dict(qpos=q, qvel=v); tg = np.zeros(3)
            P, H = 512, 3
        noise = [0.3, 1.0, 3.0][trial % 3] * rs.standard_normal((P, H, nu)); mean = 0.3 * rs.standard_normal((H, nu))
        eng.set_env_state(st)
        obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise)
        o = ref.rollout(q, v, tg, mean, noise)
        e = max(np.abs(rew - o[1]).max() / (1 + np.abs(o[1]).max()), np.abs(nobs - o[4]).max() / (1 + np.abs(o[4]).max()))
        worst = max(worst, e)
        if e > 1e-8: print(name, "trial", trial, "err %.2e" % e, flush=True)
    print("%s: %d trials, worst rel err %.2e, kernel fails %d oracle fails %d, %.0f s" % (name, n, worst, eng.solver_failures(), ref.newton_stats()["fails"], time.time() - t0), flush=True)

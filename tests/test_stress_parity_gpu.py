"""GPU: randomized parity sweep of the arm rollout at the headline size - 16 random start states (inside the joint
ranges, some pressed into limits or towards the table, slow and fast), three noise scales, random means, 4096 x 32 each
(2 x 10^6 particle-steps), every cost against the FP64 oracle.  Exercises the rarely taken branches of the two-wave
solver (several rows flipping in one particle -> general re-factorisation, contact-row flips, limit + contact rows
together).  Measured worst relative error 1e-11; asserted 1e-9; no solver failures on either side."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_states_4096x32(raw_arm, ref_arm):
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    eng = ArmRolloutEngine(raw_arm, dtype="f64")
    rs = np.random.RandomState(2026)
    lo = np.array([-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5])
    hi = np.array([1.714602, 1.3963, 1.7, 0.0, 1.5, 0.0, 1.5])
    P, H, worst = 4096, 32, 0.0
    for trial in range(16):
        q = lo + (hi - lo) * rs.rand(7)
        if trial % 3 == 0:                      # arm lowered towards the table, elbow near its upper limit
            q[1] = 0.9 + 0.4 * rs.rand()
            q[3] = -0.1 * rs.rand()
        v = rs.randn(7) * (3.0 if trial % 2 else 0.5)
        tgt = np.array([rs.uniform(-.3, .3), rs.uniform(-.2, .2), rs.uniform(-.25, .25)])
        noise = [0.3, 1.0, 3.0][trial % 3] * rs.standard_normal((P, H, 7))
        for t in range(2, H):
            noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
        mean = 0.5 * rs.standard_normal((H, 7))
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        c, _, _, _ = eng.rollout_device(P, H, mean, noise, want_actions=False)
        _, rew, _, _, _ = ref_arm.rollout(q, v, tgt, mean, noise, want_obs=False)
        err = np.abs(c.cpu().numpy() + rew) / np.maximum(1.0, np.abs(rew))
        worst = max(worst, float(err.max()))
    print("worst relative cost error over 16 x 4096 x 32: %.2e" % worst)
    assert worst < 1e-9
    assert eng.solver_failures() == 0 and ref_arm.newton_stats()["fails"] == 0


def test_random_states_tree_24dof():
    """The same sweep for the tree kernel on the 24-dof hand: 8 random configurations inside the joint ranges (arm
    lowered in half of them so that fingertips meet the table), 512 x 16 each."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.hand24 import hand24_raw
    from oracle.physics_ref import RefArm
    raw = hand24_raw()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    joints = [b.joint for b in raw.bodies if b.joint is not None]
    lo, hi = np.array([j.range[0] for j in joints]), np.array([j.range[1] for j in joints])
    rs = np.random.RandomState(7)
    P, H, worst = 512, 16, 0.0
    for trial in range(8):
        q = lo + (hi - lo) * (0.1 + 0.8 * rs.rand(24))
        if trial % 2 == 0:
            q[:4] = [0.1, 0.5 + 0.15 * rs.rand(), -0.2, 0.3]
        v = rs.randn(24) * (2.0 if trial % 4 < 2 else 0.3)
        tgt = np.array(raw.target_pos) + 0.1 * rs.randn(3)
        noise = [0.2, 0.7][trial % 2] * rs.standard_normal((P, H, 24))
        for t in range(2, H):
            noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
        mean = 0.2 * rs.standard_normal((H, 24))
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        c, _, _, _ = eng.rollout_device(P, H, mean, noise, want_actions=False)
        _, rew, _, _, _ = ref.rollout(q, v, tgt, mean, noise, want_obs=False)
        err = np.abs(c.cpu().numpy() + rew) / np.maximum(1.0, np.abs(rew))
        worst = max(worst, float(err.max()))
    print("tree: worst relative cost error over 8 x 512 x 16: %.2e" % worst)
    assert worst < 1e-9
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0

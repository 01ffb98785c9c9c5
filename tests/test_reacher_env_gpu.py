"""GPU: the reacher_7dof-v0 / continual_reacher-v0 environment objects (mjmpc/envs/basic/reacher_env.py) stepped by
the HIP engine at P = 1, against the FP64 oracle's env step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_env_steps_follow_the_oracle(ref_arm):
    from mjmpc_amd.envs.reacher_env import Reacher7DOFEnv
    env = Reacher7DOFEnv()
    ob0 = env.reset(seed=5)
    st = env.get_env_state()
    assert set(st) == {"qp", "qv", "qa", "target_pos", "timestep"} and st["timestep"] == 0
    assert ob0.shape == (20,) and np.all(ob0[:14] == 0.0)
    np.testing.assert_allclose(ob0[17:20], ob0[14:17] - st["target_pos"], atol=1e-15)
    rs = np.random.RandomState(0)
    q, v = np.zeros(7), np.zeros(7)
    for t in range(12):
        a = rs.uniform(-1.5, 1.5, 7)
        ob, r, done, info = env.step(a)
        q, v, r_ref, ob_ref = ref_arm.env_step(q, v, a, st["target_pos"])
        np.testing.assert_allclose(ob, ob_ref, rtol=1e-9, atol=1e-10)
        assert abs(r - r_ref) < 1e-9 and done is False
        assert info["state"]["timestep"] == t + 1
        assert info["goal_achieved"] == (np.linalg.norm(ob[17:20]) < 0.025)
    # state round trip: restoring a saved state reproduces the continuation
    saved = env.get_env_state()
    a = rs.uniform(-1, 1, 7)
    ob1, r1, _, _ = env.step(a)
    env.set_env_state(saved)
    ob2, r2, _, _ = env.step(a)
    np.testing.assert_array_equal(ob1, ob2)
    assert r1 == r2
    paths = [dict(env_infos=dict(goal_achieved=np.array([1] * 11 + [0] * 5))), dict(env_infos=dict(goal_achieved=np.zeros(16)))]
    assert env.evaluate_success(paths) == 50.0


def test_continual_reacher_redraws_its_target_every_50_real_steps():
    from mjmpc_amd.envs.reacher_env import ContinualReacher7DOFEnv
    env = ContinualReacher7DOFEnv()
    env.reset(seed=3)
    targets = [env.get_env_state()["target_pos"].copy()]
    for t in range(1, 102):
        env.step(np.zeros(7))
        targets.append(env.get_env_state()["target_pos"].copy())
    changed = [t for t in range(1, 102) if not np.array_equal(targets[t], targets[t - 1])]
    assert changed == [50, 100]
    lo, hi = np.array([-0.3, -0.2, -0.25]), np.array([0.3, 0.2, 0.25])
    assert np.all(targets[50] >= lo) and np.all(targets[50] <= hi)
    env.reset(seed=3)
    env.real_env_step(False)                    # rollout copies of the env never trigger timed events
    for _ in range(51):
        env.step(np.zeros(7))
    assert np.array_equal(env.get_env_state()["target_pos"], targets[0])

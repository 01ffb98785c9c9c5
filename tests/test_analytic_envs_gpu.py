"""GPU, PINNED end to end: rollouts of the reference's analytic envs by the HIP kernel, and whole
``optimize()`` closed loops (HIP rollout + HIP update), against golden vectors captured by running the
reference's own GymEnvWrapper.rollout / controllers (tests/golden/e2e.npz).  FP64; tolerance 1e-12 on
single rollouts (libm vs device sin/cos differ in the last bit), 1e-10 after closed loops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(golden, tag):
    from mjmpc_amd.envs.analytic_engine import AnalyticRolloutEngine
    g = golden("e2e")
    if tag.startswith("pend"):
        return AnalyticRolloutEngine.pendulum()
    return AnalyticRolloutEngine.lqr(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])


@pytest.mark.parametrize("tag", ["pend_mppi", "pend_rs", "lqr_cem", "lqr_dmd"])
def test_first_rollout_matches_reference(golden, tag):
    g = golden("e2e")
    eng = _engine(golden, tag)
    noise = g[tag + "_first_noise"]
    P, H, A = noise.shape
    eng.set_env_state({"state": g[tag + "_states"][0]})
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, A)), noise, "open_loop")
    np.testing.assert_allclose(obs, g[tag + "_first_observations"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(nobs, g[tag + "_first_next_observations"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(-rew, g[tag + "_first_costs"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(act, g[tag + "_first_actions"])
    assert np.array_equal(done, g[tag + "_first_dones"])


def test_lqr_mean_only(golden):
    g = golden("e2e")
    eng = _engine(golden, "lqr")
    eng.set_env_state({"state": np.array([1.0, -2.0, 0.5])})
    obs, rew, act, done, info, nobs = eng.rollout(1, 8, 0.1 * np.ones((8, 2)), None, "open_loop")
    np.testing.assert_allclose(obs, g["lqr_meanonly_obs"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(rew, g["lqr_meanonly_rew"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(act, g["lqr_meanonly_act"])


def _closed_loop(golden, tag, make, step_env):
    from mjmpc_amd.envs.arm_engine import make_rollout_fn
    g = golden("e2e")
    eng = _engine(golden, tag)
    c = make()
    c.set_sim_state_fn = eng.set_env_state
    c.rollout_fn = make_rollout_fn(eng)
    s = g[tag + "_states"][0].copy()
    for k in range(g[tag + "_actions"].shape[0]):
        np.testing.assert_allclose(s, g[tag + "_states"][k], rtol=1e-11, atol=1e-11)
        a, _ = c.optimize({"state": s.copy()})
        np.testing.assert_allclose(a, g[tag + "_actions"][k], rtol=1e-10, atol=1e-10)
        s = step_env(eng, s, a)
    np.testing.assert_allclose(c.mean_action, g[tag + "_final_mean"], rtol=1e-10, atol=1e-10)


def _gpu_env_step(eng, s, a):
    """The 'real' env stepped by the same kernel at P = 1 (state = last entries of next_obs for LQR;
    the pendulum keeps (th, thdot), recovered from the rollout of the same state)."""
    eng.set_env_state({"state": s})
    obs, rew, act, done, info, nobs = eng.rollout(1, 1, np.asarray(a, float).reshape(1, -1), None, "open_loop")
    if eng.d_obs == eng.d_state:
        return nobs[0, 0].copy()
    th = s[0] + nobs[0, 0, 2] * 0.05 if abs(nobs[0, 0, 2]) < 8.0 else None
    if th is None:                                  # clipped speed: th advanced with the UNclipped one
        from oracle.envs_ref import PendulumRef
        return PendulumRef().step(s, a)[0]
    return np.array([th, nobs[0, 0, 2]])


def test_closed_loop_pendulum_mppi(golden):
    from mjmpc_amd.control import MPPI
    kw = dict(d_state=2, d_obs=3, d_action=1, horizon=10, num_particles=48, n_iters=1,
              action_lows=np.array([-2.0]), action_highs=np.array([2.0]), seed=123)
    _closed_loop(golden, "pend_mppi", lambda: MPPI(init_cov=0.8, base_action="null", lam=0.1, step_size=0.9, alpha=1,
                                                    gamma=0.99, filter_coeffs=[0.25, 0.8, 0.0], **kw), _gpu_env_step)


def test_closed_loop_lqr_cem_and_dmd(golden):
    from mjmpc_amd.control import CEM, DMDMPC
    kw = dict(d_state=3, d_obs=3, d_action=2, horizon=8, num_particles=40, n_iters=2,
              action_lows=-np.ones(2) * 5, action_highs=np.ones(2) * 5, seed=77)
    _closed_loop(golden, "lqr_cem", lambda: CEM(init_cov=1.0, base_action="null", elite_frac=0.2, step_size=0.8,
                                                 gamma=1.0, beta=0.1, cov_type="full", filter_coeffs=[1.0, 0.0, 0.0],
                                                 **kw), _gpu_env_step)
    _closed_loop(golden, "lqr_dmd", lambda: DMDMPC(init_cov=1.0, beta=0.1, base_action="null", lam=0.5, step_size=0.7,
                                                    gamma=1.0, update_cov=True, cov_type="diagonal",
                                                    filter_coeffs=[1.0, 0.0, 0.0], **kw), _gpu_env_step)


def test_closed_loop_linear_rollouts_match_the_reference_wrapper(golden):
    """mode='closed_loop_linear' (gym_env_wrapper.py:133-136) on both analytic kernels against vectors produced
    by the reference's GymEnvWrapper.rollout; f64 to 1e-12, f32 to its rounding."""
    from mjmpc_amd.envs.analytic_engine import AnalyticRolloutEngine
    g = golden("closed_loop")
    for dtype, tol in (("f64", 1e-12), ("f32", 3e-5)):
        eng = AnalyticRolloutEngine.pendulum(dtype=dtype)
        eng.set_env_state({"state": g["pend_state"]})
        P, H, _ = g["pend_noise"].shape
        obs, rew, act, done, info, nobs = eng.rollout(P, H, g["pend_W"], g["pend_noise"], "closed_loop_linear")
        for got, want in ((obs, "pend_obs"), (rew, "pend_rew"), (act, "pend_act"), (nobs, "pend_nobs")):
            np.testing.assert_allclose(got, g[want], rtol=tol, atol=tol)
        obs, rew, act, done, info, nobs = eng.rollout(1, H, g["pend_W"], None, "closed_loop_linear")
        np.testing.assert_allclose(act, g["pend_mean_act"], rtol=tol, atol=tol)
        np.testing.assert_allclose(rew, g["pend_mean_rew"], rtol=tol, atol=tol)
        eng = AnalyticRolloutEngine.lqr(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"], dtype=dtype)
        eng.set_env_state({"state": g["lqr_state"]})
        P, H, _ = g["lqr_noise"].shape
        obs, rew, act, done, info, nobs = eng.rollout(P, H, g["lqr_W"], g["lqr_noise"], "closed_loop_linear")
        for got, want in ((obs, "lqr_obs"), (rew, "lqr_rew"), (act, "lqr_act"), (nobs, "lqr_nobs")):
            np.testing.assert_allclose(got, g[want], rtol=tol, atol=tol)

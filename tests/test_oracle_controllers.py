"""The numpy oracle (oracle/controllers_ref.py, oracle/envs_ref.py) against the golden vectors
captured from the reference (tests/golden/make_fixtures.py).  CPU only."""
import numpy as np
import pytest

from oracle import controllers_ref as cr
from oracle import envs_ref as er

TOL = dict(rtol=1e-12, atol=1e-12)      # controller maths vs reference: FP64, <= 1e-12


def test_generate_noise_bit_exact(golden):
    g = golden("noise")
    for tag in "abcde":
        eps = cr.generate_noise(g[tag + "_cov"], list(g[tag + "_coeffs"]), tuple(g[tag + "_shape"]),
                                int(g[tag + "_seed"]))
        assert np.array_equal(eps, g[tag + "_eps"]), tag
    eps = cr.generate_noise(g["g_cov"], list(g["g_coeffs"]), tuple(g["g_shape"]), int(g["g_seed"]))
    np.testing.assert_allclose(eps, g["g_eps"], rtol=1e-10, atol=1e-12)   # SVD path: LAPACK dependent


def test_isotropic_noise_is_scaled_standard_normal(golden):
    """for cov = c*I the legacy stream equals sqrt(c)*standard_normal bit-for-bit."""
    g = golden("noise")
    np.random.seed(int(g["c_seed"]))
    z = np.sqrt(3.5) * np.random.standard_normal(tuple(g["c_shape"]) + (7,))
    b0, b1, b2 = g["c_coeffs"]
    for t in range(2, z.shape[1]):
        z[:, t] = b0 * z[:, t] + b1 * z[:, t - 1] + b2 * z[:, t - 2]
    np.testing.assert_allclose(z, g["c_eps"], rtol=1e-15, atol=0)


def test_cost_to_go(golden):
    g = golden("cost_to_go")
    for tag in ("g1", "g99", "g0"):
        H = g[tag + "_costs"].shape[1]
        out = cr.cost_to_go(g[tag + "_costs"].copy(), cr.gamma_seq(float(g[tag + "_gamma"]), H))
        assert np.array_equal(out, g[tag + "_out"]), tag


def test_mppi_updates(golden):
    g = golden("updates")
    for i in range(int(g["mppi_n"])):
        t = "mppi%d" % i
        lam, alpha, tbw, gamma, step, c0 = g[t + "_cfg"]
        H = g[t + "_mean0"].shape[0]
        gs = cr.gamma_seq(gamma, H)
        m1 = cr.mppi_update(g[t + "_costs"], g[t + "_actions"], g[t + "_mean0"], g[t + "_cov0"], gs, lam,
                            int(alpha), step, bool(tbw))
        np.testing.assert_allclose(m1, g[t + "_mean1"], **TOL)
        np.testing.assert_allclose(cr.shift_mean(m1, str(g[t + "_base"])), g[t + "_mean2"], **TOL)
        if t + "_val" in g.files:
            v = cr.mppi_value(g[t + "_costs"], g[t + "_actions"], g[t + "_mean0"], g[t + "_cov0"], gs, lam, int(alpha))
            np.testing.assert_allclose(v, g[t + "_val"], **TOL)
        else:
            assert bool(tbw)


def test_cem_updates(golden):
    g = golden("updates")
    for i in range(int(g["cem_n"])):
        t = "cem%d" % i
        elite, beta, gamma, step, c0 = g[t + "_cfg"]
        H, A = g[t + "_mean0"].shape
        gs = cr.gamma_seq(gamma, H)
        m1, c1 = cr.cem_update(g[t + "_costs"], g[t + "_actions"], g[t + "_mean0"], g[t + "_cov0"], gs, elite,
                               step, str(g[t + "_covtype"]))
        np.testing.assert_allclose(m1, g[t + "_mean1"], **TOL)
        np.testing.assert_allclose(c1, g[t + "_cov1"], **TOL)
        np.testing.assert_allclose(cr.cem_shift_cov(c1, beta, np.full(A, c0)), g[t + "_cov2"], **TOL)
        np.testing.assert_allclose(cr.mean_value(g[t + "_costs"], gs), g[t + "_val"], **TOL)


def test_dmd_updates(golden):
    g = golden("updates")
    for i in range(int(g["dmd_n"])):
        t = "dmd%d" % i
        lam, beta, gamma, step, c0, ucov = g[t + "_cfg"]
        H = g[t + "_mean0"].shape[0]
        gs = cr.gamma_seq(gamma, H)
        m1, c1 = cr.dmd_update(g[t + "_costs"], g[t + "_actions"], g[t + "_mean0"], g[t + "_cov0"], gs, lam, step,
                               bool(ucov), str(g[t + "_covtype"]))
        np.testing.assert_allclose(m1, g[t + "_mean1"], **TOL)
        np.testing.assert_allclose(c1, g[t + "_cov1"], **TOL)
        np.testing.assert_allclose(cr.shift_mean(m1, "repeat"), g[t + "_mean2"], **TOL)
        np.testing.assert_allclose(cr.dmd_shift_cov(c1, beta, bool(ucov)), g[t + "_cov2"], **TOL)
        np.testing.assert_allclose(cr.dmd_value(g[t + "_costs"], gs, lam), g[t + "_val"], **TOL)


def test_rs_updates(golden):
    g = golden("updates")
    for i in range(int(g["rs_n"])):
        t = "rs%d" % i
        gamma, step, c0 = g[t + "_cfg"]
        gs = cr.gamma_seq(gamma, g[t + "_mean0"].shape[0])
        m1 = cr.rs_update(g[t + "_costs"], g[t + "_actions"], g[t + "_mean0"], gs, step)
        np.testing.assert_allclose(m1, g[t + "_mean1"], **TOL)
        np.testing.assert_allclose(cr.shift_mean(m1, "null"), g[t + "_mean2"], **TOL)
        np.testing.assert_allclose(cr.mean_value(g[t + "_costs"], gs), g[t + "_val"], **TOL)


def test_pfmpc(golden):
    g = golden("updates")
    seed = 123
    for i in range(int(g["pf_n"])):
        t = "pf%d" % i
        lam, gamma, cshift, cres = g[t + "_cfg"]
        s0 = g[t + "_samples0"]
        P, H, A = s0.shape
        # constructor draw: generate_noise(cov_resample, coeffs, (P,H), seed)
        assert np.array_equal(cr.generate_noise(cres * np.eye(A), [0.25, 0.8, 0.0], (P, H), seed), s0)
        w = cr.pf_weights(g[t + "_costs"], cr.gamma_seq(gamma, H), lam)
        s1, m1 = cr.pf_resample(s0, w, seed + 0)
        assert np.array_equal(s1, g[t + "_samples1"])
        np.testing.assert_allclose(m1, g[t + "_mean1"], **TOL)
        s2 = cr.pf_shift(s1, cshift * np.eye(A), [0.25, 0.8, 0.0], seed + 1, str(g[t + "_base"]))
        np.testing.assert_allclose(s2, g[t + "_samples2"], **TOL)


@pytest.mark.parametrize("tag", ["pend_mppi", "pend_rs"])
def test_pendulum_rollout_loop(golden, tag):
    g = golden("e2e")
    env = er.PendulumRef()
    noise = g[tag + "_first_noise"]
    P, H, _ = noise.shape
    obs, rew, act, done, nobs = er.rollout(env, g[tag + "_states"][0], P, H, np.zeros((H, 1)), noise)
    np.testing.assert_allclose(obs, g[tag + "_first_observations"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(nobs, g[tag + "_first_next_observations"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(-rew, g[tag + "_first_costs"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(act, g[tag + "_first_actions"])
    assert np.array_equal(done, g[tag + "_first_dones"])


@pytest.mark.parametrize("tag", ["lqr_cem", "lqr_dmd"])
def test_lqr_rollout_loop(golden, tag):
    g = golden("e2e")
    env = er.LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])
    noise = g[tag + "_first_noise"]
    P, H, A = noise.shape
    obs, rew, act, done, nobs = er.rollout(env, g[tag + "_states"][0], P, H, np.zeros((H, A)), noise)
    np.testing.assert_allclose(obs, g[tag + "_first_observations"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(nobs, g[tag + "_first_next_observations"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(-rew, g[tag + "_first_costs"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(act, g[tag + "_first_actions"])


def test_lqr_mean_only(golden):
    g = golden("e2e")
    env = er.LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])
    obs, rew, act, done, nobs = er.rollout(env, np.array([1.0, -2.0, 0.5]), 1, 8, 0.1 * np.ones((8, 2)), None)
    np.testing.assert_allclose(obs, g["lqr_meanonly_obs"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(rew, g["lqr_meanonly_rew"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(act, g["lqr_meanonly_act"])


def _run_e2e(make_update, env, g, tag, P, H, A, seed, n_iters, noise_cov, coeffs, base="null", zero_seq=False,
             sample=False):
    """Controller.optimize() loop (controller.py:207-257) rebuilt from the oracle pieces."""
    mean = np.zeros((H, A))
    cov = noise_cov * np.eye(A)
    state = g[tag + "_states"][0].copy()
    steps = g[tag + "_actions"].shape[0]
    for k in range(steps):
        np.testing.assert_allclose(state, g[tag + "_states"][k], rtol=1e-12, atol=1e-12)
        for _ in range(n_iters):
            noise = cr.generate_noise(cov, coeffs, (P, H), seed + k)
            if zero_seq:
                noise[-1] = -mean                       # olgaussian_mpc.py:110-111
            obs, rew, act, done, nobs = er.rollout(env, state, P, H, mean, noise)
            mean, cov = make_update(-rew, act, mean, cov)
        a = mean[0].copy()
        if sample:                                      # olgaussian_mpc.py:72-75
            a = a + cr.generate_noise(cov, coeffs, (1, 1), seed + 123 * k).reshape(A)
        np.testing.assert_allclose(a, g[tag + "_actions"][k], rtol=1e-10, atol=1e-10)
        mean, cov = make_update.shift(mean, cov)
        state, _ = env.step(state, a)
    np.testing.assert_allclose(mean, g[tag + "_final_mean"], rtol=1e-10, atol=1e-10)


def test_e2e_pendulum_mppi(golden):
    g = golden("e2e")
    H = 10
    gs = cr.gamma_seq(0.99, H)

    def upd(costs, actions, mean, cov):
        return cr.mppi_update(costs, actions, mean, cov, gs, 0.1, 1, 0.9), cov
    upd.shift = lambda mean, cov: (cr.shift_mean(mean, "null"), cov)
    _run_e2e(upd, er.PendulumRef(), g, "pend_mppi", 48, H, 1, 123, 1, 0.8, [0.25, 0.8, 0.0])


def test_e2e_pendulum_mppi_branches(golden):
    """use_zero_control_seq, base_action='random' (global numpy stream, left behind the step's (P, H) draw) and
    sample_mode='sample' (olgaussian_mpc.py:110-111, 122-123, 72-75)."""
    g = golden("e2e")
    H = 10
    gs = cr.gamma_seq(0.99, H)

    def upd(costs, actions, mean, cov):
        return cr.mppi_update(costs, actions, mean, cov, gs, 0.1, 1, 0.9), cov
    upd.shift = lambda mean, cov: (cr.shift_mean(mean, "null"), cov)
    _run_e2e(upd, er.PendulumRef(), g, "pend_zero", 48, H, 1, 123, 1, 0.8, [0.25, 0.8, 0.0], zero_seq=True)
    _run_e2e(upd, er.PendulumRef(), g, "pend_sample", 48, H, 1, 123, 1, 0.8, [0.25, 0.8, 0.0], sample=True)
    upd.shift = lambda mean, cov: (cr.shift_mean(mean, "random", np.array([0.8])), cov)
    _run_e2e(upd, er.PendulumRef(), g, "pend_random", 48, H, 1, 123, 1, 0.8, [0.25, 0.8, 0.0])
    assert np.abs(g["pend_random_final_mean"][-1]).max() > 0          # the random row really is there


def test_e2e_lqr_cem(golden):
    g = golden("e2e")
    H = 8
    gs = cr.gamma_seq(1.0, H)
    env = er.LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])

    def upd(costs, actions, mean, cov):
        return cr.cem_update(costs, actions, mean, cov, gs, 0.2, 0.8, "full")
    upd.shift = lambda mean, cov: (cr.shift_mean(mean, "null"), cr.cem_shift_cov(cov, 0.1, np.ones(2)))
    _run_e2e(upd, env, g, "lqr_cem", 40, H, 2, 77, 2, 1.0, [1.0, 0.0, 0.0])


def test_e2e_lqr_dmd(golden):
    g = golden("e2e")
    H = 8
    gs = cr.gamma_seq(1.0, H)
    env = er.LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])

    def upd(costs, actions, mean, cov):
        return cr.dmd_update(costs, actions, mean, cov, gs, 0.5, 0.7, True, "diagonal")
    upd.shift = lambda mean, cov: (cr.shift_mean(mean, "null"), cr.dmd_shift_cov(cov, 0.1, True))
    _run_e2e(upd, env, g, "lqr_dmd", 40, H, 2, 77, 2, 1.0, [1.0, 0.0, 0.0])


def test_mppiq(golden):
    """MPPIQ (mppiq.py:73-165): TD(lambda) returns, update and value, against the reference's outputs."""
    g = golden("mppiq")
    for i in range(int(g["n"])):
        t = "q%d" % i
        beta, alpha, tbw, gamma, td_lam, step, c0, with_q = g[t + "_cfg"]
        costs, actions, mean0 = g[t + "_costs"], g[t + "_actions"], g[t + "_mean0"]
        qvals = g[t + "_qvals"] if with_q else None
        cov = c0 * np.eye(actions.shape[-1])
        total = costs + beta * cr.mppiq_control_costs(mean0, cov, actions - mean0[None], int(alpha))
        np.testing.assert_allclose(cr.mppiq_returns(total, qvals, gamma, td_lam), g[t + "_returns"], **TOL)
        m1 = cr.mppiq_update(costs, actions, qvals, mean0, cov, beta, int(alpha), gamma, td_lam, step, bool(tbw))
        np.testing.assert_allclose(m1, g[t + "_mean1"], **TOL)
        v = cr.mppiq_value(costs, actions, qvals, mean0, cov, beta, int(alpha), gamma, td_lam)
        np.testing.assert_allclose(v, g[t + "_val"], **TOL)


def test_closed_loop_linear_rollout_loop(golden):
    """mode='closed_loop_linear' of GymEnvWrapper.rollout (gym_env_wrapper.py:133-136) over both analytic envs."""
    g = golden("closed_loop")
    env = er.PendulumRef()
    P, H, _ = g["pend_noise"].shape
    obs, rew, act, done, nobs = er.rollout(env, g["pend_state"], P, H, g["pend_W"], g["pend_noise"], "closed_loop_linear")
    for got, want in ((obs, "pend_obs"), (rew, "pend_rew"), (act, "pend_act"), (nobs, "pend_nobs")):
        np.testing.assert_allclose(got, g[want], rtol=1e-13, atol=1e-13)
    obs, rew, act, done, nobs = er.rollout(env, g["pend_state"], 1, H, g["pend_W"], None, "closed_loop_linear")
    np.testing.assert_allclose(act, g["pend_mean_act"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(rew, g["pend_mean_rew"], rtol=1e-13, atol=1e-13)
    env = er.LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])
    P, H, _ = g["lqr_noise"].shape
    obs, rew, act, done, nobs = er.rollout(env, g["lqr_state"], P, H, g["lqr_W"], g["lqr_noise"], "closed_loop_linear")
    for got, want in ((obs, "lqr_obs"), (rew, "lqr_rew"), (act, "lqr_act"), (nobs, "lqr_nobs")):
        np.testing.assert_allclose(got, g[want], rtol=1e-13, atol=1e-13)

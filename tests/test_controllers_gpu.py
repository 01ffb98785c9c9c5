"""GPU parity of the controller updates (HIP reductions behind the reference-shaped classes)
against the golden vectors captured from the reference (tests/golden/updates.npz, e2e.npz).
Tolerance: controller maths in FP64, <= 1e-12 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-12, atol=1e-12)
H, A, P = 10, 4, 64


def _kw(**extra):
    kw = dict(d_state=5, d_obs=6, d_action=A, horizon=H, num_particles=P, n_iters=1,
              action_lows=-np.ones(A), action_highs=np.ones(A), seed=123)
    kw.update(extra)
    return kw


def _traj(g, t, as_tensor=False, dtype=np.float64):
    d = dict(costs=g[t + "_costs"].astype(dtype), actions=g[t + "_actions"].astype(dtype))
    if as_tensor:
        import torch
        d = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    return d


def _check_cycle(c, g, t, has_val=True):
    c.mean_action = g[t + "_mean0"].copy()
    c.cov_action = g[t + "_cov0"].copy()
    if has_val and t + "_val" in g.files:
        np.testing.assert_allclose(c._calc_val(_traj(g, t)), g[t + "_val"], **TOL)
        np.testing.assert_array_equal(c.mean_action, g[t + "_mean0"])         # _calc_val must not move the mean
    c._update_distribution(_traj(g, t))
    np.testing.assert_allclose(c.mean_action, g[t + "_mean1"], **TOL)
    np.testing.assert_allclose(c.cov_action, g[t + "_cov1"], **TOL)
    c.num_steps += 1
    c._shift()
    np.testing.assert_allclose(c.mean_action, g[t + "_mean2"], **TOL)
    np.testing.assert_allclose(c.cov_action, g[t + "_cov2"], **TOL)


def test_mppi_updates(golden):
    from mjmpc_amd.control import MPPI
    g = golden("updates")
    for i in range(int(g["mppi_n"])):
        t = "mppi%d" % i
        lam, alpha, tbw, gamma, step, c0 = g[t + "_cfg"]
        c = MPPI(init_cov=c0, base_action=str(g[t + "_base"]), lam=lam, step_size=step, alpha=int(alpha), gamma=gamma,
                 time_based_weights=bool(tbw), filter_coeffs=[0.25, 0.8, 0.0], **_kw())
        _check_cycle(c, g, t)
        if bool(tbw):
            with pytest.raises(ValueError):
                c._calc_val(_traj(g, t))


def test_mppi_accepts_device_tensors_and_f32(golden):
    from mjmpc_amd.control import MPPI
    g = golden("updates")
    t = "mppi0"
    lam, alpha, tbw, gamma, step, c0 = g[t + "_cfg"]
    c = MPPI(init_cov=c0, base_action="null", lam=lam, step_size=step, alpha=1, gamma=gamma, **_kw())
    c.mean_action = g[t + "_mean0"].copy()
    c._update_distribution(_traj(g, t, as_tensor=True))
    np.testing.assert_allclose(c.mean_action, g[t + "_mean1"], **TOL)
    # f32 storage of costs/actions: accumulation stays f64, error is input rounding only.
    # lam = 0.01 amplifies cost rounding by 1/lam in the exponent -> compare on the lam = 0.2 case
    t = "mppi8"
    lam, alpha, tbw, gamma, step, c0 = g[t + "_cfg"]
    assert lam == 0.2 and alpha == 1 and not tbw
    c = MPPI(init_cov=c0, base_action="null", lam=lam, step_size=step, alpha=1, gamma=gamma, **_kw())
    c.mean_action = g[t + "_mean0"].copy()
    c._update_distribution(_traj(g, t, as_tensor=True, dtype=np.float32))
    np.testing.assert_allclose(c.mean_action, g[t + "_mean1"], rtol=0, atol=2e-5)


def test_cem_updates(golden):
    from mjmpc_amd.control import CEM
    g = golden("updates")
    for i in range(int(g["cem_n"])):
        t = "cem%d" % i
        elite, beta, gamma, step, c0 = g[t + "_cfg"]
        c = CEM(init_cov=c0, base_action="null", elite_frac=elite, step_size=step, gamma=gamma, beta=beta,
                cov_type=str(g[t + "_covtype"]), **_kw())
        _check_cycle(c, g, t)


def test_dmd_updates(golden):
    from mjmpc_amd.control import DMDMPC
    g = golden("updates")
    for i in range(int(g["dmd_n"])):
        t = "dmd%d" % i
        lam, beta, gamma, step, c0, ucov = g[t + "_cfg"]
        c = DMDMPC(init_cov=c0, beta=beta, base_action="repeat", lam=lam, step_size=step, gamma=gamma,
                   update_cov=bool(ucov), cov_type=str(g[t + "_covtype"]), **_kw())
        _check_cycle(c, g, t)


def test_rs_updates(golden):
    from mjmpc_amd.control import RandomShooting
    g = golden("updates")
    for i in range(int(g["rs_n"])):
        t = "rs%d" % i
        gamma, step, c0 = g[t + "_cfg"]
        c = RandomShooting(init_cov=c0, base_action="null", step_size=step, gamma=gamma, **_kw())
        _check_cycle(c, g, t)


def test_pfmpc(golden):
    from mjmpc_amd.control import PFMPC
    g = golden("updates")
    for i in range(int(g["pf_n"])):
        t = "pf%d" % i
        lam, gamma, cshift, cres = g[t + "_cfg"]
        c = PFMPC(cov_shift=cshift, cov_resample=cres, base_action=str(g[t + "_base"]), lam=lam, gamma=gamma,
                  filter_coeffs=[0.25, 0.8, 0.0], **_kw())
        assert np.array_equal(c.action_samples, g[t + "_samples0"])
        c._update_distribution(dict(costs=g[t + "_costs"], actions=c.action_samples.copy()))
        assert np.array_equal(c.action_samples, g[t + "_samples1"])
        np.testing.assert_allclose(c.mean_action, g[t + "_mean1"], **TOL)
        c.num_steps += 1
        c._shift()
        np.testing.assert_allclose(c.action_samples, g[t + "_samples2"], **TOL)


# ---- optimize() end to end with a user-supplied python rollout_fn (the reference's callback
# contract): the env arithmetic comes from the numpy env oracle, everything else is the product.
def _e2e(golden, tag, make, env, steps_check=True):
    from oracle import envs_ref as er
    g = golden("e2e")
    state = {"cur": None}

    def set_state(s):
        state["cur"] = np.asarray(s["state"], float).reshape(-1).copy()

    def rollout_fn(num_particles, horizon, mean, noise, mode):
        if hasattr(noise, "cpu"):               # device samplers hand over CUDA tensors
            noise = noise.cpu().numpy().astype(np.float64)
        obs, rew, act, done, nobs = er.rollout(env, state["cur"], num_particles, horizon, mean, noise)
        return dict(observations=obs, actions=act, costs=-rew, dones=done, next_observations=nobs)

    c = make()
    c.set_sim_state_fn = set_state
    c.rollout_fn = rollout_fn
    s = g[tag + "_states"][0].copy()
    for k in range(g[tag + "_actions"].shape[0]):
        np.testing.assert_allclose(s, g[tag + "_states"][k], rtol=1e-11, atol=1e-11)
        a, _ = c.optimize({"state": s.copy()})
        np.testing.assert_allclose(a, g[tag + "_actions"][k], rtol=1e-10, atol=1e-10)
        s, _ = env.step(s, a)
    np.testing.assert_allclose(c.mean_action, g[tag + "_final_mean"], rtol=1e-10, atol=1e-10)


def test_e2e_pendulum_mppi(golden):
    from mjmpc_amd.control import MPPI
    from oracle.envs_ref import PendulumRef
    kw = dict(d_state=2, d_obs=3, d_action=1, horizon=10, num_particles=48, n_iters=1,
              action_lows=np.array([-2.0]), action_highs=np.array([2.0]), seed=123)
    _e2e(golden, "pend_mppi", lambda: MPPI(init_cov=0.8, base_action="null", lam=0.1, step_size=0.9, alpha=1,
                                            gamma=0.99, filter_coeffs=[0.25, 0.8, 0.0], **kw), PendulumRef())


@pytest.mark.parametrize("tag,extra", [("pend_zero", dict(use_zero_control_seq=True)),
                                       ("pend_random", dict(base_action="random")),
                                       ("pend_sample", dict(sample_mode="sample"))])
def test_e2e_pendulum_mppi_branches(golden, tag, extra):
    """The three olgaussian_mpc.py branches (:110-111 zero-control particle, :122-123 random last row from the
    global numpy stream, :72-75 sampled action) against the reference's own optimize() sequences."""
    from mjmpc_amd.control import MPPI
    from oracle.envs_ref import PendulumRef
    kw = dict(d_state=2, d_obs=3, d_action=1, horizon=10, num_particles=48, n_iters=1,
              action_lows=np.array([-2.0]), action_highs=np.array([2.0]), seed=123)
    args = dict(init_cov=0.8, base_action="null", lam=0.1, step_size=0.9, alpha=1, gamma=0.99,
                filter_coeffs=[0.25, 0.8, 0.0])
    args.update(extra)
    _e2e(golden, tag, lambda: MPPI(**args, **kw), PendulumRef())


def test_e2e_pendulum_rs(golden):
    from mjmpc_amd.control import RandomShooting
    from oracle.envs_ref import PendulumRef
    kw = dict(d_state=2, d_obs=3, d_action=1, horizon=10, num_particles=48, n_iters=1,
              action_lows=np.array([-2.0]), action_highs=np.array([2.0]), seed=123)
    _e2e(golden, "pend_rs", lambda: RandomShooting(init_cov=0.8, base_action="null", step_size=1.0, gamma=1.0,
                                                    filter_coeffs=[1.0, 0.0, 0.0], **kw), PendulumRef())


def test_e2e_lqr_cem_and_dmd(golden):
    from mjmpc_amd.control import CEM, DMDMPC
    from oracle.envs_ref import LQRRef
    g = golden("e2e")
    env = LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])
    kw = dict(d_state=3, d_obs=3, d_action=2, horizon=8, num_particles=40, n_iters=2,
              action_lows=-np.ones(2) * 5, action_highs=np.ones(2) * 5, seed=77)
    _e2e(golden, "lqr_cem", lambda: CEM(init_cov=1.0, base_action="null", elite_frac=0.2, step_size=0.8, gamma=1.0,
                                         beta=0.1, cov_type="full", filter_coeffs=[1.0, 0.0, 0.0], **kw), env)
    _e2e(golden, "lqr_dmd", lambda: DMDMPC(init_cov=1.0, beta=0.1, base_action="null", lam=0.5, step_size=0.7,
                                            gamma=1.0, update_cov=True, cov_type="diagonal",
                                            filter_coeffs=[1.0, 0.0, 0.0], **kw), env)


def test_e2e_lqr_cem_seed_identical_on_the_device(golden):
    """Full-covariance CEM with the reference's own noise stream regenerated on the GPU (noise_mode
    'device_mt19937', general covariance through numpy's SVD colouring): the reference's optimize() sequence."""
    from mjmpc_amd.control import CEM
    from oracle.envs_ref import LQRRef
    g = golden("e2e")
    env = LQRRef(g["lqr_A"], g["lqr_B"], g["lqr_Q"], g["lqr_R"])
    kw = dict(d_state=3, d_obs=3, d_action=2, horizon=8, num_particles=40, n_iters=2,
              action_lows=-np.ones(2) * 5, action_highs=np.ones(2) * 5, seed=77, noise_mode="device_mt19937")
    _e2e(golden, "lqr_cem", lambda: CEM(init_cov=1.0, base_action="null", elite_frac=0.2, step_size=0.8, gamma=1.0,
                                         beta=0.1, cov_type="full", filter_coeffs=[1.0, 0.0, 0.0], **kw), env)


def test_device_noise_statistics():
    """Performance-mode sampler: N(0, cov) + the recursive filter, checked by moments."""
    from mjmpc_amd.control._device import DeviceUpdater
    Pn, Hn, An = 20000, 6, 3
    dev = DeviceUpdater(Hn, An, np.ones(Hn))
    rs = np.random.RandomState(0)
    B = rs.randn(An, An)
    cov = B @ B.T + 0.3 * np.eye(An)
    raw = dev.sample_noise(Pn, cov, [1.0, 0.0, 0.0], 7, 3).cpu().numpy()
    flat = raw.reshape(-1, An)
    np.testing.assert_allclose(flat.mean(0), 0, atol=5 * np.sqrt(np.diag(cov).max() / flat.shape[0]))      # five sigma of a sample mean
    np.testing.assert_allclose(np.cov(flat, rowvar=False), cov, rtol=0.05, atol=0.03)
    assert abs(np.corrcoef(raw[:, 0, 0], raw[:, 1, 0])[0, 1]) < 0.03          # white along the horizon
    co = [0.25, 0.8, 0.1]
    filt = dev.sample_noise(Pn, cov, co, 7, 3).cpu().numpy()
    want = raw.copy()
    for t in range(2, Hn):
        want[:, t] = co[0] * want[:, t] + co[1] * want[:, t - 1] + co[2] * want[:, t - 2]
    np.testing.assert_allclose(filt, want, rtol=1e-13, atol=1e-13)              # same stream, filter applied
    other = dev.sample_noise(Pn, cov, [1.0, 0.0, 0.0], 7, 4).cpu().numpy()
    assert np.abs(other - raw).max() > 1.0                                      # next step: fresh stream
    # sharding invariance: particles [1000, 1500) drawn as a shard equal the same rows of the full draw
    shard = dev.sample_noise(500, cov, [1.0, 0.0, 0.0], 7, 3, particle_offset=1000).cpu().numpy()
    np.testing.assert_array_equal(shard, raw[1000:1500])
    # the two kernels behind the sampler (one thread per element / one thread per sample vector) draw the same
    # numbers: a diagonal covariance through either gives identical bits
    import ctypes
    import torch
    from mjmpc_amd import _lib
    dcov = np.diag([0.5, 1.5, 0.8])
    a = dev.sample_noise(Pn, dcov, [1.0, 0.0, 0.0], 11, 0).clone()
    b = torch.empty_like(a)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(dev.lib.mjmpc_sample_noise(_lib.F64, vp(b), Pn, Hn, An, vp(dev._rec["chol"]), None, 11, 0, 0, None, 0,
                                          dev.stream()))
    assert torch.equal(a, b)


def test_graph_replay_matches_eager_steps(raw_arm):
    """The captured control iteration (hipGraph, fused filter / cost-to-go / update / shift) must walk
    through the same closed loop as the eager, kernel-by-kernel path: same Philox stream, same states."""
    import torch
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn

    def run(graph, steps=6):
        eng = ArmRolloutEngine(raw_arm, dtype="f64")
        c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=16, init_cov=1.0, base_action="repeat",
                 lam=0.05, num_particles=256, step_size=0.8, alpha=1, gamma=0.98, n_iters=2,
                 action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.1], seed=5,
                 noise_mode="device")
        c.rollout_fn = make_device_rollout_fn(eng)
        c.set_sim_state_fn = lambda s: None
        eng.set_env_state(dict(qp=np.array([0.1, 0.2, 0.0, -0.5, 0.0, -0.3, 0.0]), qv=np.zeros(7),
                               target_pos=np.array([0.2, -0.1, 0.2])))
        if graph:
            c.enable_graph(post_step=eng.step_state)
        acts = []
        for _ in range(steps):
            a, _ = c.optimize({})
            if not graph:
                eng.step_state(a)
            acts.append(a)
        torch.cuda.synchronize()
        return np.array(acts), c.mean_action.copy()

    a_e, m_e = run(False)
    a_g, m_g = run(True)
    # forward- vs reverse-order cost-to-go summation and the fused merge order differ in the last bits
    np.testing.assert_allclose(a_g, a_e, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(m_g, m_e, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("kind", ["cem_full", "cem_diag", "dmd_full", "dmd_static"])
def test_adapting_covariance_stays_on_the_device_and_graph_matches_eager(raw_arm, kind):
    """CEM / DMD-MPC with update_cov: the covariance is refit, grown (shift) and Cholesky-factored for the sampler
    on the GPU.  The captured iteration walks the same closed loop as the eager path, and the eager path agrees
    with a run whose covariance round-trips through the host every step (the reference's data flow)."""
    import torch
    from mjmpc_amd.control import CEM, DMDMPC
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn

    def make(eng):
        kw = dict(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=12, num_particles=512, n_iters=1,
                  action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=9,
                  noise_mode="device", gamma=0.99, base_action="null")
        if kind.startswith("cem"):
            return CEM(init_cov=0.6, elite_frac=0.1, step_size=0.7, beta=0.05,
                       cov_type="full" if kind == "cem_full" else "diagonal", **kw)
        return DMDMPC(init_cov=0.6, beta=0.05, lam=0.2, step_size=0.7, update_cov=kind == "dmd_full", cov_type="full", **kw)

    def run(mode, steps=5):
        eng = ArmRolloutEngine(raw_arm, dtype="f64")
        c = make(eng)
        c.rollout_fn = make_device_rollout_fn(eng)
        c.set_sim_state_fn = lambda s: None
        eng.set_env_state(dict(qp=np.array([0.1, 0.2, 0.0, -0.5, 0.0, -0.3, 0.0]), qv=np.zeros(7),
                               target_pos=np.array([0.2, -0.1, 0.2])))
        if mode == "graph":
            c.enable_graph(post_step=eng.step_state)
        acts, covs = [], []
        for _ in range(steps):
            if mode == "host_cov":
                c.cov_action = c.cov_action.copy()        # force the host round trip + re-upload every step
            a, _ = c.optimize({})
            if mode != "graph":
                eng.step_state(a)
            acts.append(a)
            covs.append(c.cov_action.copy())
        torch.cuda.synchronize()
        assert int(c.dev.chol_status.item()) == 0
        return np.array(acts), np.array(covs), c.mean_action.copy()

    a_e, c_e, m_e = run("eager")
    a_h, c_h, m_h = run("host_cov")
    a_g, c_g, m_g = run("graph")
    if kind == "dmd_static":                                          # (fixed covariance: the fused MPPI-style path)
        assert np.array_equal(c_e[-1], c_e[0])
    else:
        assert np.abs(c_e[-1] - c_e[0]).max() > 1e-3                  # the covariance really adapts
    if kind in ("cem_full", "dmd_full"):
        assert np.abs(c_e[-1] - np.diag(np.diag(c_e[-1]))).max() > 1e-4
    for a, c, m in ((a_h, c_h, m_h), (a_g, c_g, m_g)):
        np.testing.assert_allclose(a, a_e, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(c, c_e, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(m, m_e, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("P_,k_frac", [(1000, 0.1), (16384, 0.1), (777, 0.5), (64, 1.0), (20000, 0.1), (40000, 0.05)])
def test_cem_elite_selection_with_ties_negative_costs_and_large_populations(P_, k_frac):
    """The radix select behind CEM's elite set: heavy ties across the threshold (broken by particle index),
    negative and zero costs, populations up to and beyond the BASELINE CEM configuration (16 384): the three
    instantiations of the select kernel (16 or 32 keys per thread in registers, or streamed from memory)."""
    from mjmpc_amd.control import CEM
    rs = np.random.RandomState(P_)
    Hh, Aa = 6, 3
    costs = np.round(rs.randn(P_, Hh) * 2.0, 1)                # quantised: many exactly equal cost-to-go values
    costs[rs.rand(P_) < 0.05] = 0.0
    mean0 = 0.2 * rs.randn(Hh, Aa)
    actions = mean0[None] + rs.randn(P_, Hh, Aa)
    c = CEM(init_cov=1.0, base_action="null", elite_frac=k_frac, step_size=0.9, gamma=1.0, beta=0.0, cov_type="full",
            d_state=5, d_obs=6, d_action=Aa, horizon=Hh, num_particles=P_, n_iters=1, action_lows=-np.ones(Aa),
            action_highs=np.ones(Aa), seed=1)
    c.mean_action = mean0.copy()
    c._update_distribution(dict(costs=costs, actions=actions))
    k = int(P_ * k_frac)
    q0 = costs[:, ::-1].cumsum(axis=1)[:, -1]                  # gamma = 1: plain sum, in cost_to_go's order
    ids = np.argsort(q0, kind="stable")[:k]                    # (q0, index) order
    assert np.sum(q0 == np.sort(q0)[k - 1]) > 1 or P_ == 64    # the threshold value really is tied
    d = (actions - mean0[None])[ids].reshape(Hh * k, Aa)
    want_cov = 0.1 * np.eye(Aa) + 0.9 * np.cov(d, rowvar=False)
    want_mean = 0.1 * mean0 + 0.9 * actions[ids].mean(axis=0)
    np.testing.assert_allclose(c.mean_action, want_mean, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(c.cov_action, want_cov, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("case", ["all_equal", "threshold_in_a_large_tie", "everything_elite", "two_values"])
def test_cem_elite_selection_degenerate_populations(case):
    """The select kernel's corner paths: every key equal (no byte differs: the passes are skipped and the tie walk names
    the cut), more ties at the threshold than the direct ranking takes (> 256), k = P, and keys that differ in one
    low byte only."""
    from mjmpc_amd.control import CEM
    rs = np.random.RandomState(11)
    P_, Hh, Aa = 5000, 4, 3
    costs = np.zeros((P_, Hh))
    frac = 0.3
    if case == "all_equal":
        costs[:] = 1.25
    elif case == "threshold_in_a_large_tie":
        costs[:, 0] = np.where(rs.rand(P_) < 0.6, 0.0, rs.randn(P_) ** 2 + 0.5)       # 60 % exact zeros, k = 30 %
    elif case == "everything_elite":
        costs = rs.randn(P_, Hh)
        frac = 1.0
    else:
        costs[:, 0] = np.where(rs.rand(P_) < 0.5, 1.0, np.nextafter(1.0, 2.0))
    mean0 = 0.2 * rs.randn(Hh, Aa)
    actions = mean0[None] + rs.randn(P_, Hh, Aa)
    c = CEM(init_cov=1.0, base_action="null", elite_frac=frac, step_size=1.0, gamma=1.0, beta=0.0, cov_type="full",
            d_state=5, d_obs=6, d_action=Aa, horizon=Hh, num_particles=P_, n_iters=1, action_lows=-np.ones(Aa),
            action_highs=np.ones(Aa), seed=1)
    c.mean_action = mean0.copy()
    c._update_distribution(dict(costs=costs, actions=actions))
    k = int(P_ * frac)
    q0 = costs[:, ::-1].cumsum(axis=1)[:, -1]
    ids = np.argsort(q0, kind="stable")[:k]
    d = (actions - mean0[None])[ids].reshape(Hh * k, Aa)
    np.testing.assert_allclose(c.mean_action, actions[ids].mean(axis=0), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(c.cov_action, np.cov(d, rowvar=False), rtol=1e-11, atol=1e-12)


def test_mppiq_returns_update_and_value(golden):
    """MPPIQ (mppiq.py:73-165) against the reference's outputs: TD(lambda) returns kernel, update, value."""
    from mjmpc_amd.control import MPPIQ
    g = golden("mppiq")
    for i in range(int(g["n"])):
        t = "q%d" % i
        beta, alpha, tbw, gamma, td_lam, step, c0, with_q = g[t + "_cfg"]
        Pq, Hq, Aq = g[t + "_actions"].shape
        c = MPPIQ(init_cov=c0, base_action="null", beta=beta, step_size=step, alpha=int(alpha), gamma=gamma, n_iters=1,
                  td_lam=td_lam, time_based_weights=bool(tbw), filter_coeffs=[1.0, 0.0, 0.0], d_state=5, d_obs=6,
                  d_action=Aq, horizon=Hq, num_particles=Pq, action_lows=-np.ones(Aq), action_highs=np.ones(Aq), seed=3)
        traj = dict(costs=g[t + "_costs"], actions=g[t + "_actions"])
        if with_q:
            traj["qvals"] = g[t + "_qvals"]
        c.mean_action = g[t + "_mean0"].copy()
        c._sync_in()
        np.testing.assert_allclose(c._returns(traj).cpu().numpy(), g[t + "_returns"], **TOL)
        np.testing.assert_allclose(c._calc_val(traj), g[t + "_val"], **TOL)
        np.testing.assert_array_equal(c.mean_action, g[t + "_mean0"])
        c._update_distribution(traj)
        np.testing.assert_allclose(c.mean_action, g[t + "_mean1"], **TOL)


def test_mppiq_matches_oracle_at_larger_size_and_f32():
    from mjmpc_amd.control import MPPIQ
    from oracle import controllers_ref as cr
    rs = np.random.RandomState(8)
    Pq, Hq, Aq = 1000, 32, 7
    mean0 = 0.2 * rs.randn(Hq, Aq)
    actions = mean0[None] + 0.5 * rs.randn(Pq, Hq, Aq)
    costs = rs.rand(Pq, Hq) * 2
    qvals = rs.rand(Pq, Hq) * 3
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        c = MPPIQ(init_cov=0.9, base_action="null", beta=0.4, step_size=0.8, alpha=0, gamma=0.97, n_iters=1, td_lam=0.85,
                  time_based_weights=True, d_state=5, d_obs=6, d_action=Aq, horizon=Hq, num_particles=Pq,
                  action_lows=-np.ones(Aq), action_highs=np.ones(Aq), seed=3)
        c.mean_action = mean0.copy()
        c._update_distribution(dict(costs=costs.astype(dtype), actions=actions.astype(dtype), qvals=qvals.astype(dtype)))
        want = cr.mppiq_update(costs, actions, qvals, mean0, 0.9 * np.eye(Aq), 0.4, 0, 0.97, 0.85, 0.8, True)
        np.testing.assert_allclose(c.mean_action, want, rtol=tol, atol=tol)


def test_clgaussian_mpc_closed_loop_rollouts(golden):
    """CLGaussianMPC.generate_rollouts / _get_next_action (clgaussian_mpc.py:63-116) over the pendulum engine:
    the policy rollouts reproduce the reference wrapper's closed_loop_linear vectors."""
    from mjmpc_amd.control import CLGaussianMPC
    from mjmpc_amd.envs.analytic_engine import AnalyticRolloutEngine
    g = golden("closed_loop")

    class Probe(CLGaussianMPC):
        def _update_distribution(self, trajectories):
            pass

    eng = AnalyticRolloutEngine.pendulum()
    Pn, Hn, _ = g["pend_noise"].shape

    def rollout_fn(num_particles, horizon, mean, noise, mode):
        obs, rew, act, done, info, nobs = eng.rollout(num_particles, horizon, mean, noise, mode)
        return dict(observations=obs, actions=act, costs=-rew, dones=done, next_observations=nobs)

    c = Probe(d_state=2, d_obs=3, d_action=1, action_lows=-2 * np.ones(1), action_highs=2 * np.ones(1), horizon=Hn,
              init_cov=0.25, init_mean=g["pend_W"].copy(), num_particles=Pn, gamma=1.0, n_iters=1,
              filter_coeffs=[1.0, 0.0, 0.0], set_sim_state_fn=eng.set_env_state, rollout_fn=rollout_fn, seed=5)
    c.sample_noise = lambda: g["pend_noise"].copy()
    traj = c.generate_rollouts({"state": g["pend_state"]})
    np.testing.assert_allclose(traj["observations"], g["pend_obs"], **TOL)
    np.testing.assert_allclose(traj["actions"], g["pend_act"], **TOL)
    np.testing.assert_allclose(traj["costs"], -g["pend_rew"], **TOL)
    np.testing.assert_allclose(traj["next_observations"], g["pend_nobs"], **TOL)
    a, _ = c.optimize({"state": g["pend_state"]})
    s = g["pend_state"]
    want = g["pend_W"].T @ np.array([np.cos(s[0]), np.sin(s[0]), s[1], 1.0])
    np.testing.assert_allclose(a, want, **TOL)
    with pytest.raises(ValueError):
        eng.rollout(Pn, Hn, g["pend_W"], g["pend_noise"], "closed_loop")


def test_capture_failure_falls_back_to_eager(raw_arm, monkeypatch):
    """controller.py, _optimize_graphed: if the runtime refuses to capture the control iteration the controller
    warns once and runs the same iteration eagerly - the closed loop must walk through the same actions."""
    import torch
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn

    def run(break_capture, steps=5):
        eng = ArmRolloutEngine(raw_arm, dtype="f64")
        c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=16, init_cov=1.0, base_action="null",
                 lam=0.05, num_particles=256, step_size=1.0, alpha=1, gamma=1.0, n_iters=1,
                 action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=5,
                 noise_mode="device")
        c.rollout_fn = make_device_rollout_fn(eng)
        c.set_sim_state_fn = lambda s: None
        eng.set_env_state(dict(qp=np.array([0.1, 0.2, 0.0, -0.5, 0.0, -0.3, 0.0]), qv=np.zeros(7),
                               target_pos=np.array([0.2, -0.1, 0.2])))
        c.enable_graph(post_step=eng.step_state, mono=False)     # (the fused iteration is launched directly: nothing to capture)
        if break_capture:
            class Refuse:
                def __init__(self, *a, **k):
                    raise RuntimeError("capture refused (test)")
            monkeypatch.setattr(torch.cuda, "graph", Refuse)
        acts = []
        with pytest.warns(UserWarning, match="capture") if break_capture else _nullcontext():
            for _ in range(steps):
                a, _ = c.optimize({})
                acts.append(a)
        torch.cuda.synchronize()
        assert bool(getattr(c, "graph_fallback", False)) == break_capture
        return np.array(acts), eng.get_env_state()[0]

    class _nullcontext:
        def __enter__(self):
            return None

        def __exit__(self, *a):
            return False

    a_graph, _ = run(False)
    a_eager, _ = run(True)
    np.testing.assert_allclose(a_eager, a_graph, rtol=0, atol=1e-9)


def test_semidefinite_covariance_samples_and_indefinite_raises():
    """The device-resident covariance path factors cov on the GPU.  numpy's multivariate_normal (the reference's
    sampler) accepts a positive SEMI-definite covariance; so does the kernel (zero column on a vanished pivot).
    An indefinite covariance has no factor: the flag reaches the host and optimize() would raise (check_status)."""
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.control._device import DeviceUpdater
    Pn, Hn, An = 20000, 4, 3
    dev = DeviceUpdater(Hn, An, np.ones(Hn))
    B = np.array([[1.0, 0.0], [0.5, 1.0], [-1.0, 2.0]])
    cov = B @ B.T                                            # rank 2
    dev.cov.copy_(torch.from_numpy(cov))
    x = dev.sample_noise(Pn, None, [1.0, 0.0, 0.0], 3, 0).cpu().numpy().reshape(-1, An)
    assert np.isfinite(x).all()
    np.testing.assert_allclose(np.cov(x, rowvar=False), cov, rtol=0.05, atol=0.05)
    L = dev._rec["chol"].cpu().numpy().reshape(An, An)
    np.testing.assert_allclose(L @ L.T, cov, rtol=1e-12, atol=1e-12)
    dev.check_status()                                       # nothing flagged
    dev.cov.copy_(torch.from_numpy(np.diag([1.0, -0.5, 1.0])))
    dev.sample_noise(Pn, None, [1.0, 0.0, 0.0], 3, 0)
    torch.cuda.synchronize()
    with pytest.raises(_lib.MjmpcError, match="indefinite"):
        dev.check_status()
    dev.check_status()                                       # the flag is cleared once reported


def test_diverged_rollouts_do_not_poison_the_update():
    """A rollout whose simulation diverged numerically carries NaN / inf costs (MuJoCo would have reset it): the
    MPPI and CEM updates treat its return as +inf - zero weight, never elite - i.e. they equal the update over the
    finite particles alone, instead of a NaN mean."""
    from mjmpc_amd.control._device import DeviceUpdater
    from oracle import controllers_ref as cr
    P, H, A = 256, 8, 3
    rs = np.random.RandomState(4)
    costs, actions = rs.rand(P, H) * 3.0, rs.randn(P, H, A)
    mean0, cov0, gseq = 0.1 * rs.randn(H, A), np.eye(A), cr.gamma_seq(0.97, H)
    bad = np.array([3, 77, 200, 255])
    dirty = costs.copy()
    dirty[3, 2] = np.nan
    dirty[77, 5] = np.inf
    dirty[200, :] = np.nan
    dirty[255, 0] = 1e308
    dirty[255, 1] = 1e308          # overflows in the cost-to-go
    keep = np.setdiff1d(np.arange(P), bad)
    dev = DeviceUpdater(H, A, gseq)
    dev.set_mean(mean0)
    dev.softmax_update(dirty, actions, 0.7, 0.9)
    got = dev.mean.cpu().numpy()
    want = cr.mppi_update(costs[keep], actions[keep], mean0, cov0, gseq, 0.7, 1, 0.9)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    # CEM: the elite are the best of the finite ones
    dev2 = DeviceUpdater(H, A, gseq)
    dev2.set_mean(mean0)
    dev2.set_cov(cov0)
    k = 25
    dev2.cem_update(dirty, actions, k, 0.8, False)
    q0 = cr.cost_to_go(costs.copy(), gseq)[:, 0]
    q0[bad] = np.inf
    ids = np.argsort(q0, kind="stable")[:k]
    assert not set(ids) & set(bad)
    want_mean = 0.2 * mean0 + 0.8 * actions[ids].mean(0)
    assert np.isfinite(dev2.mean.cpu().numpy()).all()
    np.testing.assert_allclose(dev2.mean.cpu().numpy(), want_mean, rtol=0, atol=1e-12)


@pytest.mark.parametrize("mode,grow", [(0, None), (1, "diag"), (1, "identity"), (0, "diag")])
def test_step_tail_is_action_shift_counter_and_covariance_growth(mode, grow):
    """``mjmpc_step_tail`` = the end of a device-resident control step in one launch: action <- mean[0] (device copy
    and pinned host copy), the shift of olgaussian_mpc.py:116-129, num_steps + 1, and the covariance growth of
    cem.py:94 / gaussian_dmd.py:111 - against the separate entries and numpy."""
    import torch
    from mjmpc_amd.control._device import DeviceUpdater
    rs = np.random.RandomState(5)
    Hn, An = 9, 5
    mean, cov, diag = rs.randn(Hn, An), np.cov(rs.randn(40, An), rowvar=False), rs.rand(An) + 0.1
    dev = DeviceUpdater(Hn, An, np.ones(Hn))
    dev.set_mean(mean)
    dev.set_cov(cov)
    act = torch.zeros(An, dtype=torch.float64, device="cuda")
    pin = torch.zeros(2 * (An + 1), dtype=torch.float64).pin_memory()
    step = torch.full((1,), 41, dtype=torch.int64, device="cuda")
    args = None if grow is None else ((diag if grow == "diag" else None), 0.3)
    dev.step_tail(mode, act, pin, step, args)
    torch.cuda.synchronize()
    want = np.vstack([mean[1:], np.zeros(An) if mode == 0 else mean[-1]])
    want_cov = cov if grow is None else cov + 0.3 * np.diag(diag if grow == "diag" else np.ones(An))
    np.testing.assert_array_equal(act.cpu().numpy(), mean[0])
    np.testing.assert_array_equal(pin[:An].numpy(), mean[0])
    assert pin[An].item() == 42.0               # the completion flag behind the action: the new step count
    np.testing.assert_array_equal(dev.get_mean(), want)
    np.testing.assert_allclose(dev.get_cov(), want_cov, rtol=0, atol=1e-15)
    assert int(step.item()) == 42
    # the separate entries leave the same mean and covariance behind
    dev2 = DeviceUpdater(Hn, An, np.ones(Hn))
    dev2.set_mean(mean)
    dev2.set_cov(cov)
    dev2.shift(mode)
    if args is not None:
        dev2.add_cov_diag(*args)
    np.testing.assert_array_equal(dev2.get_mean(), dev.get_mean())
    np.testing.assert_array_equal(dev2.get_cov(), dev.get_cov())


@pytest.mark.parametrize("P_,Hh,Aa,frac,cov_type", [(1000, 64, 24, 0.1, "full"), (300, 3, 40, 0.5, "full"),
                                                     (5000, 50, 7, 0.1, "full"), (777, 5, 2, 0.3, "diagonal"),
                                                     (4096, 32, 7, 0.02, "full"), (130, 7, 1, 0.9, "full")])
def test_cem_moments_over_the_elite_list_at_other_shapes(P_, Hh, Aa, frac, cov_type):
    """The elite-row moment kernels choose their layout by shape: the LDS-staged scatter when 16 x H x A deltas fit, the
    sliced one otherwise (wide actions, long horizons, A x A larger than a workgroup) - every branch against numpy."""
    from mjmpc_amd.control import CEM
    rs = np.random.RandomState(P_ + Aa)
    costs = rs.rand(P_, Hh)
    mean0 = 0.2 * rs.randn(Hh, Aa)
    actions = mean0[None] + rs.randn(P_, Hh, Aa)
    c = CEM(init_cov=1.0, base_action="null", elite_frac=frac, step_size=0.7, gamma=1.0, beta=0.0, cov_type=cov_type,
            d_state=5, d_obs=6, d_action=Aa, horizon=Hh, num_particles=P_, n_iters=1, action_lows=-np.ones(Aa),
            action_highs=np.ones(Aa), seed=1)
    c.mean_action = mean0.copy()
    c._update_distribution(dict(costs=costs, actions=actions))
    k = int(P_ * frac)
    q0 = costs[:, ::-1].cumsum(axis=1)[:, -1]
    ids = np.argsort(q0, kind="stable")[:k]
    d = (actions - mean0[None])[ids].reshape(Hh * k, Aa)
    upd = np.atleast_2d(np.cov(d, rowvar=False)) if cov_type == "full" else np.diag(np.var(d, axis=0))
    np.testing.assert_allclose(c.mean_action, 0.3 * mean0 + 0.7 * actions[ids].mean(axis=0), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(c.cov_action, 0.3 * np.eye(Aa) + 0.7 * upd, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_full_covariance_sampler_staged_and_direct_stores_agree(dtype):
    """``noise_full_kernel`` writes through an LDS tile when H is a multiple of 4 and directly otherwise: the draws are
    keyed by (particle, channel, t // 4), so the first steps of an H = 8 and an H = 7 draw are the same numbers; a
    population that leaves the last workgroup partly idle; and the diagonal-covariance kernel draws them too."""
    from mjmpc_amd.control._device import DeviceUpdater
    An, Pn = 5, 37
    rs = np.random.RandomState(2)
    B = rs.randn(An, An)
    cov = B @ B.T + 0.5 * np.eye(An)
    x8 = DeviceUpdater(8, An, np.ones(8)).sample_noise(Pn, cov, [1.0, 0.0, 0.0], 9, 3, dtype=dtype).cpu().numpy()
    x7 = DeviceUpdater(7, An, np.ones(7)).sample_noise(Pn, cov, [1.0, 0.0, 0.0], 9, 3, dtype=dtype).cpu().numpy()
    assert np.isfinite(x8).all() and np.abs(x8).max() > 1.0
    np.testing.assert_array_equal(x8[:, :7], x7)
    # an isotropic covariance through the full-covariance kernel = the diagonal kernel's draws (same Philox keys)
    iso = 0.49 * np.eye(An)
    near = iso.copy()
    near[0, 1] = near[1, 0] = 1e-300                        # off-diagonal entry: takes the full-covariance kernel
    d8 = DeviceUpdater(8, An, np.ones(8)).sample_noise(Pn, iso, [1.0, 0.0, 0.0], 9, 3, dtype=dtype).cpu().numpy()
    f8 = DeviceUpdater(8, An, np.ones(8)).sample_noise(Pn, near, [1.0, 0.0, 0.0], 9, 3, dtype=dtype).cpu().numpy()
    np.testing.assert_allclose(f8, d8, rtol=1e-6 if dtype == "f32" else 1e-14, atol=0)

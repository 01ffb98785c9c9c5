"""GPU: the fused control iteration (``mjmpc_arm_mppi_step``: sampling + rollout + cost-to-go in one launch, MPPI update,
action, shift and the real env's step in a second - reference controller.py:207-257 + example_mpc.py:165-168) against
 (a) the FP64 oracle: rollouts of the same samples + ``mppi_update`` (costs rel 1e-9, mean abs 1e-9), and
 (b) the multi-launch iteration it replaces (sampler kernel, rollout, two update launches, env-step launch), whose parts
     are each held to the golden vectors / the oracle elsewhere: same closed-loop action sequence at 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctrl(eng, P, H, dtype="f64", lam=0.05, cls="mppi", base="null", seed=11, filt=(0.25, 0.8, 0.0), cov=0.6):
    from mjmpc_amd.control import DMDMPC, MPPI
    kw = dict(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, init_cov=cov, base_action=base, lam=lam,
              num_particles=P, step_size=0.9, gamma=0.98, n_iters=1, action_lows=eng.action_lows,
              action_highs=eng.action_highs, filter_coeffs=list(filt), seed=seed, noise_mode="device", noise_dtype=dtype)
    if cls == "mppi":
        return MPPI(alpha=1, **kw)
    return DMDMPC(beta=0.1, update_cov=False, cov_type="diagonal", **kw)


def _engine(dtype="f64"):
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dtype)
    eng.set_env_state(dict(qp=np.array([0.1, 0.3, -0.2, -0.5, 0.2, -0.3, 0.1]), qv=np.zeros(7),
                           target_pos=np.array([0.1, 0.1, 0.1])))
    return eng


def _closed_loop(P, H, steps, dtype="f64", cls="mppi", base="null", filt=(0.25, 0.8, 0.0), **graph_kw):
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    eng = _engine(dtype)
    c = _ctrl(eng, P, H, dtype, cls=cls, base=base, filt=filt)
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c.enable_graph(post_step=eng.step_state, **graph_kw)
    acts = np.array([c.optimize({})[0] for _ in range(steps)])
    import torch
    torch.cuda.synchronize()
    _, nobs = eng.step_state(np.zeros(7))
    return acts, c.mean_action.copy(), nobs.cpu().numpy(), c, eng


@pytest.mark.parametrize("P,dtype", [(512, "f64"), (4096, "f64"), (1000, "f32"), (3000, "f32")])
def test_one_launch_iteration_equals_the_multi_launch_iteration(P, dtype):
    H, steps = 12, 6
    a1, m1, o1, c1, e1 = _closed_loop(P, H, steps, dtype, mono=True)
    assert c1._mono and not getattr(c1, "graph_fallback", False)
    a0, m0, o0, c0, e0 = _closed_loop(P, H, steps, dtype, mono=False)
    assert not c0._mono
    tol = 1e-9 if dtype == "f64" else 2e-3      # (f32: the two kernels may contract FMAs differently; the closed loop amplifies it)
    np.testing.assert_allclose(a1[0], a0[0], rtol=0, atol=1e-9 if dtype == "f64" else 2e-5)     # the first iteration alone
    np.testing.assert_allclose(a1, a0, rtol=0, atol=tol)
    np.testing.assert_allclose(m1, m0, rtol=0, atol=tol)
    np.testing.assert_allclose(o1, o0, rtol=0, atol=tol * 10)       # the real arm ended up in the same place
    assert e1.solver_failures() == 0 and c1.num_steps == steps


@pytest.mark.parametrize("cls,base,filt", [("dmd", "repeat", (0.25, 0.8, 0.0)), ("mppi", "repeat", (1.0, 0.0, 0.0)),
                                           ("mppi", "null", (0.5, 0.3, 0.2))])
def test_one_launch_iteration_options(cls, base, filt):
    """DMD-MPC without covariance adaptation shares the launch; 'repeat' shift; no filter / a three-tap filter."""
    a1, m1, o1, c1, _ = _closed_loop(640, 10, 5, cls=cls, base=base, filt=filt, mono=True)
    a0, m0, o0, c0, _ = _closed_loop(640, 10, 5, cls=cls, base=base, filt=filt, mono=False)
    assert c1._mono
    np.testing.assert_allclose(a1, a0, rtol=0, atol=1e-9)
    np.testing.assert_allclose(m1, m0, rtol=0, atol=1e-9)


def test_large_populations_keep_the_separate_launches():
    """Above one wavefront per SIMD pair the controller does not take the fused iteration (and still runs)."""
    a, m, o, c, e = _closed_loop(8200, 8, 3, mono=True)
    assert not c._mono and np.isfinite(a).all() and e.solver_failures() == 0
    # the entry point itself still handles any population (one-wave-per-group instantiations)
    import torch
    eng = _engine()
    c2 = _ctrl(eng, 20000, 8)
    chol, coeffs, _ = c2.dev.prepare_noise(c2.cov_action, c2.filter_coeffs)
    step_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    eng.mppi_step(20000, 8, c2.dev.mean, c2.dev.mean_alt, c2.dev.gseq, coeffs, chol, 5, 0, 0, step_dev, 0.05, 1.0, 0)
    c3 = _ctrl(eng, 20000, 8)
    noise = c3.dev.sample_noise(20000, c3.cov_action, c3.filter_coeffs, 5, 0, filtered=False)
    costs, acts, q0 = eng.rollout_fused(20000, 8, c3.dev.mean, noise, coeffs, c3.dev.gseq)
    c3.dev.mppi_fused_update(q0, acts, 0.05, 1.0, 0, None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(c2.dev.mean_alt.cpu().numpy(), c3.dev.get_mean(), rtol=0, atol=1e-9)


def test_lookahead_returns_the_same_actions():
    """Iteration k + 1 enqueued before the host waits for action k: same closed loop, one iteration more in flight."""
    a1, m1, o1, c1, _ = _closed_loop(512, 12, 9, mono=True, lookahead=True)
    a0, m0, o0, c0, _ = _closed_loop(512, 12, 9, mono=True, lookahead=False)
    np.testing.assert_array_equal(a1, a0)
    assert c1._ahead == 1 and c0._ahead == 0
    c1.reset()
    assert c1._ahead == 0


@pytest.mark.parametrize("P", [96, 4096])
def test_one_launch_iteration_against_the_oracle(P):
    """Trajectories and update of ONE launch vs the oracle run on the same samples (drawn by the sampler kernel, whose
    stream the launch reproduces sample for sample)."""
    import torch
    from oracle import controllers_ref as cr
    from oracle.physics_ref import RefArm
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = _engine()
    H, A, lam, step = 16, 7, 1.0, 0.9
    c = _ctrl(eng, P, H, lam=lam)
    mean0 = 0.1 * np.random.RandomState(4).standard_normal((H, A))
    c.mean_action = mean0.copy()
    c._sync_in()
    st = eng.get_env_state()[0]
    raw_noise = c.dev.sample_noise(P, c.cov_action, c.filter_coeffs, c.seed_val, 0, filtered=True).cpu().numpy()
    chol, coeffs, diag = c.dev.prepare_noise(c.cov_action, c.filter_coeffs)
    assert diag == 1
    step_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    act_dev = torch.zeros(A, dtype=torch.float64, device="cuda")
    slots = torch.zeros(2 * (A + 1), dtype=torch.float64).pin_memory()
    costs, acts, q0 = eng.mppi_step(P, H, c.dev.mean, c.dev.mean_alt, c.dev.gseq, coeffs, chol, c.seed_val, 0, 0, step_dev, lam,
                                    step, 0, action_out=act_dev, action_slots=slots, env_step=True, want_trajectories=True)
    torch.cuda.synchronize()
    ref = RefArm(reacher7dof_raw().to_flat())
    _, rew, o_act, _, _ = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean0, raw_noise)
    np.testing.assert_allclose(acts.cpu().numpy(), o_act, rtol=0, atol=1e-15)       # mean + the same filtered samples
    np.testing.assert_allclose(costs.cpu().numpy(), -rew, rtol=1e-9, atol=1e-9)
    gs = cr.gamma_seq(0.98, H)
    np.testing.assert_allclose(q0.cpu().numpy(), cr.cost_to_go(-rew, gs)[:, 0], rtol=1e-9)
    new_mean = cr.mppi_update(-rew, o_act, mean0, np.eye(A), gs, lam, 1, step)
    action = slots.numpy()[:A].copy()
    assert slots.numpy()[A] == 1.0 and int(step_dev.item()) == 1                    # flag = new step count, slot 0
    np.testing.assert_allclose(action, new_mean[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(act_dev.cpu().numpy(), new_mean[0], rtol=0, atol=1e-9)
    shifted = np.vstack([new_mean[1:], np.zeros((1, A))])
    np.testing.assert_allclose(c.dev.mean_alt.cpu().numpy(), shifted, rtol=0, atol=1e-9)       # (written to the other buffer)
    np.testing.assert_allclose(c.dev.get_mean(), mean0, rtol=0, atol=0)                           # (the input is only read)
    # the real arm took one env step with that action
    q1, v1, _, _ = ref.env_step(st["qp"], st["qv"], new_mean[0], st["target_pos"])
    _, nobs = eng.step_state(np.zeros(A))           # (reads the state back through a zero-action step's observation ...)
    q2, v2, _, _ = ref.env_step(q1, v1, np.zeros(A), st["target_pos"])
    np.testing.assert_allclose(nobs.cpu().numpy()[:7], q2, rtol=0, atol=1e-9)
    np.testing.assert_allclose(nobs.cpu().numpy()[7:14], v2, rtol=0, atol=1e-8)
    assert eng.solver_failures() == 0


def test_sharded_record_of_the_launch():
    """N > 1: the launch leaves this GPU's softmax record {max, S, W} and touches nothing else; two half-populations'
    records combined = the update of the whole population (particle_offset keys the samples globally)."""
    import torch
    eng = _engine()
    P, H, A, lam = 1024, 8, 7, 0.2
    c = _ctrl(eng, P, H, lam=lam)
    chol, coeffs, _ = c.dev.prepare_noise(c.cov_action, c.filter_coeffs)
    mean_before = c.dev.mean.clone()
    step_dev = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    recs = []
    for r in range(2):
        rec = torch.zeros(2 + H * A, dtype=torch.float64, device="cuda")
        eng.mppi_step(P // 2, H, c.dev.mean, None, c.dev.gseq, coeffs, chol, c.seed_val, 0, r * (P // 2), step_dev, lam, 1.0,
                      0, record=rec, env_step=True)
        recs.append(rec)
    torch.cuda.synchronize()
    assert torch.equal(c.dev.mean, mean_before) and int(step_dev.item()) == 3       # untouched
    whole = torch.zeros(2 + H * A, dtype=torch.float64, device="cuda")
    eng.mppi_step(P, H, c.dev.mean, None, c.dev.gseq, coeffs, chol, c.seed_val, 0, 0, step_dev, lam, 1.0, 0, record=whole)
    torch.cuda.synchronize()
    r0, r1, w = (x.cpu().numpy() for x in (recs[0], recs[1], whole))
    m = max(r0[0], r1[0])
    assert m == w[0]
    s = np.exp(r0[0] - m) * r0[1] + np.exp(r1[0] - m) * r1[1]
    W = np.exp(r0[0] - m) * r0[2:] + np.exp(r1[0] - m) * r1[2:]
    np.testing.assert_allclose(s, w[1], rtol=1e-12)
    np.testing.assert_allclose(W, w[2:], rtol=1e-10, atol=1e-13)


def test_combine_rejects_bad_arguments():
    """``mjmpc_arm_mppi_combine`` (the launch behind the record all-gather of a sharded run): argument checks."""
    import ctypes
    import torch
    from mjmpc_amd import _lib
    eng = _engine()
    lib = _lib.require_gpu()
    H, A = 8, 7
    recs = torch.zeros(2, 2 + H * A, dtype=torch.float64, device="cuda")
    mean = torch.zeros(H, A, dtype=torch.float64, device="cuda")
    alt = torch.zeros_like(mean)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())

    def call(mean_out, n_rec=2, shift=0):
        return lib.mjmpc_arm_mppi_combine(eng._h, eng._code, vp(recs), n_rec, H, vp(mean), vp(mean_out), vp(step), 1.0, shift,
                                          None, None, 0, None, None, None)

    assert call(alt) == 0
    torch.cuda.synchronize()
    for bad in (call(mean), call(alt, n_rec=0), call(alt, shift=2)):       # aliased mean, no records, unknown shift mode
        assert bad != 0
        assert lib.mjmpc_last_error()

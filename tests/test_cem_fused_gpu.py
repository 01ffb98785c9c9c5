"""GPU: the fused CEM step (round 4): ``mjmpc_cem_select_moments`` (selection + elite list + moments in
one launch) and ``mjmpc_cem_finish`` (refit, covariance growth, Cholesky factor, action, shift, step counter and the next
step's raw samples in one launch) - reference cem.py:65-95 - against
 (a) the separate launches they replace (which tests/test_controllers_gpu.py holds to the golden vectors at 1e-12): same
     closed loop; the moments are the shifted-data form of the two-pass np.cov (scatter about a provisional centre, then
     - N (mu - c)(mu - c)'), equal to it to rounding;
 (b) the oracle: rollouts + numpy ``cem_update`` at BASELINE config 4's size on one GPU, 16384 x 32, full covariance;
 (c) the sampler kernel: the samples the finish launch draws for the NEXT step are ``mjmpc_sample_noise``'s, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FILT = [0.25, 0.8, 0.0]
MOVING = dict(qp=np.array([0.3, 0.5, -0.2, -1.0, 0.4, -0.6, 0.2]), qv=np.array([0.5, -1.0, 0.3, 2.0, -0.5, 1.0, 0.1]),
              qa=np.zeros(7), target_pos=np.array([-0.25, 0.15, 0.2]), timestep=0)


def _cem(eng, P, H, cov_type, fused, elite_frac=0.1, seed=7, step_size=0.8, beta=0.02, base="null", in_kernel=True):
    from mjmpc_amd.control import CEM
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    c = CEM(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, init_cov=0.5, base_action=base,
            elite_frac=elite_frac, num_particles=P, step_size=step_size, gamma=1.0, n_iters=1, beta=beta, cov_type=cov_type,
            action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=FILT, seed=seed, noise_mode="device",
            noise_dtype=eng.dtype)
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c._want_cem_fused = fused
    c._want_cem_in_kernel = in_kernel
    c.enable_graph(post_step=eng.step_state)
    return c


def _loop(P, H, cov_type, fused, steps, dtype="f64", **kw):
    import torch
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dtype)
    eng.set_env_state(MOVING)
    c = _cem(eng, P, H, cov_type, fused, **kw)
    acts = np.array([c.optimize({})[0] for _ in range(steps)])
    torch.cuda.synchronize()
    assert c._cem_fused() == fused and not getattr(c, "graph_fallback", False)
    return acts, c.mean_action.copy(), c.cov_action.copy(), eng, c


@pytest.mark.parametrize("P,H,cov_type,kw", [(2048, 12, "full", {}), (2048, 12, "diagonal", {}), (4096, 32, "full", dict(base="repeat")),
                                              (20000, 8, "full", dict(elite_frac=0.05)), (1000, 10, "full", dict(elite_frac=0.003))])
def test_fused_step_equals_the_separate_launches(P, H, cov_type, kw):
    a1, m1, c1, e1, _ = _loop(P, H, cov_type, True, 6, **kw)
    a0, m0, c0, e0, _ = _loop(P, H, cov_type, False, 6, **kw)
    np.testing.assert_allclose(a1[0], a0[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a1, a0, rtol=0, atol=1e-9)
    np.testing.assert_allclose(m1, m0, rtol=0, atol=1e-9)
    np.testing.assert_allclose(c1, c0, rtol=1e-8, atol=1e-12)
    assert e1.solver_failures() == 0


def test_fused_step_against_the_oracle_16384x32(ref_arm):
    """BASELINE config 4 on one GPU: one whole captured control step (rollout launch + two CEM launches + env step)
    against oracle rollouts + numpy cem_update + shift on the same Philox samples; then the samples of step 2."""
    import torch
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    from oracle import controllers_ref as cr
    P, H, A = 16384, 32, 7
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    eng.set_env_state(MOVING)
    c = _cem(eng, P, H, "full", True)
    action, _ = c.optimize({})
    torch.cuda.synchronize()
    assert c._cem_fused()
    next_raw = c.dev._rec[("noise", "f64")].clone()
    mean0, cov0 = np.zeros((H, A)), 0.5 * np.eye(A)
    noise = c.dev.sample_noise(P, cov0, FILT, 7, 0, filtered=True).cpu().numpy()
    _, rew, act, _, _ = ref_arm.rollout(MOVING["qp"], MOVING["qv"], MOVING["target_pos"], mean0, noise, want_obs=False)
    mean1, cov1 = cr.cem_update(-rew, act, mean0, cov0, cr.gamma_seq(1.0, H), 0.1, 0.8, "full")
    np.testing.assert_allclose(action, mean1[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(c.mean_action, cr.shift_mean(mean1, "null"), rtol=0, atol=1e-9)
    cov2 = cr.cem_shift_cov(cov1, 0.02, 0.5 * np.ones(A))
    np.testing.assert_allclose(c.cov_action, cov2, rtol=1e-9, atol=1e-12)
    # the finish launch drew step 2's raw samples with the new covariance's factor: the sampler kernel's stream
    want = c.dev.sample_noise(P, None, FILT, 7, 1, filtered=False)          # (factor of the device-resident covariance)
    torch.cuda.synchronize()
    assert torch.equal(next_raw, want)
    assert eng.solver_failures() == 0


def test_fused_step_degenerate_populations():
    """Every cost equal (the elite set is the first k particles by index), and costs that differ in their last bit."""
    import torch
    from mjmpc_amd.control._device import DeviceUpdater
    P, H, A, k = 3000, 6, 5, 37
    rs = np.random.RandomState(0)
    for q0 in (np.full(P, 2.5), 1.0 + rs.randint(0, 3, P) * 2.0 ** -52):
        dev = DeviceUpdater(H, A, np.ones(H))
        assert dev.cem_fused_supported(P, k)
        actions = torch.from_numpy(rs.standard_normal((P, H, A))).cuda()
        mean0, cov0 = 0.1 * rs.standard_normal((H, A)), np.diag(rs.uniform(0.5, 1.0, A))
        dev.set_mean(mean0)
        dev.set_cov(cov0)
        ws = dev.workspace(P)
        dev._q0_view(ws, P).copy_(torch.from_numpy(q0).cuda())
        step = torch.full((1,), 4, dtype=torch.int64, device="cuda")
        act = torch.zeros(A, dtype=torch.float64, device="cuda")
        pin = torch.full((A + 1,), -1.0, dtype=torch.float64).pin_memory()
        dev.cem_fused_step(actions, k, 0.7, True, 1, act, pin, step, (None, 0.25), None, 3, 0)
        torch.cuda.synchronize()
        # mapped host copy: the action, then the new step count as the completion flag a captured loop polls
        np.testing.assert_array_equal(pin[:A].numpy(), act.cpu().numpy())
        assert pin[A].item() == 5.0
        ids = np.lexsort((np.arange(P), q0))[:k]                      # (q0, index) order: ties go to the smaller index
        el = actions.cpu().numpy()[ids]
        d = (el - mean0[None]).reshape(k * H, A)
        cov1 = 0.3 * cov0 + 0.7 * np.cov(d, rowvar=False) + 0.25 * np.eye(A)
        mean1 = 0.3 * mean0 + 0.7 * el.mean(0)
        np.testing.assert_allclose(act.cpu().numpy(), mean1[0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(dev.get_cov(), cov1, rtol=1e-10, atol=1e-13)
        shifted = np.vstack([mean1[1:], mean1[-1:]])
        np.testing.assert_allclose(dev.get_mean(), shifted, rtol=0, atol=1e-12)
        assert int(step.item()) == 5
        L = dev._rec["chol"].cpu().numpy().reshape(A, A)
        np.testing.assert_allclose(L @ L.T, cov1, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("P,H,cov_type,dtype", [(4096, 32, "full", "f64"), (1024, 12, "diagonal", "f64"), (2000, 16, "full", "f32")])
def test_rollout_that_draws_its_own_full_covariance_samples(P, H, cov_type, dtype):
    """At most one wavefront per SIMD pair (P <= 4096 on 256 CUs) the CEM step needs no sample buffer: the rollout launch
    colours its own Philox draws with the whole lower triangle of the factor the finish launch left (``mjmpc_arm_rollout_sampled``,
    chol_full) - the same closed loop as with the samples drawn into a buffer by the finish launch."""
    a1, m1, c1, e1, ctl1 = _loop(P, H, cov_type, True, 6, dtype=dtype, in_kernel=True)
    assert ctl1._cem_in_kernel()
    a0, m0, c0, e0, ctl0 = _loop(P, H, cov_type, True, 6, dtype=dtype, in_kernel=False)
    assert not ctl0._cem_in_kernel()
    tol = 1e-9 if dtype == "f64" else 2e-3
    np.testing.assert_allclose(a1[0], a0[0], rtol=0, atol=1e-12 if dtype == "f64" else 2e-5)
    np.testing.assert_allclose(a1, a0, rtol=0, atol=tol)
    np.testing.assert_allclose(c1, c0, rtol=1e-7 if dtype == "f64" else 1e-2, atol=1e-12)
    assert e1.solver_failures() == 0


@pytest.mark.parametrize("kind,P", [("cem_in_kernel", 2048), ("cem_buffer", 2048), ("cem_unfused", 2048), ("dmd_cov", 1024),
                                     ("mppi_static", 1024)])
def test_covariance_assigned_on_the_host_between_captured_steps(kind, P):
    """``cov_action`` assigned between two captured ``optimize()`` calls (the reference lets users edit it at any time) takes
    effect in the NEXT step's samples, as in the eager loop: the samples the previous finish launch drew ahead - or the
    factor it left for the rollout launch that colours its own draws - are made again from the uploaded covariance
    (they were not, so the rollout sampled the old covariance while the refit blended with the new one)."""
    import torch
    from mjmpc_amd.control import CEM, DMDMPC, MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    rs = np.random.RandomState(3)
    B = rs.standard_normal((7, 7))
    new_cov = 0.05 * B @ B.T + 0.2 * np.eye(7)
    if kind == "mppi_static":
        new_cov = np.diag(np.diag(new_cov))

    def run(graph):
        eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
        eng.set_env_state(MOVING)
        kw = dict(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=12, num_particles=P, n_iters=1, gamma=1.0,
                  action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=FILT, seed=11,
                  noise_mode="device", base_action="null")
        if kind.startswith("cem"):
            c = CEM(init_cov=0.5, elite_frac=0.1, step_size=0.8, beta=0.02, cov_type="full", **kw)
            c._want_cem_fused = kind != "cem_unfused"
            c._want_cem_in_kernel = kind == "cem_in_kernel"
        elif kind == "dmd_cov":
            c = DMDMPC(init_cov=0.5, beta=0.02, lam=0.2, step_size=0.8, update_cov=True, cov_type="full", **kw)
        else:
            c = MPPI(init_cov=0.5, lam=0.2, step_size=0.8, alpha=1, **kw)
        c.rollout_fn = make_device_rollout_fn(eng)
        c.set_sim_state_fn = lambda s: None
        if graph:
            c.enable_graph(post_step=eng.step_state)
        acts = []
        for k in range(6):
            if k == 3:
                c.cov_action = new_cov.copy()
            a, _ = c.optimize({})
            if not graph:
                eng.step_state(a)
            acts.append(a)
        torch.cuda.synchronize()
        assert not getattr(c, "graph_fallback", False)
        return np.array(acts), c.mean_action.copy(), c.cov_action.copy()

    a_g, m_g, c_g = run(True)
    a_e, m_e, c_e = run(False)
    np.testing.assert_allclose(a_g, a_e, rtol=0, atol=1e-9)
    np.testing.assert_allclose(m_g, m_e, rtol=0, atol=1e-9)
    np.testing.assert_allclose(c_g, c_e, rtol=1e-8, atol=1e-12)

"""GPU: the seed-identical sampler (MT19937 + legacy polar method regenerated on the device) against the
reference's noise itself - golden vectors captured from control_utils.generate_noise and live numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    """distance in units in the last place (same-sign neighbours; exact integer arithmetic)"""
    ia, ib = np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64)
    same_sign = (ia < 0) == (ib < 0)
    d = np.where(same_sign, np.abs(ia - ib), np.iinfo(np.int64).max)
    return d


def test_matches_the_reference_golden_noise(golden):
    from mjmpc_amd.control._device import DeviceUpdater
    g = golden("noise")
    for tag in "abcd":
        cov, coeffs, (P, H), seed = g[tag + "_cov"], g[tag + "_coeffs"], g[tag + "_shape"], int(g[tag + "_seed"])
        A = cov.shape[0]
        dev = DeviceUpdater(int(H), A, np.ones(int(H)))
        got = dev.sample_noise_mt19937(int(P), cov, list(coeffs), seed, 0).cpu().numpy()
        want = g[tag + "_eps"]
        assert got.shape == want.shape
        d = _ulp_diff(got, want)
        # same stream, same alignment: the raw samples are bit-identical up to libm's last bit, the filter
        # (contracted multiply-adds on the device) may move the last bits of the filtered ones
        np.testing.assert_allclose(got, want, rtol=4e-15, atol=4e-15)   # filtered values can cancel towards 0
        assert (d <= 1).mean() > 0.5
        assert int(dev._rec["mt_status"].item()) == 0


def test_full_size_stream_alignment_and_step_counter():
    """4096 x 32 x 7 (917 504 normals, ~2.3 M twister words): unfiltered stream vs numpy, and the device-side
    step counter selecting seed + step."""
    import torch
    from mjmpc_amd.control._device import DeviceUpdater
    P, H, A, seed = 4096, 32, 7, 123
    dev = DeviceUpdater(H, A, np.ones(H))
    for step in (0, 5):
        np.random.seed(seed + step)
        want = np.sqrt(1.7) * np.random.standard_normal((P, H, A))
        d_step = torch.full((1,), step, dtype=torch.int64, device="cuda")
        got = dev.sample_noise_mt19937(P, 1.7 * np.eye(A), [1.0, 0.0, 0.0], seed, 0, d_step=d_step).cpu().numpy()
        d = _ulp_diff(got, want)
        assert d.max() <= 4, d.max()                  # device log() vs glibc + the sqrt(c) scaling: last bits
        assert (d == 0).mean() > 0.95
    # a non-isotropic (here: diagonal) covariance goes through numpy's SVD colouring (test below), no longer refused
    from mjmpc_amd.control.control_utils import generate_noise
    cov = np.diag([1.0, 2, 1, 1, 1, 1, 1])
    got = dev.sample_noise_mt19937(8, cov, [1.0, 0.0, 0.0], 1, 0).cpu().numpy()
    np.testing.assert_allclose(got, generate_noise(cov, [1.0, 0.0, 0.0], (8, H), 1), rtol=1e-12, atol=1e-13)


def test_jump_ahead_segmentation_is_bit_identical_to_the_serial_stream():
    """The stream cut into 2 / 7 / 32 jumped-ahead segments equals the single-workgroup stream bit for bit
    (and therefore numpy's), including a short last segment."""
    from mjmpc_amd.control._device import DeviceUpdater
    P, H, A, seed = 1024, 32, 7, 77
    outs = {}
    for nseg in (0, 2, 7, 32):
        dev = DeviceUpdater(H, A, np.ones(H))
        dev.mt_segments = nseg
        outs[nseg] = dev.sample_noise_mt19937(P, np.eye(A), [1.0, 0.0, 0.0], seed, 3).cpu().numpy()
        assert dev._rec["mt_jump"][4] == nseg
        assert int(dev._rec["mt_status"].item()) == 0
    for nseg in (2, 7, 32):
        np.testing.assert_array_equal(outs[nseg], outs[0])
    np.random.seed(seed + 3)
    want = np.random.standard_normal((P, H, A))
    assert _ulp_diff(outs[0], want).max() <= 4


def test_device_resident_loop_reproduces_the_reference_stream_loop(raw_arm):
    """End to end: the captured, fully device-resident MPPI loop fed by the on-device MT19937 sampler walks
    the same closed loop as the SAME controller fed by the reference's host noise (control_utils.generate_noise,
    uploaded) - identical seeds, identical particles, identical actions (to rounding)."""
    import torch
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn, make_rollout_fn

    def run(noise_mode, graph, steps=5):
        eng = ArmRolloutEngine(raw_arm, dtype="f64")
        c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=16, init_cov=0.8, base_action="null",
                 lam=0.05, num_particles=512, step_size=0.9, alpha=1, gamma=0.99, n_iters=1,
                 action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=321,
                 noise_mode=noise_mode)
        c.rollout_fn = make_device_rollout_fn(eng) if graph else make_rollout_fn(eng)
        state = dict(qp=np.array([0.1, 0.2, 0.0, -0.5, 0.0, -0.3, 0.0]), qv=np.zeros(7), qa=np.zeros(7),
                     target_pos=np.array([0.2, -0.1, 0.2]), timestep=0)
        eng.set_env_state(state)
        c.set_sim_state_fn = (lambda s: None) if graph else eng.set_env_state
        if graph:
            c.enable_graph(post_step=eng.step_state)
        acts = []
        for _ in range(steps):
            a, _ = c.optimize(state)
            acts.append(a)
            if not graph:                          # host path: step the real env through the engine at P = 1
                _, nobs = eng.step_state(a)
                o = nobs.cpu().numpy()
                state = dict(state, qp=o[:7].copy(), qv=o[7:14].copy())
        torch.cuda.synchronize()
        return np.array(acts)

    a_host = run("host", False)
    a_dev = run("device_mt19937", True)
    np.testing.assert_allclose(a_dev, a_host, rtol=1e-8, atol=1e-9)


def test_sharded_block_of_the_stream():
    """particle_offset: a rank's block [offset, offset + P_local) of the ONE global stream equals the same rows of
    the full draw (and therefore numpy's), for unfiltered and filtered noise."""
    from mjmpc_amd.control._device import DeviceUpdater
    P, H, A, seed = 768, 16, 7, 4242
    full = DeviceUpdater(H, A, np.ones(H)).sample_noise_mt19937(P, 0.9 * np.eye(A), [0.25, 0.8, 0.0], seed, 2).cpu().numpy()
    for G in (2, 3):
        n = P // G
        for g in range(G):
            dev = DeviceUpdater(H, A, np.ones(H))
            mine = dev.sample_noise_mt19937(n, 0.9 * np.eye(A), [0.25, 0.8, 0.0], seed, 2, particle_offset=g * n)
            np.testing.assert_array_equal(mine.cpu().numpy(), full[g * n:(g + 1) * n])
            assert int(dev._rec["mt_status"].item()) == 0
    from mjmpc_amd.control.control_utils import generate_noise
    want = generate_noise(0.9 * np.eye(A), [0.25, 0.8, 0.0], (P, H), seed + 2)
    np.testing.assert_allclose(full, want, rtol=4e-15, atol=4e-15)


def test_general_covariance_matches_numpy_svd_colouring():
    """np.random.multivariate_normal colours its standard-normal stream with sqrt(s)[:, None] * v from svd(cov)
    (control_utils.py:30).  The device regenerates the stream and applies the same (host-computed) matrix in a fixed
    order: agreement to rounding of the 7-term products, not bit for bit (BLAS' dot has its own order)."""
    from mjmpc_amd.control._device import DeviceUpdater
    from mjmpc_amd.control.control_utils import generate_noise
    P, H, A, seed = 300, 12, 5, 77
    rs = np.random.RandomState(1)
    Bm = rs.randn(A, A)
    cov = Bm @ Bm.T + 0.2 * np.eye(A)
    dev = DeviceUpdater(H, A, np.ones(H))
    for coeffs in ([1.0, 0.0, 0.0], [0.25, 0.8, 0.1]):
        got = dev.sample_noise_mt19937(P, cov, coeffs, seed, 4).cpu().numpy()
        want = generate_noise(cov, coeffs, (P, H), seed + 4)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    assert int(dev._rec["mt_status"].item()) == 0
    # a sharded rank keeps its block of the one stream
    mine = dev.sample_noise_mt19937(100, cov, [1.0, 0.0, 0.0], seed, 4, particle_offset=100).cpu().numpy()
    np.testing.assert_allclose(mine, generate_noise(cov, [1.0, 0.0, 0.0], (P, H), seed + 4)[100:200], rtol=1e-12, atol=1e-13)

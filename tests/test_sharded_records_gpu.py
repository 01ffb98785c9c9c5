"""GPU: the multi-GPU record path on ONE GPU.  Particles are split into G contiguous shards, each
shard's record is produced by the HIP kernels through the C ABI exactly as a rank would, the
records are stacked (what the all-gather delivers) and combined by the HIP combine kernels with
G > 1.  Result must equal the unsharded reference update (oracle pinned on golden vectors)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import controllers_ref as cr  # noqa: E402


def _vp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def problem():
    rs = np.random.RandomState(7)
    P, H, A = 96, 12, 5
    mean = 0.3 * rs.randn(H, A)
    B = rs.randn(A, A)
    cov = B @ B.T / A + 0.5 * np.eye(A)
    actions = mean[None] + rs.randn(P, H, A)
    costs = rs.rand(P, H) * 2 + 0.1 * np.abs(actions).sum(-1)
    return dict(P=P, H=H, A=A, mean=mean, cov=cov, actions=actions, costs=costs, gs=cr.gamma_seq(0.97, H))


@pytest.mark.parametrize("G", [2, 3])
def test_softmax_records_combine(problem, G):
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.control._device import DeviceUpdater
    pr = problem
    P, H, A = pr["P"], pr["H"], pr["A"]
    lam, step = 0.08, 0.6
    n = P // G
    devs = [DeviceUpdater(H, A, pr["gs"]) for _ in range(G)]
    lib = devs[0].lib
    rlen = lib.mjmpc_softmax_record_len(H, A, 0)
    recs = torch.empty((G, rlen), dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        d.set_mean(pr["mean"])
        c = torch.from_numpy(pr["costs"][g * n:(g + 1) * n].copy()).cuda()
        a = torch.from_numpy(pr["actions"][g * n:(g + 1) * n].copy()).cuda()
        _lib.check(lib.mjmpc_softmax_stats(_lib.F64, n, H, A, _vp(c), _vp(a), _vp(d.mean), None, _vp(d.gseq), 0, lam, 1,
                                           0, 1, _vp(recs[g]), _vp(d.workspace(n)), d.stream()))
        want = cr.softmax_record(pr["costs"][g * n:(g + 1) * n], pr["actions"][g * n:(g + 1) * n], pr["mean"], pr["gs"],
                                 lam, want_cov=True)
        np.testing.assert_allclose(recs[g].cpu().numpy(), want, rtol=1e-12, atol=1e-12)
    d0 = devs[0]
    d0.set_cov(pr["cov"])
    _lib.check(lib.mjmpc_softmax_combine(_vp(recs), G, H, A, 0, lam, step, 2, float(P), _vp(d0.mean), _vp(d0.cov),
                                         _vp(d0.value), _vp(d0.wnorm), d0.stream()))
    m_ref, c_ref = cr.dmd_update(pr["costs"], pr["actions"], pr["mean"], pr["cov"], pr["gs"], lam, step, True, "full")
    np.testing.assert_allclose(d0.get_mean(), m_ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(d0.get_cov(), c_ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(d0.value.item(), cr.dmd_value(pr["costs"], pr["gs"], lam), rtol=1e-12)


@pytest.mark.parametrize("G", [2, 4])
def test_cem_and_rs_records_combine(problem, G):
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.control._device import DeviceUpdater
    pr = problem
    P, H, A = pr["P"], pr["H"], pr["A"]
    step, k = 0.7, int(P * 0.2)
    n = P // G
    devs = [DeviceUpdater(H, A, pr["gs"]) for _ in range(G)]
    lib = devs[0].lib
    cs = [torch.from_numpy(pr["costs"][g * n:(g + 1) * n].copy()).cuda() for g in range(G)]
    acts = [torch.from_numpy(pr["actions"][g * n:(g + 1) * n].copy()).cuda() for g in range(G)]
    q_all = torch.empty(P, dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        d.set_mean(pr["mean"])
        d.set_cov(pr["cov"])
        _lib.check(lib.mjmpc_traj_cost(_lib.F64, n, H, A, _vp(cs[g]), _vp(d.gseq), 0, _vp(d.workspace(n)), d.stream()))
        q_all[g * n:(g + 1) * n] = d._q0_view(d.workspace(n), n)
    np.testing.assert_allclose(q_all.cpu().numpy(), cr.cost_to_go(pr["costs"].copy(), pr["gs"])[:, 0], rtol=1e-14)
    # random shooting
    rrecs = torch.empty((G, 2 + H * A), dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        _lib.check(lib.mjmpc_rs_best(_lib.F64, n, H, A, _vp(acts[g]), g * n, _vp(rrecs[g]), _vp(d.workspace(n)), d.stream()))
    scratch = devs[0].mean.clone()
    _lib.check(lib.mjmpc_rs_combine(_vp(rrecs), G, H, A, step, _vp(scratch), devs[0].stream()))
    np.testing.assert_allclose(scratch.cpu().numpy(), cr.rs_update(pr["costs"], pr["actions"], pr["mean"], pr["gs"], step),
                               rtol=1e-13, atol=1e-13)
    # CEM: global-rank elites, two gathered records
    srecs = torch.empty((G, 1 + H * A), dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        _lib.check(lib.mjmpc_cem_elite_sums(_lib.F64, n, H, A, _vp(acts[g]), _vp(q_all), P, g * n, k, _vp(srecs[g]),
                                            _vp(d.workspace(n)), d.stream()))
    assert srecs[:, 0].sum().item() == k
    crecs = torch.empty((G, A * A), dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        _lib.check(lib.mjmpc_cem_elite_cov(_lib.F64, n, H, A, _vp(acts[g]), _vp(d.mean), _vp(srecs), G, _vp(crecs[g]),
                                           _vp(d.workspace(n)), d.stream()))
    for full in (1, 0):
        d = devs[0]
        d.set_mean(pr["mean"])
        d.set_cov(pr["cov"])
        _lib.check(lib.mjmpc_cem_final(_vp(crecs), G, n, H, A, float(k), full, step, _vp(d.mean), _vp(d.cov),
                                       _vp(d.workspace(n)), d.stream()))
        m_ref, c_ref = cr.cem_update(pr["costs"], pr["actions"], pr["mean"], pr["cov"], pr["gs"], 0.2, step,
                                     "full" if full else "diagonal")
        np.testing.assert_allclose(d.get_mean(), m_ref, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(d.get_cov(), c_ref, rtol=1e-11, atol=1e-12)
    # CEM as the controllers run it: ONE record per GPU after the q0 gather (scatter about the rank's own mean
    # delta), pooled by mjmpc_cem_combine - two collectives per iteration
    recs = torch.empty((G, 1 + H * A + A * A), dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        d.set_mean(pr["mean"])
        recs[g, :1 + H * A] = srecs[g]
        _lib.check(lib.mjmpc_cem_elite_cov(_lib.F64, n, H, A, _vp(acts[g]), _vp(d.mean), _vp(srecs[g]), 1,
                                           _vp(crecs[g]), _vp(d.workspace(n)), d.stream()))
        recs[g, 1 + H * A:] = crecs[g]
    for full in (1, 0):
        d = devs[0]
        d.set_mean(pr["mean"])
        d.set_cov(pr["cov"])
        _lib.check(lib.mjmpc_cem_combine(_vp(recs), G, H, A, float(k), full, step, _vp(d.mean), _vp(d.cov), d.stream()))
        m_ref, c_ref = cr.cem_update(pr["costs"], pr["actions"], pr["mean"], pr["cov"], pr["gs"], 0.2, step,
                                     "full" if full else "diagonal")
        np.testing.assert_allclose(d.get_mean(), m_ref, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(d.get_cov(), c_ref, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("G", [2, 3, 4])
def test_fused_mppi_records_and_single_kernel_combine(problem, G):
    """The fused MPPI path sharded: per-shard records [xmax | S | W] from mjmpc_mppi_fused_update (record mode),
    stacked as the all-gather delivers them, merged by mjmpc_mppi_fused_combine (mean update, action, mapped host
    copy + completion flag, step counter, shift in one launch) = the unsharded MPPI update + shift."""
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.control._device import DeviceUpdater
    pr = problem
    P, H, A = pr["P"], pr["H"], pr["A"]
    lam, step = 0.08, 0.6
    n = P // G
    q0 = cr.cost_to_go(pr["costs"].copy(), pr["gs"])[:, 0]
    devs = [DeviceUpdater(H, A, pr["gs"]) for _ in range(G)]
    lib = devs[0].lib
    recs = torch.empty((G, 2 + H * A), dtype=torch.float64, device="cuda")
    for g, d in enumerate(devs):
        d.set_mean(pr["mean"])
        q = torch.from_numpy(q0[g * n:(g + 1) * n].copy()).cuda()
        a = torch.from_numpy(pr["actions"][g * n:(g + 1) * n].copy()).cuda()
        _lib.check(lib.mjmpc_mppi_fused_update(_lib.F64, n, H, A, _vp(q), _vp(a), lam, 0.0, -1, _vp(d.mean), None,
                                               _vp(recs[g]), None, None, None, _vp(d.workspace(n)), d.stream()))
        np.testing.assert_array_equal(d.get_mean(), pr["mean"])              # record mode leaves the mean alone
    d0 = devs[0]
    act = torch.zeros(A, dtype=torch.float64, device="cuda")
    pinned = torch.zeros(A + 1, dtype=torch.float64).pin_memory()
    counter = torch.full((1,), 41, dtype=torch.int64, device="cuda")
    _lib.check(lib.mjmpc_mppi_fused_combine(_vp(recs), G, float(P), H, A, lam, step, 1, _vp(d0.mean), _vp(act), None,
                                            _vp(pinned), _vp(counter), d0.stream()))
    torch.cuda.synchronize()
    m1 = cr.mppi_update(pr["costs"], pr["actions"], pr["mean"], pr["cov"], pr["gs"], lam, 1, step)
    np.testing.assert_allclose(act.cpu().numpy(), m1[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pinned.numpy()[:A], m1[0], rtol=1e-12, atol=1e-12)
    assert pinned.numpy()[A] == 42.0 and int(counter.item()) == 42          # completion flag = the new step count
    np.testing.assert_allclose(d0.get_mean(), cr.shift_mean(m1, "repeat"), rtol=1e-12, atol=1e-12)

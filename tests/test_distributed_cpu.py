"""world_size-2 gloo tests (CPU): the N > 1 path = contiguous particle blocks + ONE all-gather of a
small per-GPU record per reduction + an identical combine on every rank.  The HIP kernels cannot
run here, so the numpy record oracle stands in for them; what is under test is the product's
sharding map (mjmpc_amd.control.sharding), its communicator (TorchDistComm over torch.distributed)
and the record algebra: sharded == unsharded."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import controllers_ref as cr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mjmpc_amd.control._device import TorchDistComm
        from mjmpc_amd.control.sharding import local_block, slice_local
        comm = TorchDistComm()
        assert (comm.rank, comm.world_size) == (rank, world)
        # a code-path decision every rank takes together (graph replay -> eager fallback, controller.py)
        assert comm.all_agree(True) is True
        assert comm.all_agree(rank == 0) is False
        P, H, A, lam, step = 64, 6, 3, 0.05, 0.7
        rs = np.random.RandomState(42)                       # same data on every rank
        mean = rs.randn(H, A) * 0.2
        cov = np.eye(A) * 0.8
        # noise sharding: every rank generates the reference's full stream and keeps its block
        full = cr.generate_noise(cov, [0.25, 0.8, 0.0], (P, H), 123)
        off, n = local_block(P, rank, world)
        mine = slice_local(full, rank, world)
        assert mine.shape[0] == n and np.array_equal(mine, full[off:off + n])
        actions = mean[None] + full
        costs = rs.rand(P, H) * 3 + 0.1 * np.abs(full).sum(-1)
        gs = cr.gamma_seq(0.98, H)
        # ---- softmax family: one all-gather of the record
        rec = cr.softmax_record(costs[off:off + n], actions[off:off + n], mean, gs, lam, want_cov=True)
        recs = comm.all_gather(torch.from_numpy(rec)).numpy()
        assert recs.shape == (world, rec.size)
        m1, c1, val = cr.softmax_combine(recs, mean, cov, H, A, lam, step, 2, P)
        m_ref, c_ref = cr.dmd_update(costs, actions, mean, cov, gs, lam, step, True, "full")
        np.testing.assert_allclose(m1, m_ref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(c1, c_ref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(val, cr.dmd_value(costs, gs, lam), rtol=1e-12)
        np.testing.assert_allclose(m1, cr.mppi_update(costs, actions, mean, cov, gs, lam, 1, step), rtol=1e-12, atol=1e-13)
        # ---- CEM: all-gather q0, global-rank elite flags, then ONE all-gather of {n_g | sum of elite actions |
        # scatter of the rank's elite deltas about ITS OWN mean delta}, pooled by the pairwise-variance identity
        # (mjmpc_cem_combine): two collectives per iteration
        k = int(P * 0.25)
        q0 = cr.cost_to_go(costs.copy(), gs)[:, 0]
        q_all = comm.all_gather_flat(torch.from_numpy(q0[off:off + n].copy())).numpy()
        assert np.array_equal(q_all, q0)                                       # worker-order concatenation
        flags = cr.elite_flags(q0[off:off + n], q_all, off, k)
        n_g = float(flags.sum())
        asum = actions[off:off + n][flags].sum(0)
        mu_g = (asum / n_g - mean).mean(0) if n_g else np.zeros(A)
        d = (actions[off:off + n][flags] - mean[None] - mu_g).reshape(-1, A)
        rec = np.concatenate([[n_g], asum.reshape(-1), (d.T @ d).reshape(-1)])
        recs = comm.all_gather(torch.from_numpy(rec)).numpy()
        assert recs[:, 0].sum() == k
        elite_mean = recs[:, 1:1 + H * A].sum(0).reshape(H, A) / k
        mu = (elite_mean - mean).mean(0)
        S = np.zeros((A, A))
        for r in recs:
            S += r[1 + H * A:].reshape(A, A)
            if r[0]:
                dm = (r[1:1 + H * A].reshape(H, A) / r[0] - mean).mean(0) - mu
                S += H * r[0] * np.outer(dm, dm)
        cov_cem = (1 - step) * cov + step * S / (H * k - 1)
        mean_cem = (1 - step) * mean + step * elite_mean
        m_ref, c_ref = cr.cem_update(costs, actions, mean, cov, gs, 0.25, step, "full")
        np.testing.assert_allclose(mean_cem, m_ref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(cov_cem, c_ref, rtol=1e-11, atol=1e-13)
        # ---- random shooting: all-gather of (min, global index, actions)
        i = int(np.argmin(q0[off:off + n]))
        rrec = np.concatenate([[q0[off + i], off + i], actions[off + i].reshape(-1)])
        rrecs = comm.all_gather(torch.from_numpy(rrec)).numpy()
        best = min(range(world), key=lambda g: (rrecs[g, 0], rrecs[g, 1]))
        m_rs = (1 - step) * mean + step * rrecs[best, 2:].reshape(H, A)
        np.testing.assert_allclose(m_rs, cr.rs_update(costs, actions, mean, gs, step), rtol=1e-13)
        # ---- PFMPC: the (P,H) costs of the local rollouts are all-gathered; weights and the systematic
        # resampling are then replicated work with identical seeds -> identical survivors on every rank
        from mjmpc_amd.control.particle_filter_controller import systematic_resample_indices
        c_all = comm.all_gather_flat(torch.from_numpy(costs[off:off + n].reshape(-1).copy())).numpy().reshape(P, H)
        assert np.array_equal(c_all, costs)
        w_pf = cr.pf_weights(c_all, gs, 0.4)
        idx_pf = systematic_resample_indices(w_pf, 0.3 / P)
        s_ref, _ = cr.pf_resample(full, cr.pf_weights(costs, gs, 0.4), 7)
        import random as _random
        _random.seed(7)
        assert np.array_equal(full[systematic_resample_indices(w_pf, _random.uniform(0.0, 1.0 / P * 1.0))], s_ref)
        # every rank ends with bit-identical results
        digest = torch.from_numpy(np.concatenate([m1.reshape(-1), c1.reshape(-1), mean_cem.reshape(-1), m_rs.reshape(-1),
                                                  idx_pf.astype(np.float64)]))
        both = comm.all_gather(digest)
        assert torch.equal(both[0], both[1])
        q.put((rank, "ok"))
    except Exception as e:                # surface the failure to the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_updates_match_unsharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d:\n%s" % (rank, msg)


def test_local_block_mapping():
    from mjmpc_amd.control.sharding import local_block
    assert [local_block(16, r, 4) for r in range(4)] == [(0, 4), (4, 4), (8, 4), (12, 4)]
    with pytest.raises(AssertionError):
        local_block(10, 0, 4)

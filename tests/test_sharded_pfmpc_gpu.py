"""GPU, two processes sharing cuda:0 over gloo: PFMPC with sharded rollouts (each rank rolls out its contiguous
block, the (P,H) costs are all-gathered, weights and resampling are replicated) walks exactly the same closed
loop as the single-process controller."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(comm, steps=4):
    from mjmpc_amd.control import PFMPC
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    c = PFMPC(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=10, cov_shift=0.05, cov_resample=0.6,
              base_action="null", lam=0.3, num_particles=128, gamma=0.99, n_iters=1, action_lows=eng.action_lows,
              action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=11, comm=comm)
    c.set_sim_state_fn = eng.set_env_state
    c.rollout_fn = make_rollout_fn(eng)
    state = dict(qp=np.array([0.1, 0.2, 0.0, -0.5, 0.0, -0.3, 0.0]), qv=np.zeros(7), qa=np.zeros(7),
                 target_pos=np.array([0.2, -0.1, 0.2]), timestep=0)
    acts = []
    for _ in range(steps):
        a, _ = c.optimize(state)
        acts.append(a)
        eng.set_env_state(state)
        _, nobs = eng.step_state(a)
        nobs = nobs.cpu().numpy()
        state = dict(qp=nobs[:7].copy(), qv=nobs[7:14].copy(), qa=np.zeros(7), target_pos=state["target_pos"], timestep=0)
    return np.array(acts), c.action_samples.copy()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mjmpc_amd.control._device import TorchDistComm
        acts, samples = _run(TorchDistComm())
        q.put((rank, acts, samples))
    finally:
        dist.destroy_process_group()


def test_sharded_pfmpc_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, acts, samples = q.get(timeout=240)
        got[rank] = (acts, samples)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_acts, ref_samples = _run(None)
    np.testing.assert_array_equal(got[0][0], got[1][0])          # the replicas agree with each other ...
    for rank in (0, 1):                                          # ... and with the unsharded controller
        np.testing.assert_array_equal(got[rank][0], ref_acts)
        np.testing.assert_array_equal(got[rank][1], ref_samples)

"""The sharded control iteration on ONE GPU: a world-size-1 RCCL group whose communicator CLAIMS two ranks (this process
is rank 0 and owns the first half of the particles; the all-gather really gathers one record).  The captured graph then
holds what a rank of a multi-GPU run replays - rollout + record launches, the RCCL all-gather, the combine kernel, the
env step - and its closed loop must equal a plain single-GPU run over that half population (same global particle
indices, hence the same samples).  Run by tools/test_dist_onegpu.sh and tests/test_rccl_world1_gpu.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mjmpc_amd.control import MPPI
from mjmpc_amd.control._device import TorchDistComm
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
from mjmpc_amd.models.reacher7dof import reacher7dof_raw


COMMS = []


class ClaimsTwoRanks(TorchDistComm):
    def __init__(self):
        super().__init__()
        COMMS.append(self)
        self.world_size = 2         # sharding map and code paths of a two-rank run; the gather returns ONE record

    def all_gather(self, t):
        self.world_size = 1
        try:
            return super().all_gather(t)
        finally:
            self.world_size = 2


def run(P, comm, mono):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    c = MPPI(d_state=25, d_obs=20, d_action=7, horizon=16, init_cov=1.0, base_action="null", lam=0.05, num_particles=P,
             step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows, action_highs=eng.action_highs,
             filter_coeffs=[0.25, 0.8, 0.0], seed=3, noise_mode="device", comm=comm)
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c.enable_graph(post_step=eng.step_state, mono=mono)
    acts = [c.optimize({})[0] for _ in range(6)]
    torch.cuda.synchronize()
    # ... and once more after a reset (bench.py resets controller and env behind its process warm-up: the iteration is
    # captured a second time, collective included)
    c.reset()
    eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=eng.model.target_default.copy()))
    again = [c.optimize({})[0] for _ in range(6)]
    torch.cuda.synchronize()
    assert np.abs(np.array(again) - np.array(acts)).max() < 1e-12, "the closed loop after reset() differs"
    info = (c.local_particles, c._mono, getattr(c, "graph_fallback", False), c._graph is not None, c.launch_mode)
    c._graph = None             # (captured graphs hold the communicator's resources: gone before the group is)
    return np.array(acts), info


ref, _ = run(512, None, True)                       # 512 particles on one GPU
for mono in (True, False):
    comm2 = ClaimsTwoRanks()
    # which exchange path this communicator gives the controllers (bench.py's config.collectives), and what it cost to make
    want = "torch.distributed" if os.environ.get("MJMPC_TORCH_COLLECTIVES") else "library RCCL"
    assert comm2.collectives == want and not comm2.fell_back, (comm2.collectives, comm2.why_fell_back)
    print("collectives: %s (communicator made in %.3f s)" % (comm2.collectives, comm2.init_seconds))
    got, info = run(1024, comm2, mono)              # "1024 over two ranks": this rank's 512 are the same particles
    assert info[0] == 512 and info[1] == mono and not info[2]
    assert info[3], "the sharded iteration should run as a captured graph, from its launch tape or as direct launches"
    # the all-gather is issued by the library (TorchDistComm.lib_collectives): the fused iteration is three direct launches,
    # the separate launches run from the launch tape; with MJMPC_TORCH_COLLECTIVES=1 (torch.distributed's collective) both
    # replay a hipGraph
    if os.environ.get("MJMPC_TORCH_COLLECTIVES"):
        assert info[4] == "hipGraph replay", info[4]
    else:
        assert info[4].startswith("launched directly" if mono else "launch tape"), info[4]
    err = np.abs(got - ref).max()
    if os.environ.get("RCCL_W1_DEBUG"):
        print(np.abs(got - ref).max(axis=1)); print(ref[:2]); print(got[:2])
    print("sharded iteration (%s; %s) with an RCCL all-gather: max |d action| vs the single-GPU run = %.2e"
          % ("rollout + record launches" if mono else "separate launches", info[4], err))
    assert err < 1e-9 or os.environ.get('RCCL_W1_DEBUG')

# The other controllers' exchanges inside a captured graph: CEM (the q0 all-gather + the elite-record all-gather), DMD-MPC
# with an adapting covariance and random shooting (one record all-gather each) over a REAL world-size-1 RCCL communicator
# (its collectives are forced even where one rank could skip them) = the same controller without a communicator.
from mjmpc_amd.control import CEM, DMDMPC, RandomShooting


class AlwaysCollective(TorchDistComm):
    always_collective = True

    def __init__(self):
        super().__init__()
        COMMS.append(self)


def run_other(make, comm):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    kw = dict(d_state=25, d_obs=20, d_action=7, horizon=16, num_particles=512, n_iters=1, gamma=1.0, step_size=0.8,
              action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=5,
              base_action="null", noise_mode="device", comm=comm)
    c = make(kw)
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    assert c._graph_capable()
    c.enable_graph(post_step=eng.step_state)
    acts = np.array([c.optimize({})[0] for _ in range(5)])
    torch.cuda.synchronize()
    assert c._graph is not None and not getattr(c, "graph_fallback", False)
    if comm is not None and not os.environ.get("MJMPC_TORCH_COLLECTIVES"):
        assert c.launch_mode.startswith("launch tape"), c.launch_mode      # (its exchanges are library calls)
    c._graph = None             # (captured graphs hold the communicator's resources: gone before the group is)
    return acts


for name, make in (("CEM full covariance", lambda kw: CEM(init_cov=1.0, elite_frac=0.1, beta=0.1, cov_type="full", **kw)),
                   ("DMD-MPC update_cov", lambda kw: DMDMPC(init_cov=1.0, lam=0.1, beta=0.1, update_cov=True, cov_type="diagonal", **kw)),
                   ("random shooting", lambda kw: RandomShooting(init_cov=1.0, **kw))):
    a0, a1 = run_other(make, None), run_other(make, AlwaysCollective())
    err = np.abs(a0 - a1).max()
    print("%s: captured iteration with its RCCL exchanges vs without a communicator: max |d action| = %.2e" % (name, err))
    assert err < 1e-10
print("ok", flush=True)


def shutdown():
    """Captured graphs that hold RCCL kernels must be gone before the process group is: the watchdog thread of the group
    otherwise races interpreter shutdown (seen as an abort at exit, one run in a few)."""
    import gc
    gc.collect()
    torch.cuda.synchronize()
    for c in COMMS:
        c.close()               # (the library's own communicators)
    dist.destroy_process_group()


if "--time" not in sys.argv:
    shutdown()

if "--time" in sys.argv:        # per-step time of the sharded iteration (4096 particles on this rank) beside the single-GPU one
    import time

    def timed(P, comm, mono, H=32):
        eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
        c = MPPI(d_state=25, d_obs=20, d_action=7, horizon=H, init_cov=1.0, base_action="null", lam=0.01, num_particles=P,
                 step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows, action_highs=eng.action_highs,
                 filter_coeffs=[0.25, 0.8, 0.0], seed=3, noise_mode="device", comm=comm)
        c.rollout_fn = make_device_rollout_fn(eng)
        c.set_sim_state_fn = lambda s: None
        c.enable_graph(post_step=eng.step_state, mono=mono)
        for _ in range(60):
            c.optimize({})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            c.optimize({})
        torch.cuda.synchronize()
        c._graph = None
        return (time.perf_counter() - t0) / 200 * 1e3

    print("single GPU, 4096 particles:                      %.4f ms per step" % timed(4096, None, True))
    print("rank of a sharded run (RCCL all-gather, 3 launches): %.4f ms per step" % timed(8192, ClaimsTwoRanks(), True))
    print("rank of a sharded run, separate launches:          %.4f ms per step" % timed(8192, ClaimsTwoRanks(), False))
    shutdown()

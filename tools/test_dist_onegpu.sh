#!/bin/bash
# N > 1 code paths on a ONE-GPU box: (a) RCCL with world_size 1 -> is the collective capturable in a hipGraph?
# (b) two ranks sharing cuda:0 over gloo -> rank / sharding / barrier / MAX-reduce plumbing of bench.py
cd "$(dirname "$0")/.."
echo "== (a) nccl world 1, graph"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 180 python - <<'PY'
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mjmpc_amd.control import MPPI
from mjmpc_amd.control._device import TorchDistComm
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
class ForceSharded(TorchDistComm):          # pretend to be sharded so that the collective path is exercised
    pass
def run(use_comm, graph):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    comm = ForceSharded() if use_comm else None
    if comm is not None:
        comm.world_size_real = comm.world_size
    c = MPPI(d_state=25, d_obs=20, d_action=7, horizon=16, init_cov=1.0, base_action="null", lam=0.05, num_particles=512,
             step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows, action_highs=eng.action_highs,
             filter_coeffs=[0.25, 0.8, 0.0], seed=3, noise_mode="device", comm=comm)
    if comm is not None:
        c.dev.comm.world_size = 1
    c.rollout_fn = make_device_rollout_fn(eng); c.set_sim_state_fn = lambda s: None
    if graph:
        c.enable_graph(post_step=eng.step_state)
    acts = []
    for _ in range(5):
        a, _ = c.optimize({})
        if not graph: eng.step_state(a)
        acts.append(a)
    return np.array(acts), getattr(c, "graph_fallback", False)
ref, _ = run(False, False)
import mjmpc_amd.control._device as D
# route the single-GPU fused update through the collective (sharded) branch: the communicator claims two ranks for
# the duration of the call, the all-gather over the real world-size-1 RCCL group returns one record
orig = D.DeviceUpdater.mppi_fused_update
def patched(self, *a, **k):
    if not hasattr(self.comm, "backend"):
        return orig(self, *a, **k)
    ws, gather = self.comm.world_size, self.comm.all_gather
    def true_size_gather(t):
        self.comm.world_size = ws
        try:
            return gather(t)
        finally:
            self.comm.world_size = 2
    self.comm.world_size, self.comm.all_gather = 2, true_size_gather
    try:
        return orig(self, *a, **k)
    finally:
        self.comm.world_size, self.comm.all_gather = ws, gather
D.DeviceUpdater.mppi_fused_update = patched
got, fb = run(True, True)
print("rccl-in-graph fallback:", fb, " max |d action| vs eager single:", np.abs(got - ref).max())
assert np.abs(got - ref).max() < 1e-9
dist.destroy_process_group()
PY
echo "== (b) gloo world 2 on cuda:0"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo --device 0 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-700

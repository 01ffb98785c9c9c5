"""Host-side cost of the captured control step: python tools/host_overhead.py (needs the GPU)."""
import os, sys, time, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.control import MPPI
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=32, init_cov=1.0, base_action="null", lam=0.01,
         num_particles=4096, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
         action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=123, noise_mode="device", noise_dtype="f64")
c.rollout_fn = make_device_rollout_fn(eng); c.set_sim_state_fn = lambda s: None
eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
c.enable_graph(post_step=eng.step_state)
state = {"resident": True}
for _ in range(5): c.optimize(state)
torch.cuda.synchronize()
def t(fn, n=20000):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("sync_in        %.2f us" % t(c._sync_in))
print("deepcopy state %.2f us" % t(lambda: copy.deepcopy(state)))
print("noise_ahead    %.2f us" % t(c._noise_ahead))
print("fused_capable  %.2f us" % t(c._fused_capable))
print("flag copy      %.2f us" % t(lambda: c._action_np[:7].copy()))
launch = c._device_iteration if c._graph == "direct" else c._graph.replay
N = 300
t0 = time.perf_counter()
for _ in range(N): launch()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("iteration enqueue %.2f us (%s)" % ((t1 - t0) / N * 1e6, "two direct launches" if c._graph == "direct" else "graph replay"))
c._step_dev.fill_(c.num_steps); torch.cuda.synchronize()
c._noise_valid = False
# host time of one optimize() with the GPU work taken out: the launch and the wait replaced by no-ops
real_iter, real_wait = c._device_iteration, c._wait_action
c._device_iteration = lambda: None
c._wait_action = lambda: c._action_np[:7].copy()
print("optimize() host path without launch and wait %.2f us" % t(lambda: c.optimize(state), 5000))
c._device_iteration, c._wait_action = real_iter, real_wait
c.reset(); eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
for _ in range(100): c.optimize(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): c.optimize(state)
torch.cuda.synchronize()
print("optimize loop  %.2f us per step" % ((time.perf_counter() - t0) / 2000 * 1e6))

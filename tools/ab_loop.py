"""In-process A/B of the control loop: episodes of `warmup + steps` control steps from qpos0 (what the driver's bench run
measures), the variants interleaved and repeated, medians reported.   python tools/ab_loop.py [P] [steps] [episodes]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.control import MPPI
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
from mjmpc_amd.models.reacher7dof import reacher7dof_raw

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
episodes = int(sys.argv[3]) if len(sys.argv) > 3 else 9
warm, H = 5, 32
variants = {"separate launches (graph)": dict(mono=False), "fused": dict(mono=True), "fused + lookahead": dict(mono=True, lookahead=True)}
setups = {}
for name, kw in variants.items():
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, init_cov=1.0, base_action="null", lam=0.01,
             num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
             action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=123, noise_mode="device", noise_dtype="f64")
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c.enable_graph(post_step=eng.step_state, **kw)
    setups[name] = (eng, c, kw)
times = {k: [] for k in variants}
st = {"resident": True}
for ep in range(episodes):
    for name, (eng, c, kw) in setups.items():
        torch.cuda.synchronize()
        c.reset()
        c.enable_graph(post_step=eng.step_state, **kw)
        eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
        for _ in range(warm):
            c.optimize(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            c.optimize(st)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / steps * 1e3)
for name, v in times.items():
    v = np.array(v[1:])
    print("%-28s median %.4f ms  min %.4f  max %.4f   (%d episodes of %d+%d steps, P=%d)" % (name, np.median(v), v.min(), v.max(), len(v), warm, steps, P))

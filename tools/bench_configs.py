"""The other BASELINE.json configurations on ONE GPU (bench.py stays the headline: MPPI 4096 x 32).

    python tools/bench_configs.py [--steps 30] [--dtype f64]

  cfg1  reacher_7dof-v0 MPPI, 1024 particles x H32, lam 0.01
  cfg3  reacher_7dof-v0 CEM full covariance, 16384 particles x H32, elite_frac 0.1   (BASELINE shards it over 4 GPUs;
        here the whole population on one)
  cfg4* DMD-MPC, 65536 particles x H64 on the 7-dof arm (BASELINE names pen-v0, whose assets are not in the
        reference tree: throughput of the same controller path on the model we have, flagged)
  cfg4t DMD-MPC, 65536 particles x H64 on the synthetic 24-dof hand-on-an-arm TREE (mjmpc_amd/models/hand24.py,
        tree kernel): pen-v0's SHAPE of work - a branching 24-hinge tree, gravity, fingertip contacts; flagged
  cfg4p DMD-MPC, 65536 particles x H64 on the synthetic pen-in-hand model (mjmpc_amd/models/pen_hand.py): the hand with
        position servos, a 6-dof pen, capsule-capsule contacts with friction, pen-v0's orientation reward; flagged
One step = Controller.optimize() + stepping the real arm on the device, all data resident in HBM.  One JSON line each.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _run_arm(name, make, P, H, steps, warmup, dtype, note):
    import torch
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dtype)
    ctrl = make(eng, P, H)
    ctrl.rollout_fn = make_device_rollout_fn(eng)
    ctrl.set_sim_state_fn = lambda s: None
    eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
    graphed = ctrl._graph_capable()
    if graphed:
        ctrl.enable_graph(post_step=eng.step_state)
    state = {"resident": True}

    def step():
        a, _ = ctrl.optimize(state)
        if not graphed:
            eng.step_state(a)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _, nobs = eng.step_state(np.zeros(7))
    print(json.dumps({"config": name, "particles": P, "horizon": H, "dtype": dtype, "steps": steps,
                      "ms_per_step": dt / steps * 1e3, "control_loop_hz": steps / dt,
                      "particle_steps_per_s": P * H * ctrl.n_iters * steps / dt,
                      "launch": _launch_kind(ctrl, graphed),
                      "final_distance_to_target": float(torch.linalg.norm(nobs[17:20]).item()),
                      "solver_failures": eng.solver_failures(), "note": note}), flush=True)


def _launch_kind(ctrl, graphed):
    if not graphed:
        return "eager launches"
    return "two kernels per iteration, launched directly" if getattr(ctrl, "_graph", None) == "direct" else "hipGraph replay"


def run_tree(name, make, P, H, steps, warmup, dtype, note, raw_fn=None, env_cls=None, noise_scale=0.5):
    """The same loop on the tree engine (the real env kept on the device and captured with the iteration).  With `env_cls`
    (a locomotion model) that class draws the reference's reset noise for the start state."""
    import torch
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.hand24 import hand24_raw
    raw = (raw_fn or hand24_raw)()
    eng = TreeRolloutEngine(raw, dtype=dtype)
    ctrl = make(eng, P, H)
    ctrl.rollout_fn = make_device_rollout_fn(eng)
    ctrl.set_sim_state_fn = lambda s: None               # the real env lives on the device (TreeRolloutEngine.step_state)
    env = env_cls(dtype=dtype) if env_cls else None
    if env is not None:
        env.reset(seed=123)
        eng.set_env_state(env.get_env_state())
        x0 = float(env.get_env_state()["qpos"][0])
    else:
        eng.reset()
    graphed = ctrl._graph_capable()
    if graphed:
        ctrl.enable_graph(post_step=eng.step_state)
    resident = {"resident": True}

    def step():
        a, _ = ctrl.optimize(resident)
        if not graphed:
            eng.step_state(a)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    state = eng.get_state_device()
    _, nobs_t = eng.step_state(np.zeros(eng.d_action))
    nobs = nobs_t.cpu().numpy()
    # the rollout kernel alone, back to back on the run's own buffers
    noise_t = ctrl.dev._rec[("noise", dtype)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.rollout_device(P, H, ctrl.dev.mean, noise_t)
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / 3
    nv = eng.model.nv
    out = {"config": name, "particles": P, "horizon": H, "dtype": dtype, "steps": steps,
           "ms_per_step": dt / steps * 1e3, "control_loop_hz": steps / dt,
           "particle_steps_per_s": P * H * ctrl.n_iters * steps / dt, "rollout_kernel_ms": kern_ms,
           "launch": _launch_kind(ctrl, graphed), "dofs": nv, "frame_skip": raw.frame_skip,
           "solver_failures": eng.solver_failures(), "note": note}
    if env is not None:
        out["forward_progress_m"] = float(state["qpos"][0]) - x0       # since the reset (whose noise moves qpos[0] too)
        # the CPU restatement on this host's cores, same model and start state, through bench.py's cpu_baseline leg
        # (a bounded sample; for scale, not a target)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        cb = bench.cpu_baseline(P, H, 4.0, raw=raw, qpos=state["qpos"], qvel=state["qvel"], noise_scale=noise_scale)
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "single_thread_value")}
    else:
        out["final_distance_to_target"] = float(np.linalg.norm(nobs[2 * nv + 3:2 * nv + 6]))
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--tree-particles", type=int, default=65536)
    ap.add_argument("--only-tree", action="store_true")
    ap.add_argument("--only-hand", action="store_true", help="of the tree configurations, only the 24-dof hand")
    ap.add_argument("--pen", action="store_true", help="also cfg4p: the pen-in-hand model at the same size (about 1 s per step)")
    ap.add_argument("--only", default="", help="run only the arm configurations whose name starts with this (cfg1, cfg3, cfg4)")
    args = ap.parse_args()
    from mjmpc_amd.control import CEM, DMDMPC, MPPI

    def kw(eng, P, H):
        return dict(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, num_particles=P, n_iters=1,
                    action_lows=eng.action_lows, action_highs=eng.action_highs, seed=123, noise_mode="device",
                    noise_dtype=args.dtype)

    def run(name, *a):
        if name.startswith(args.only):
            _run_arm(name, *a)

    if not args.only_tree:
        run("cfg1 reacher_7dof-v0 MPPI 1024xH32",
            lambda e, P, H: MPPI(init_cov=1.0, base_action="null", lam=0.01, step_size=1.0, alpha=1, gamma=1.0,
                                 filter_coeffs=[0.25, 0.8, 0.0], **kw(e, P, H)),
            1024, 32, args.steps, args.warmup, args.dtype, "")
        run("cfg3 reacher_7dof-v0 CEM full-cov 16384xH32 elite 0.1 (one GPU)",
            lambda e, P, H: CEM(init_cov=1.0, base_action="null", elite_frac=0.1, step_size=1.0, gamma=1.0, beta=0.1,
                                cov_type="full", filter_coeffs=[0.25, 0.8, 0.0], **kw(e, P, H)),
            16384, 32, args.steps, args.warmup, args.dtype, "covariance adapts every step on the device (moments, refit, "
            "Cholesky): the iteration replays as one hipGraph")
        run("cfg4* DMD-MPC 65536xH64 on the 7-dof arm (pen-v0 assets absent)",
            lambda e, P, H: DMDMPC(init_cov=1.0, beta=0.1, base_action="null", lam=0.1, step_size=1.0, gamma=1.0,
                                   update_cov=False, cov_type="diagonal", filter_coeffs=[0.25, 0.8, 0.0], **kw(e, P, H)),
            65536, 64, max(5, args.steps // 3), 2, args.dtype, "stand-in model; throughput only")

    def kw24(eng, P, H):
        return dict(kw(eng, P, H), d_action=24)

    if args.only:
        return
    if not args.only_hand:
        from mjmpc_amd.envs.locomotion_env import HalfCheetahEnv, SwimmerEnv
        from mjmpc_amd.models.half_cheetah import half_cheetah_raw
        from mjmpc_amd.models.swimmer import swimmer_raw
        for nm, raw_fn, env_cls, A in (("HalfCheetah-v0", half_cheetah_raw, HalfCheetahEnv, 6), ("Swimmer-v0", swimmer_raw, SwimmerEnv, 4)):
            run_tree("loco %s MPPI 4096xH32 (reference-registered env over its vendored XML; no reference experiment file)" % nm,
                     lambda e, P, H, A=A: MPPI(init_cov=0.3, base_action="null", lam=0.2, step_size=1.0, alpha=1, gamma=1.0,
                                               filter_coeffs=[0.25, 0.8, 0.0], **dict(kw(e, P, H), d_action=A)),
                     4096, 32, max(10, args.steps // 2), 2, args.dtype, "tree engine, full instantiation", raw_fn, env_cls,
                     noise_scale=float(np.sqrt(0.3)))
    run_tree("cfg4t DMD-MPC 65536xH64 on the synthetic 24-dof hand tree (pen-v0 assets absent)",
             lambda e, P, H: DMDMPC(init_cov=0.3, beta=0.1, base_action="null", lam=0.1, step_size=1.0, gamma=1.0,
                                    update_cov=False, cov_type="diagonal", filter_coeffs=[0.25, 0.8, 0.0], **kw24(e, P, H)),
             args.tree_particles, 64, max(3, args.steps // 10), 1, args.dtype,
             "synthetic tree with pen-v0's shape of work (24 hinges, gravity, 5 fingertip contacts); throughput only")
    if args.pen:        # bench.py has this workload (servo set points as the nominal control, 'repeat' shift): its line
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "pen_hand", "--controller", "dmd",
                        "--particles", str(args.tree_particles), "--horizon", "64", "--steps", str(max(3, args.steps // 10)),
                        "--warmup", "1", "--process-warmup", "0", "--no-cpu-baseline", "--dtype", args.dtype], check=True)


if __name__ == "__main__":
    main()

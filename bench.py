#!/usr/bin/env python
"""Headline benchmark: reacher_7dof-v0 MPPI, 4096 particles x H=32 per GPU (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one control iteration, i.e. one ``Controller.optimize()``: sample noise ->
roll out every particle for H env steps (2 MuJoCo substeps each) -> MPPI update -> shift -> D2H of
the action, followed by stepping the "real" arm with that action.  Everything but the action stays
in HBM.  N > 1: weak scaling, each rank owns a contiguous block of 4096 particles; the per-GPU softmax
record is all-gathered (RCCL over xGMI) once per iteration.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = arm_rollout_kernel, timed live with
events on the launch stream; the HBM block SURVEY 8d defines plus `valu`, the roofline that actually binds:
FLOPs per particle-step COUNTED by the instrumented oracle build, oracle/flop_count.cpp) and `cpu_baseline`
(oracle/ on the host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6       # vector FP64; FP32 157.3
FP32_VALU_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=4096, help="particles PER GPU")
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--noise", choices=["device", "mt19937", "host"], default="device",
                    help="device: Philox on the GPU; mt19937: the reference's own numpy stream regenerated on the GPU "
                         "(seed-identical particles); host: numpy on the host, uploaded")
    ap.add_argument("--workload", choices=["reacher", "half_cheetah", "swimmer", "hand24"], default="reacher",
                    help="reacher: the BASELINE.json headline (default).  The others run the same MPPI loop on the tree engine "
                         "(SURVEY 8f rank 4: the reference's vendored HalfCheetah / Swimmer models, the synthetic 24-dof hand) "
                         "and print the same line for them; one GPU")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)          # test knobs: gloo on one GPU
    ap.add_argument("--device", type=int, default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_baseline(P, H, budget_s, raw=None, qpos=None, qvel=None, noise_scale=1.0):
    """The FP64 C oracle (kind "port") on this box's host cores: same model, same start state, same
    noise recipe; a bounded sample of the workload (batches of >= 32 particles per thread x H until the budget).
    `raw` (default: the reacher) lets tools/bench_configs.py time the other models through this same leg."""
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    from oracle.physics_ref import RefArm, threads
    raw = raw or reacher7dof_raw()
    arm = RefArm(raw.to_flat())
    A = len(raw.actuators)
    q0 = np.zeros(arm.nv) if qpos is None else np.asarray(qpos, float)
    v0 = np.zeros(arm.nv) if qvel is None else np.asarray(qvel, float)
    avail = len(os.sched_getaffinity(0))
    try:                                    # a cgroup CPU quota below the visible core count: more threads only thrash
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            avail = max(1, min(avail, int(np.ceil(int(q) / int(per)))))
    except (OSError, ValueError):
        pass
    cores = threads(avail)
    rs = np.random.RandomState(123)
    batch = max(512, 32 * cores)            # >= 32 particles per thread, or OpenMP fork/join dominates
    noise = noise_scale * rs.standard_normal((batch, H, A))
    for t in range(2, H):
        noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
    mean = np.zeros((H, A))
    tgt = np.array([0.1, 0.1, 0.1])
    arm.rollout(q0, v0, tgt, mean, noise[:64], want_obs=False)     # warm up
    n, t0 = 0, time.time()
    while time.time() - t0 < 0.8 * budget_s:
        arm.rollout(q0, v0, tgt, mean, noise, want_obs=False)
        n += batch
    dt = time.time() - t0
    # the same code on ONE thread (2 s), to tell host-core scaling limits (cgroup quotas, SMT) from code speed
    threads(1)
    n1, t1 = 0, time.time()
    while time.time() - t1 < 0.2 * budget_s:
        arm.rollout(q0, v0, tgt, mean, noise[:64], want_obs=False)
        n1 += 64
    dt1 = time.time() - t1
    threads(cores)
    quota = ""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota = "; cgroup cpu.max = %s" % f.read().strip()
    except OSError:
        pass
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": n * H / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": n1 * H / dt1, "cpu_model": cpu_model, "visible_cpus": len(os.sched_getaffinity(0)),
            "sample": "%d particles x H=%d rollouts of the same workload (OpenMP over particles, %d threads), %.1f s; "
                      "one thread: %d particles in %.1f s%s" % (n, H, cores, dt, n1, dt1, quota)}


def tree_workload(args):
    """The same line for a tree-engine model: MPPI closed loop, the real env kept on the device
    (``TreeRolloutEngine.step_state``) and captured with the iteration in a hipGraph, as for the reacher."""
    import torch
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.compile_tree import compile_tree
    if args.workload == "hand24":
        from mjmpc_amd.models.hand24 import hand24_raw
        raw, env, name, lam, cov = hand24_raw(), None, "hand_tree-v0 (synthetic 24-dof hand)", 0.05, 0.3
    else:
        from mjmpc_amd.envs import locomotion_env
        from mjmpc_amd.models.half_cheetah import half_cheetah_raw
        from mjmpc_amd.models.swimmer import swimmer_raw
        raw = dict(half_cheetah=half_cheetah_raw, swimmer=swimmer_raw)[args.workload]()
        env = dict(half_cheetah=locomotion_env.HalfCheetahEnv, swimmer=locomotion_env.SwimmerEnv)[args.workload](dtype=args.dtype)
        name, lam, cov = dict(half_cheetah="HalfCheetah-v0", swimmer="Swimmer-v0")[args.workload], 0.2, 0.3
    torch.cuda.set_device(0)
    P, H = args.particles, args.horizon
    eng = TreeRolloutEngine(raw, dtype=args.dtype)
    A = eng.d_action
    ctrl = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=A, horizon=H, init_cov=cov, base_action="null", lam=lam,
                num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
                action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=123, noise_mode="device",
                noise_dtype=args.dtype)
    ctrl.rollout_fn = make_device_rollout_fn(eng)
    ctrl.set_sim_state_fn = lambda s: None           # the "real" env lives on the device (step_state)
    if env is not None:
        env.reset(seed=123)                         # the reference's reset noise, then the engine owns the state
        eng.set_env_state(env.get_env_state())
    else:
        eng.reset()
    graphed = not args.no_graph and ctrl._graph_capable()
    if graphed:
        ctrl.enable_graph(post_step=eng.step_state)
    resident = {"resident": True}

    def control_step():
        a, _ = ctrl.optimize(resident)
        if not graphed:
            eng.step_state(a)

    for _ in range(args.warmup):
        control_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        control_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    state = eng.get_state_device()
    noise_t = ctrl.dev._rec[("noise", args.dtype)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.rollout_device(P, H, ctrl.dev.mean, noise_t)
    e0.record()
    for _ in range(5):
        eng.rollout_device(P, H, ctrl.dev.mean, noise_t)
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / 5
    s = 8 if args.dtype == "f64" else 4
    b_alg = (3 * A + 2) * s
    achieved = b_alg * P * H / (kern_ms * 1e-3) / 1e9
    qpos = state["qpos"] if "qpos" in state else state["qp"]
    qvel = state["qvel"] if "qvel" in state else state["qv"]
    valu = None
    try:
        from oracle.physics_ref import count_flops
        rs = np.random.RandomState(123)
        fl = count_flops(raw.to_flat(), qpos, qvel, np.zeros(3), np.zeros((H, A)), np.sqrt(cov) * rs.standard_normal((32, H, A)))
        peak_tf = FP64_VALU_PEAK_TF if args.dtype == "f64" else FP32_VALU_PEAK_TF
        tf = fl["flops"] * P * H / (kern_ms * 1e-3) / 1e12
        valu = {"bound": "valu", "flops_per_particle_step": fl["flops"], "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s",
                "frac": tf / peak_tf,
                "counted_by": "oracle/flop_count.cpp on a 32 x H sample from the run's last state: the ORACLE's formulation "
                              "(dense Jacobian-built mass matrix and Cholesky), which on a tree does more arithmetic than the "
                              "kernel's sparse one - an upper bound on the kernel's useful FLOP rate, not its instruction count"}
    except Exception as e:
        valu = {"bound": "valu", "error": "FLOP-counting oracle build unavailable: %s" % (e,)}
    m = compile_tree(raw)
    out = {"metric": "particle-steps/sec (%s MPPI %dp x H%d, control loop incl. noise, rollout, update, shift)" % (name, P, H),
           "value": P * H * args.steps / dt, "unit": "particle-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "%s MPPI lam=%g H=%d, %d particles, frame_skip %d, %d dofs, filter [0.25,0.8,0], closed loop "
                                  "(tree engine; not a BASELINE.json configuration)" % (name, lam, H, P, raw.frame_skip, m.nv),
                      "noise": "device", "particles_per_gpu": P, "horizon": H, "ranks_seen": 1, "backend": None,
                      "launch": "hipGraph replay" if (graphed and not getattr(ctrl, "graph_fallback", False)) else "eager"},
           "control_loop_hz": args.steps / dt,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "traffic_source": None, "valu": valu, "alg_bytes_per_launch": b_alg * P * H,
                        "kernel": "tree_rollout_kernel<%s>" % ("double" if args.dtype == "f64" else "float"), "kernel_ms": kern_ms,
                        "kernel_entry": "mjmpc_tree_rollout", "alg_bytes_per_particle_step": b_alg,
                        "note": "latency-bound path (DESIGN 4.6): serial rounds of the tree factorisation and solves, the Newton "
                                "loop of the wavefront's slowest particle"},
           "solver_failures": eng.solver_failures()}
    if env is not None:
        out["forward_progress_m"] = float(qpos[0])
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(P, H, args.cpu_seconds, raw=raw, qpos=qpos, qvel=qvel, noise_scale=float(np.sqrt(cov)))
    print(json.dumps(out), flush=True)


def counted_flops(H):
    """algorithmic FLOPs per particle-step, counted (not estimated) by running the instrumented build of
    the oracle (oracle/flop_count.cpp: reacher_ref.c compiled with a counting scalar) on a sample of this workload."""
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    from oracle.physics_ref import count_flops
    rs = np.random.RandomState(123)
    noise = rs.standard_normal((64, H, 7))
    for t in range(2, H):
        noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
    d = count_flops(reacher7dof_raw().to_flat(), np.zeros(7), np.zeros(7), np.array([0.1, 0.1, 0.1]),
                    np.zeros((H, 7)), noise)
    d.pop("rew")
    return d


def profile_figure(name, dtype, P, H):
    """A per-launch figure that needs PMC counters (separate rocprofv3 passes, MI355X_MICROARCH.md): measured
    offline for the headline shape and kept under profiles/; other shapes report null."""
    for rnd in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_%s_%dx%d.json" % (rnd, name, dtype, P, H))
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f), os.path.relpath(path, ROOT)
    return None, None


def main():
    args = parse()
    if args.workload != "reacher":
        if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
            raise SystemExit("--workload %s runs on one GPU" % args.workload)
        return tree_workload(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch --gpus %d through torch.distributed.run (one process per GPU)" % args.gpus)
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
        from mjmpc_amd.control._device import TorchDistComm
        comm = TorchDistComm()

    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw

    P_loc, H, A = args.particles, args.horizon, 7
    P_tot = P_loc * world
    eng = ArmRolloutEngine(reacher7dof_raw(), device=local, dtype=args.dtype)
    ctrl = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=A, horizon=H, init_cov=1.0, base_action="null",
                lam=0.01, num_particles=P_tot, step_size=1.0, alpha=1, gamma=1.0, n_iters=1,
                action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0],
                seed=123, noise_mode={"mt19937": "device_mt19937"}.get(args.noise, args.noise), noise_dtype=args.dtype,
                device=local, comm=comm)
    base_fn = make_device_rollout_fn(eng)
    ev = []

    def rollout_fn(num_particles, horizon, mean, noise, mode):      # events bracket exactly the rollout launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = base_fn(num_particles, horizon, mean, noise, mode)
        e1.record()
        ev.append((e0, e1))
        return out

    rollout_fn.accepts_device = True
    graphed = (not args.no_graph and args.noise in ("device", "mt19937") and (world == 1 or args.backend == "nccl"))
    ctrl.rollout_fn = base_fn if graphed else rollout_fn
    ctrl.set_sim_state_fn = lambda s: None          # the "real" arm lives on the device (step_state)
    eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])))
    state = {"resident": True}

    if graphed:
        ctrl.enable_graph(post_step=eng.step_state)      # the env step is captured with the iteration

    def control_step():
        action, _ = ctrl.optimize(state)
        if not graphed:
            eng.step_state(action)          # (in graph mode the env step is part of the captured iteration)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        control_step()
    ev.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        control_step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if graphed:
        # inside a replayed graph there is nothing to bracket from the host: time the dominant kernel right here
        # with events on its launch stream - 20 back-to-back launches of the SAME entry point the captured
        # iteration uses (mjmpc_arm_rollout_fused: noise filter + cost-to-go fused in) on the run's own buffers
        noise_t = ctrl.dev._rec[("noise_mt" if args.noise == "mt19937" else "noise", args.dtype)]
        fused = args.noise == "device"          # (the MT19937 sampler hands over filtered samples: plain entry point)
        coeffs = ctrl.dev.record("coeffs", 3)

        def launch():
            if fused:
                eng.rollout_fused(P_loc, H, ctrl.dev.mean, noise_t, coeffs, ctrl.dev.gseq)
            else:
                eng.rollout_device(P_loc, H, ctrl.dev.mean, noise_t)

        n_t = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch()
        e0.record()
        for _ in range(n_t):
            launch()
        e1.record()
        torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / n_t
    else:
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
    _, nobs = eng.step_state(np.zeros(A))
    dist_to_target = float(torch.linalg.norm(nobs[17:20]).item())
    fails = eng.solver_failures()

    # HBM bytes of one launch of the dominant kernel from the PMC counters (separate rocprofv3 passes,
    # FETCH_SIZE x2 on gfx950, calibrated on a known copy): measured offline, kept under profiles/
    tj, traffic_src = profile_figure("traffic", args.dtype, P_loc, H)
    traffic = tj["traffic_bytes_per_launch"] if tj else None
    ij, issue_src = profile_figure("valu_issue", args.dtype, P_loc, H)

    psteps = P_tot * H * ctrl.n_iters * args.steps
    s = 8 if args.dtype == "f64" else 4
    b_alg = (3 * A + 2) * s                       # delta in, action + cost out, action + cost re-read
    achieved = b_alg * P_loc * H / (kern_ms * 1e-3) / 1e9
    valu = None
    fl = None
    if rank == 0:
        try:
            fl = counted_flops(H)
        except Exception as e:          # the counting build is measurement infrastructure: never lose the line over it
            valu = {"bound": "valu", "error": "FLOP-counting oracle build unavailable: %s" % (e,)}
    if fl is not None:
        peak_tf = FP64_VALU_PEAK_TF if args.dtype == "f64" else FP32_VALU_PEAK_TF
        tf = fl["flops"] * P_loc * H / (kern_ms * 1e-3) / 1e12
        valu = {"bound": "valu", "flops_per_particle_step": fl["flops"],
                "counted": {k: fl[k] for k in ("add", "mul", "div", "sqrt", "trig", "cmp")},
                "counted_by": "oracle/flop_count.cpp (oracle/reacher_ref.c compiled with a counting scalar), 64 x H sample "
                              "of this workload; an FMA counts as 2, compares are listed but not counted",
                "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf,
                # share of the chip's VALU issue slots the kernel's own instruction stream fills (SQ_INSTS_VALU x
                # cycles per wave64 instruction / (SIMDs x duration x clock)): from the SQ PMC pass kept under profiles/
                "issue_frac": ij["valu_issue_frac"] if ij else None, "issue_frac_source": issue_src}
    out = {
        "metric": "particle-steps/sec (reacher_7dof-v0 MPPI %dp x H%d per GPU, control loop incl. noise, rollout, update, shift)" % (P_loc, H),
        "value": psteps / dt, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "reacher_7dof-v0 MPPI lam=0.01 H=%d, %d particles per GPU (%d total), frame_skip 2, "
                               "filter [0.25,0.8,0], closed loop from qpos0 to target [0.1,0.1,0.1]" % (H, P_loc, P_tot),
                   "noise": args.noise, "particles_per_gpu": P_loc, "horizon": H,
                   "ranks_seen": dist.get_world_size() if world > 1 else 1,
                   "backend": (dist.get_backend() if world > 1 else None),
                   "launch": "hipGraph replay" if (graphed and not getattr(ctrl, "graph_fallback", False)) else "eager"},
        "control_loop_hz": args.steps / dt,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": ("offline PMC passes, %s" % traffic_src) if traffic_src else None,
                     "valu": valu,
                     "alg_bytes_per_launch": b_alg * P_loc * H,
                     "kernel": "arm_rollout_kernel<%s>" % ("double" if args.dtype == "f64" else "float"),
                     "kernel_ms": kern_ms, "kernel_entry": ("mjmpc_arm_rollout_fused" if (graphed and args.noise == "device") else "mjmpc_arm_rollout"),
                     "alg_bytes_per_particle_step": b_alg,
                     "note": "latency/VALU-bound path (SURVEY 8d): ~160 counted FLOP per algorithmic byte, HBM fraction is small by "
                             "construction; `valu` is the roofline that binds"},
        "solver_failures": fails, "final_distance_to_target": dist_to_target,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P_loc, H, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: reacher_7dof-v0 MPPI, 4096 particles x H=32 per GPU (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8 --steps 50 --warmup 5          # starts its own ranks (one fresh child process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one control iteration, i.e. one ``Controller.optimize()``: sample noise ->
roll out every particle for H env steps (frame_skip MuJoCo substeps each) -> distribution update -> shift -> D2H of
the action, followed by stepping the "real" env with that action.  Everything but the action stays
in HBM.  N > 1: each rank owns a contiguous block of particles (the reference's worker mapping,
subproc_vec_env.py:161-168); the per-GPU record of the update is all-gathered (RCCL over xGMI) once per iteration
(twice for CEM: cost-to-go for the global elite threshold, then the elite moments).

    --scaling weak    (default) --particles is the block of ONE GPU, the population grows with N
    --scaling strong  --particles is the whole population, divided over the N GPUs
    --controller {mppi,cem,dmd}; --workload {reacher,half_cheetah,swimmer,hand24,pen_hand}

BASELINE.json configurations: 2 = ``--particles 1024``; 3 = default (sawyer.xml is the vendored 7-dof arm);
4 = ``--controller cem --particles 16384 --scaling strong --gpus 4``;
5 = ``--workload hand24 --controller dmd --particles 65536 --horizon 64 --scaling strong --gpus 8`` (stand-in model).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the rollout kernel, timed live with
events on the launch stream; the HBM block SURVEY 8d defines plus `valu`, the roofline that actually binds:
FLOPs per particle-step COUNTED by the instrumented oracle build, oracle/flop_count.cpp, beside the kernel's own count
and issue fraction from the SQ counter passes kept under profiles/), `cpu_baseline` (oracle/ on the host cores, N = 1
only) and, for the fused two-launch iteration on one GPU, `pipelined`: the same closed loop with the next iteration
enqueued before the host waits for the current action (reported beside the headline, never as it).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6       # vector FP64; FP32 157.3
FP32_VALU_PEAK_TF = 157.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=4096,
                    help="particles PER GPU (--scaling weak, the default) or in total (--scaling strong)")
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--noise", choices=["device", "mt19937", "host"], default="device",
                    help="device: Philox on the GPU; mt19937: the reference's own numpy stream regenerated on the GPU "
                         "(seed-identical particles); host: numpy on the host, uploaded")
    ap.add_argument("--workload", choices=["reacher", "half_cheetah", "swimmer", "hand24", "pen_hand", "cartpole", "tray", "door", "gripper"],
                    default="reacher",
                    help="reacher: the BASELINE.json headline (default).  The others run the same loop on the tree engine "
                         "(SURVEY 8f rank 4: the reference's vendored HalfCheetah / Swimmer models, the synthetic 24-dof hand, "
                         "and pen_hand: a 6-dof pen on that hand - position servos, capsule-capsule contacts with friction "
                         "cones, pen-v0's shape of reward); cartpole / tray / door: the round-4 synthetic MJCF models of the "
                         "general kernel instantiation (friction loss; free joint + boxes; equality + static geoms)")
    ap.add_argument("--controller", choices=["mppi", "cem", "dmd"], default="mppi",
                    help="mppi (headline); cem: full covariance, elite_frac 0.1 (BASELINE config 4); dmd: DMD-MPC (config 5)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-mono", action="store_true",
                    help="keep the control iteration in its separate launches (sampler, rollout, two update launches, env step) "
                         "instead of the one-launch iteration (mjmpc_arm_mppi_step)")
    ap.add_argument("--no-tape", action="store_true",
                    help="replay the captured iteration as a hipGraph instead of as the list of its library calls "
                         "(Controller.enable_graph(tape=False); one GPU)")
    ap.add_argument("--lookahead", action="store_true",
                    help="enqueue iteration k+1 before waiting for the action of iteration k (the real env lives on the device, so "
                         "nothing the host provides enters an iteration); off by default: every optimize() then starts after the "
                         "previous action has reached the host")
    ap.add_argument("--process-warmup", type=int, default=60,
                    help="throw-away control steps before controller and env are reset and the W warm-up steps begin, so that "
                         "the timed steps do not pay for a fresh process (idle clocks, first-touch, lazy runtime initialisation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--engine", choices=["auto", "tree"], default="auto",
                    help="synthetic workloads: 'auto' = the arm kernels where the model fits them (cart-pole), 'tree' = always the tree engine")
    ap.add_argument("--ab-timeout", type=int, default=180, help="--collectives both: seconds the A/B of the exchange paths may take "
                                                                "before the line is printed without it")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)          # test knobs: gloo on one GPU
    ap.add_argument("--collectives", choices=["auto", "library", "torch", "both"], default="both",
                    help="N > 1: which path carries the control iteration's exchanges - libmjmpc_amd.so's own RCCL communicator "
                         "(direct launches / launch tape) or torch.distributed (hipGraph replay).  'both' (default) times the "
                         "headline on the library's path and the same loops again on torch's, reported beside it as "
                         "`collectives_ab` (DESIGN 6); 'auto' = the library's where it is available")
    ap.add_argument("--device", type=int, default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as a FRESH child process tree
    (``python -m torch.distributed.run``, one process per GPU) and hand its exit code back.  This process has not
    touched the GPU (torch is not even imported yet) and never replaces itself with another program."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(P, H, budget_s, raw=None, qpos=None, qvel=None, noise_scale=1.0):
    """The FP64 C oracle (kind "port") on this box's host cores: same model, same start state, same
    noise recipe; a bounded sample of the workload (batches of >= 32 particles per thread x H until the budget).
    `raw` (default: the reacher) lets tools/bench_configs.py time the other models through this same leg."""
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    from oracle.physics_ref import RefArm, build_flags, threads
    raw = raw or reacher7dof_raw()
    arm = RefArm(raw.to_flat(), native=True)       # -O3 -march=native build of the oracle source for THIS host
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
    cores = threads(avail, native=True)
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
    threads(1, native=True)
    n1, t1 = 0, time.time()
    while time.time() - t1 < 0.2 * budget_s:
        arm.rollout(q0, v0, tgt, mean, noise[:64], want_obs=False)
        n1 += 64
    dt1 = time.time() - t1
    threads(cores, native=True)
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
                      "one thread: %d particles in %.1f s; oracle built with `%s`%s"
                      % (n, H, cores, dt, n1, dt1, build_flags(), quota)}


def counted_flops(raw, H, A, qpos, qvel, target, noise_scale, filtered):
    """algorithmic FLOPs per particle-step, counted (not estimated) by running the instrumented build of
    the oracle (oracle/flop_count.cpp: reacher_ref.c compiled with a counting scalar) on a sample of this workload."""
    from oracle.physics_ref import count_flops
    rs = np.random.RandomState(123)
    noise = noise_scale * rs.standard_normal((64 if filtered else 32, H, A))
    if filtered:
        for t in range(2, H):
            noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
    d = count_flops(raw.to_flat(), qpos, qvel, target, np.zeros((H, A)), noise)
    d.pop("rew", None)
    return d


def profile_figure(name, dtype, P, H, prefix=""):
    """A per-launch figure that needs PMC counters (separate rocprofv3 passes, MI355X_MICROARCH.md): measured
    offline and kept under profiles/ (the headline shape; since round 5 the reacher at 16384 / 65536 particles and the cart-pole,
    door, tray and gripper workloads too); other shapes report null."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s%s_%s_%dx%d.json" % (rnd, prefix, name, dtype, P, H))
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f), os.path.relpath(path, ROOT)
    return None, None


# ---------------------------------------------------------------------------------------------------------- workloads
def launch_kind_of(ctrl, graphed):
    """How a wired controller issues its control iteration (the `launch` entries of the JSON line)."""
    kind = "hipGraph replay" if (graphed and not getattr(ctrl, "graph_fallback", False)) else "eager"
    if graphed and getattr(ctrl, "_graph", None) == "direct":
        kind = "two kernels per iteration, launched directly"
    elif graphed and kind != "eager" and getattr(ctrl, "launch_mode", None):
        kind = ctrl.launch_mode             # "hipGraph replay" or "launch tape (n calls, k kernels)"
    return kind


def make_workload(args, local, comm, P_tot):
    """Engine + controller + bookkeeping of one (workload, controller) pair."""
    from mjmpc_amd.control import CEM, DMDMPC, MPPI
    H = args.horizon
    w = {}
    if args.workload == "reacher":
        from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
        from mjmpc_amd.models.reacher7dof import reacher7dof_raw
        raw = reacher7dof_raw()
        eng = ArmRolloutEngine(raw, device=local, dtype=args.dtype)
        w.update(name="reacher_7dof-v0", lam={"mppi": 0.01, "dmd": 0.1}, cov=1.0, env=None,
                 kernel="arm_rollout_kernel", target=np.array([0.1, 0.1, 0.1]), frame_skip=2, nv=7,
                 tail="closed loop from qpos0 to target [0.1,0.1,0.1]")
        w["reset"] = lambda: eng.set_env_state(dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=w["target"]))
        w["reset"]()
    else:
        from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
        from mjmpc_amd.models.compile_tree import compile_tree
        start = None
        if args.workload == "hand24":
            from mjmpc_amd.models.hand24 import hand24_raw
            raw, env, name, lam = hand24_raw(), None, "hand_tree-v0 (synthetic 24-dof hand)", {"mppi": 0.05, "dmd": 0.1}
        elif args.workload == "pen_hand":
            from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
            raw, env, name, lam = pen_hand_raw(), None, "pen_hand-v0 (synthetic 6-dof pen in a 24-dof hand)", {"mppi": 0.05, "dmd": 0.1}
            start = holding_state()
        elif args.workload in ("cartpole", "tray", "door", "gripper"):
            from mjmpc_amd.models.synthetic import start_state, synthetic_raw
            raw, env = synthetic_raw(args.workload), None
            name = "%s_synthetic-v0 (mjmpc_amd/models/assets/%s.xml)" % (args.workload, args.workload)
            lam = {"mppi": 0.05, "dmd": 0.1}
            gen_start = start_state(args.workload, raw)
        else:
            from mjmpc_amd.envs import locomotion_env
            from mjmpc_amd.models.half_cheetah import half_cheetah_raw
            from mjmpc_amd.models.swimmer import swimmer_raw
            raw = dict(half_cheetah=half_cheetah_raw, swimmer=swimmer_raw)[args.workload]()
            env = dict(half_cheetah=locomotion_env.HalfCheetahEnv, swimmer=locomotion_env.SwimmerEnv)[args.workload](
                dtype=args.dtype, device=local)
            name, lam = dict(half_cheetah="HalfCheetah-v0", swimmer="Swimmer-v0")[args.workload], {"mppi": 0.2, "dmd": 0.2}
        # (round 6: models the serial-chain arm kernels take - the cart-pole: slide joints + dry friction, their extended-joint
        # build - run there; --engine tree keeps the general tree engine for A/B)
        from mjmpc_amd.envs import make_engine
        eng = (TreeRolloutEngine(raw, device=local, dtype=args.dtype) if (args.engine == "tree" or env is not None)
               else make_engine(raw, device=local, dtype=args.dtype))
        on_arm = not isinstance(eng, TreeRolloutEngine)
        if on_arm:
            name += " on the arm engine (extended-joint build)"
        w.update(name=name, lam=lam, cov=0.3, env=env, kernel="arm_rollout_kernel" if on_arm else "tree_rollout_kernel",
                 target=np.asarray(raw.target_pos, float),
                 frame_skip=raw.frame_skip, nv=compile_tree(raw).nv,
                 tail="closed loop (tree engine; not a BASELINE.json configuration%s)"
                      % ("" if args.workload not in ("hand24", "pen_hand") else "; stand-in for pen-v0, whose assets are absent"))
        if env is not None:
            env.reset(seed=123)                         # the reference's reset noise, then the engine owns the state
            st0 = env.get_env_state()
            w["reset"] = lambda: eng.set_env_state(st0)
            w["x0"] = float(st0["qpos"][0])
        elif args.workload in ("cartpole", "tray", "door", "gripper"):
            w["reset"] = lambda: eng.set_env_state(gen_start)
            w["start"] = gen_start
            w["cov"] = 0.01 if args.workload in ("tray", "gripper") else 0.3
        elif start is not None:
            st0 = dict(start, target_pos=np.asarray(raw.target_pos, float))
            w["reset"] = lambda: eng.set_env_state(st0)
            w["hold"] = start["qp"][6:].copy()          # servo targets of the start pose: the controller's initial mean
            w["cov"] = 0.01
        else:
            w["reset"] = eng.reset
        w["reset"]()
    A = eng.d_action
    eng.on_env_reset = "warn"         # (a benchmark reports the real env's resets - `env_resets` - instead of raising)
    base_reset = w["reset"]

    def make_ctrl(P_total, comm=comm):
        """A controller over P_total particles (this rank's block of them) on this workload's engine, and the reset
        that goes with it (``comm``: another communicator than the run's - the A/B of the exchange paths)."""
        kw = dict(d_state=eng.d_state, d_obs=eng.d_obs, d_action=A, horizon=H, num_particles=P_total, n_iters=1,
                  action_lows=eng.action_lows, action_highs=eng.action_highs, seed=123, base_action="null", gamma=1.0,
                  step_size=1.0, filter_coeffs=[0.25, 0.8, 0.0], init_cov=w["cov"],
                  noise_mode={"mt19937": "device_mt19937"}.get(args.noise, args.noise), noise_dtype=args.dtype,
                  device=local, comm=comm)
        if args.controller == "mppi":
            ctrl = MPPI(lam=w["lam"]["mppi"], alpha=1, **kw)
            desc = "MPPI lam=%g" % w["lam"]["mppi"]
        elif args.controller == "cem":
            ctrl = CEM(elite_frac=0.1, beta=0.1, cov_type="full", **kw)
            desc = "CEM full-cov elite_frac=0.1 beta=0.1"
        else:
            ctrl = DMDMPC(lam=w["lam"]["dmd"], beta=0.1, update_cov=False, cov_type="diagonal", **kw)
            desc = "DMD-MPC lam=%g" % w["lam"]["dmd"]
        reset = base_reset
        if "hold" in w:                 # position servos: the nominal control is the pose, not zero
            def reset(ctrl=ctrl, pose=w["hold"]):
                base_reset()
                ctrl.mean_action = np.tile(pose, (H, 1))
            ctrl.base_action = "repeat"
            reset()
        return ctrl, desc, reset

    ctrl, desc, w["reset"] = make_ctrl(P_tot)
    w["make_ctrl"] = make_ctrl
    w.update(raw=raw, eng=eng, ctrl=ctrl, A=A, desc=desc)
    return w


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))         # before anything touches the GPU
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
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
        comm = TorchDistComm(device=torch.device("cuda", local) if args.backend == "nccl" else None,
                             library_collectives={"torch": False, "library": True}.get(args.collectives))

    from mjmpc_amd.build import build_info
    from mjmpc_amd.control.controller import resident_state
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn

    H = args.horizon
    if args.scaling == "strong":
        if args.particles % world:
            raise SystemExit("--scaling strong: %d particles do not divide over %d GPUs" % (args.particles, world))
        P_tot, P_loc = args.particles, args.particles // world
    else:
        P_loc, P_tot = args.particles, args.particles * world
    w = make_workload(args, local, comm, P_tot)
    eng, ctrl, A, raw = w["eng"], w["ctrl"], w["A"], w["raw"]
    base_fn = make_device_rollout_fn(eng)
    last = {}

    def rollout_fn(num_particles, horizon, mean, noise, mode):      # (eager runs: keeps the noise buffer for the kernel timing)
        last["noise"] = noise
        return base_fn(num_particles, horizon, mean, noise, mode)

    rollout_fn.accepts_device = True
    state = {"resident": True}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def wire(ctrl):
        """Rollout callbacks + captured iteration of a controller on this workload's engine; returns (graphed, step)."""
        ctrl.rollout_fn = base_fn
        graphed = not args.no_graph and args.noise != "host" and ctrl._graph_capable()
        if not graphed:
            ctrl.rollout_fn = rollout_fn
            if hasattr(base_fn, "fused"):
                rollout_fn.fused = base_fn.fused
        ctrl.set_sim_state_fn = resident_state          # the "real" env lives on the device (step_state)
        if graphed:
            ctrl.enable_graph(post_step=eng.step_state, mono=not args.no_mono, lookahead=args.lookahead, tape=not args.no_tape)   # the env step is captured with the iteration

        def control_step():
            action, _ = ctrl.optimize(state)
            if not graphed:
                eng.step_state(action)          # (in graph mode the env step is part of the captured iteration)

        return graphed, control_step

    def timed_loop(ctrl, control_step, reset, process_warmup):
        """process warm-up, reset, W warm-up steps, K timed steps between barriers; MAX over ranks."""
        # Process warm-up (outside the W warm-up steps and the timed region): a fresh process runs its first control steps
        # up to 7 % slow - idle GPU clocks, first-touch of pinned and device buffers, lazily initialised runtime paths
        # (measured: 20 timed steps after 5 warm-up steps 0.218-0.225 ms per step in a fresh process, 0.205 ms for the same
        # 25 steps repeated in a warm one).  --process-warmup throw-away control steps run first; controller and env are
        # then reset, so that the W warm-up steps and the K timed steps are the closed loop from the initial state.
        if process_warmup > 0:
            for _ in range(process_warmup):
                control_step()
            sync()
            ctrl.reset()
            reset()
        for _ in range(args.warmup):
            control_step()
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
        return dt

    graphed, control_step = wire(ctrl)
    dt = timed_loop(ctrl, control_step, w["reset"], args.process_warmup)

    # where the closed loop ended up (read BEFORE the launches below)
    extra = {}
    on_arm = w["kernel"] == "arm_rollout_kernel"
    if on_arm:
        nv_ = w["nv"]
        _, nobs = eng.step_state(np.zeros(A))
        extra["final_distance_to_target"] = float(torch.linalg.norm(nobs[2 * nv_ + 3:2 * nv_ + 6]).item())
        st0 = eng.get_env_state()[0]                    # (the host mirror of the state the run STARTED from)
        qpos, qvel = (np.zeros(7), np.zeros(7)) if args.workload == "reacher" else (np.asarray(w["start"]["qp"], float), np.asarray(w["start"]["qv"], float))
    else:
        st = eng.get_state_device()
        qpos = st["qpos"] if "qpos" in st else st["qp"]
        qvel = st["qvel"] if "qvel" in st else st["qv"]
        if w["env"] is not None:
            extra["forward_progress_m"] = float(qpos[0]) - w["x0"]      # since the reset (whose noise moves qpos[0] too)
    # The same closed loop run once more with the NEXT iteration enqueued before the host waits for the current action (the env
    # lives on the device, so iteration k + 1 needs nothing from the host): reported beside the headline, never as it -
    # `value` is the reference's call pattern, one optimize() at a time.
    pipelined = None
    if graphed and getattr(ctrl, "_mono", False) and world == 1 and not args.lookahead:
        ctrl._lookahead = True
        for _ in range(args.process_warmup):    # (this call pattern gets its own process warm-up, like the headline's)
            control_step()
        sync()
        ctrl.reset()                    # the same closed loop from the same initial state, W warm-up + K timed steps
        w["reset"]()
        for _ in range(args.warmup):
            control_step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            control_step()
        sync()
        dtp = time.perf_counter() - t1
        ctrl._lookahead = False
        pipelined = {"ms_per_step": dtp / args.steps * 1e3, "value": P_tot * H * ctrl.n_iters * args.steps / dtp,
                     "what": "Controller.enable_graph(lookahead=True): iteration k+1 is in the queue while the host reads action k"}
    # The dominant kernel, timed with events on its launch stream: 20 back-to-back launches of the SAME entry point the
    # control iteration uses (inside a replayed graph there is nothing to bracket from the host) on the run's own buffers
    fused_entry = (args.noise == "device" and hasattr(base_fn, "fused") and not ctrl.use_zero_control_seq
                   and (ctrl._fused_capable() or ctrl._wants_q0()))
    noise_t = ctrl.dev._rec.get(("noise_mt" if args.noise == "mt19937" else "noise", args.dtype))
    if noise_t is None:
        noise_t = last.get("noise")
    coeffs = ctrl.dev.record("coeffs", 3)

    mono = graphed and getattr(ctrl, "_mono", False)
    sampled_entry = graphed and hasattr(ctrl, "_cem_in_kernel") and ctrl._cem_in_kernel()
    if mono:
        chol_t, coeffs_t, _ = ctrl.dev.prepare_noise(ctrl._cov_host, ctrl.filter_coeffs)
        t_step = torch.zeros(1, dtype=torch.int64, device="cuda")
        t_rec = torch.zeros(2 + H * A, dtype=torch.float64, device="cuda") if world > 1 else None

    def launch(shift=-2):
        if mono:        # the fused iteration's rollout launch (shift -2) / both launches, from the state the run ended in
            eng.mppi_step(P_loc, H, ctrl.dev.mean, ctrl.dev.mean_alt, ctrl.dev.gseq, coeffs_t, chol_t, ctrl.seed_val, 0, 0, t_step, ctrl.lam,
                          ctrl.step_size, shift, record=t_rec if shift != -2 else None, env_step=False)
        elif sampled_entry:     # CEM on a small launch: the rollout kernel draws its own full-covariance samples
            eng.rollout_sampled(P_loc, H, ctrl.dev.mean, ctrl.dev.gseq, coeffs, ctrl.dev.record("chol", A * A), True,
                                ctrl.seed_val, 0, 0, ctrl._step_dev)
        elif fused_entry:
            eng.rollout_fused(P_loc, H, ctrl.dev.mean, noise_t, coeffs, ctrl.dev.gseq)
        else:
            eng.rollout_device(P_loc, H, ctrl.dev.mean, noise_t)

    n_t = 20 if w["kernel"] == "arm_rollout_kernel" else 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launch()
    e0.record()
    for _ in range(n_t):
        launch()
    e1.record()
    torch.cuda.synchronize()
    kern_final_ms = e0.elapsed_time(e1) / n_t
    both_ms = None
    if mono:            # ... and rollout + finish launch together (update, action, shift; no env step)
        launch(0)
        e0.record()
        for _ in range(n_t):
            launch(0)
        e1.record()
        torch.cuda.synchronize()
        both_ms = e0.elapsed_time(e1) / n_t

    # ... and over the states the timed loop VISITED: the same closed loop once more from the reset (it is deterministic), the
    # entry point launched and timed from up to 25 of the timed steps' states - `kernel_ms` is their mean, what the roofline
    # figures below are quoted on; the end state's figure above stays beside it (a contact-rich end state costs more than
    # the loop's average step: with it alone a line could show a kernel slower than the step that contains it)
    ctrl.reset()
    w["reset"]()
    for _ in range(args.warmup):
        control_step()
    n_v = 4 if w["kernel"] == "arm_rollout_kernel" else 2
    stride = max(1, args.steps // 25)
    pairs = []
    for k in range(args.steps):
        if k % stride == 0:
            torch.cuda.synchronize()
            launch()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n_v):
                launch()
            b.record()
            pairs.append((a, b))
        control_step()
    sync()
    kern_ms = float(np.mean([a.elapsed_time(b) / n_v for a, b in pairs]))

    # N > 1, weak scaling (the default the driver runs): the SAME invocation also answers the other reading of the metric
    # ("4096 particles x H=32 reported at 1, 2, 4 and 8": the reference's num_particles is a total,
    # examples/example_mpc.py:78-79) - a second controller over --particles IN TOTAL, this rank's block = particles / N,
    # on the same engine, timed the same way.  Reported beside the headline as `strong`, never as `value`.
    import gc

    def side_loop(P_total, comm_):
        """The same closed loop once more with a controller of its own (P_total particles in all, exchanges through comm_):
        (seconds for the K timed steps, launch kind)."""
        ctrl._graph = None                  # graphs holding RCCL kernels of the loop before go first
        gc.collect()
        torch.cuda.synchronize()
        ctrl_s, _, reset_s = w["make_ctrl"](P_total, comm_)
        graphed_s, step_s = wire(ctrl_s)
        reset_s()
        dts = timed_loop(ctrl_s, step_s, reset_s, min(args.process_warmup, 20))
        kind = launch_kind_of(ctrl_s, graphed_s)
        ctrl_s._graph = None
        del ctrl_s
        gc.collect()
        torch.cuda.synchronize()
        return dts, kind

    strong = None
    strong_ok = world > 1 and args.scaling == "weak" and not (args.particles % world or (args.particles // world) % 8)
    if world > 1 and args.scaling == "weak":
        if not strong_ok:
            strong = {"skipped": "%d particles do not split into blocks of a multiple of 8 over %d GPUs" % (args.particles, world)}
        else:
            dts, kind_s = side_loop(args.particles, comm)
            strong = {"ms_per_step": dts / args.steps * 1e3, "value": args.particles * H * ctrl.n_iters * args.steps / dts,
                      "unit": "particle-steps/s", "control_loop_hz": args.steps / dts, "particles_total": args.particles,
                      "particles_per_gpu": args.particles // world, "scaling": "strong",
                      "launch": kind_s, "collectives": comm.collectives,
                      "what": "the same closed loop with --particles as the TOTAL population, split over the ranks "
                              "(subproc_vec_env.py:161-168); same steps / warmup, MAX over ranks"}
    fails = eng.solver_failures()

    # HBM bytes of one launch of the dominant kernel from the PMC counters (separate rocprofv3 passes,
    # FETCH_SIZE x2 on gfx950, calibrated on a known copy): measured offline, kept under profiles/
    prefix = "" if args.workload == "reacher" else args.workload + "_"
    tj, traffic_src = profile_figure("traffic", args.dtype, P_loc, H, prefix)
    traffic = tj["traffic_bytes_per_launch"] if tj else None
    ij, issue_src = profile_figure("valu_issue", args.dtype, P_loc, H, prefix)

    psteps = P_tot * H * ctrl.n_iters * args.steps
    s = 8 if args.dtype == "f64" else 4
    b_alg = (3 * A + 2) * s                       # delta in, action + cost out, action + cost re-read
    achieved = b_alg * P_loc * H / (kern_ms * 1e-3) / 1e9
    valu = None
    noise_scale = float(np.sqrt(w["cov"]))
    if rank == 0:
        try:
            fl = counted_flops(raw, H, A, qpos, qvel, w["target"], noise_scale, args.workload == "reacher")
            peak_tf = FP64_VALU_PEAK_TF if args.dtype == "f64" else FP32_VALU_PEAK_TF
            tf = fl["flops"] * P_loc * H / (kern_ms * 1e-3) / 1e12
            valu = {"bound": "valu", "flops_per_particle_step": fl["flops"],
                    "counted": {k: fl[k] for k in ("add", "mul", "div", "sqrt", "trig", "cmp") if k in fl},
                    "counted_by": "oracle/flop_count.cpp (oracle/reacher_ref.c compiled with a counting scalar) on a sample of "
                                  "this workload; an FMA counts as 2, compares are listed but not counted"
                                  + ("" if args.workload == "reacher" else
                                     "; the ORACLE's formulation (dense Jacobian-built mass matrix and Cholesky), which on a tree "
                                     "does more arithmetic than the kernel's sparse one - see kernel_flops_per_particle_step for "
                                     "the kernel's own count"),
                    "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf,
                    # share of the chip's VALU issue slots the kernel's own instruction stream fills (SQ_INSTS_VALU x
                    # cycles per wave64 instruction / (SIMDs x duration x clock)) and the FLOPs the kernel itself executes
                    # (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64/F32 x active lanes): from the SQ PMC passes kept under profiles/
                    "issue_frac": ij.get("valu_issue_frac") if ij else None,
                    "kernel_flops_per_particle_step": ij.get("kernel_flops_per_particle_step") if ij else None,
                    "kernel_flops_frac": (ij["kernel_flops_per_particle_step"] * P_loc * H / (kern_ms * 1e-3) / 1e12 / peak_tf
                                          if ij and ij.get("kernel_flops_per_particle_step") else None),
                    "issue_frac_source": issue_src}
        except Exception as e:          # the counting build is measurement infrastructure: never lose the line over it
            valu = {"bound": "valu", "error": "FLOP-counting oracle build unavailable: %s" % (e,)}
    launch_kind = launch_kind_of(ctrl, graphed)
    out = {
        "metric": "particle-steps/sec (%s %s %dp x H%d per GPU, control loop incl. noise, rollout, update, shift)"
                  % (w["name"], args.controller.upper() if args.controller != "dmd" else "DMD-MPC", P_loc, H),
        "value": psteps / dt, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s %s H=%d, %d particles per GPU (%d total), frame_skip %d, %d dofs, filter [0.25,0.8,0], %s"
                               % (w["name"], w["desc"], H, P_loc, P_tot, w["frame_skip"], w["nv"], w["tail"]),
                   "controller": args.controller, "noise": args.noise, "particles_per_gpu": P_loc, "particles_total": P_tot,
                   "horizon": H, "ranks_seen": dist.get_world_size() if world > 1 else 1,
                   "backend": (dist.get_backend() if world > 1 else None),
                   # which path carried the exchanges of the timed loop, what making the library's communicator took, and
                   # whether any rank fell back from it (then EVERY rank did: TorchDistComm decides by all-reduce)
                   "collectives": (comm.collectives if world > 1 else None),
                   "comm_init_s": (round(comm.init_seconds, 4) if world > 1 else None),
                   "collectives_fallback": ((comm.why_fell_back or True) if (world > 1 and comm.fell_back) else False),
                   "collectives_per_step": (0 if world == 1 else (2 if args.controller == "cem" else 1)),
                   "launch": launch_kind + (", rollout + record launches, all-gather, combine, env step" if (mono and world > 1) else "")
                             + (", next iteration enqueued ahead" if (mono and args.lookahead and world == 1) else ""),
                   "process_warmup_steps": args.process_warmup,
                   # which tuning alternative each HIP source was compiled with (mjmpc_amd/build.py: the first that this
                   # compiler accepts; profiles/ were measured with alternative 0 of every source)
                   "build": {k: (v["flags"] or "default") + ("" if v.get("alternative", 0) == 0 else " [FALLBACK alternative %d of %d]" % (v["alternative"], v["of"]))
                             for k, v in build_info().get("sources", {}).items() if v.get("of", 0) > 0 or v.get("alternative", 0) != 0}},
        "control_loop_hz": args.steps / dt,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": ("offline PMC passes, %s" % traffic_src) if traffic_src else None,
                     "valu": valu,
                     "alg_bytes_per_launch": b_alg * P_loc * H,
                     "kernel": "%s<%s>" % (w["kernel"], "double" if args.dtype == "f64" else "float"),
                     "kernel_ms": kern_ms,
                     "kernel_ms_how": "mean over %d states the timed loop visited (the closed loop run once more from the reset): HIP events "
                                      "around %d back-to-back launches of the iteration's entry point on the run's own buffers from each" % (len(pairs), n_v),
                     "kernel_ms_final_state": kern_final_ms,
                     "kernel_ms_final_state_how": "%d back-to-back launches from the state the run ended in (a contact-rich end state can "
                                                  "cost more than the loop's average step)" % n_t,
                     "kernel_entry": ("mjmpc_arm_mppi_step, launch 1 of 2 (sampling + rollout + cost-to-go + per-workgroup softmax records); "
                                      "with launch 2 (arm_mppi_finish_kernel: update + action + shift, here without its env step): "
                                      "%.4f ms" % both_ms
                                      if mono else ("mjmpc_arm_rollout_sampled (Philox draws + full-covariance colouring in the kernel)" if sampled_entry
                                                    else ("mjmpc_arm_rollout" if on_arm else "mjmpc_tree_rollout")
                                                    + ("_fused" if fused_entry else ""))),
                     "alg_bytes_per_particle_step": b_alg,
                     "note": "latency/VALU-bound path (SURVEY 8d): >100 counted FLOP per algorithmic byte, HBM fraction is small by "
                             "construction; `valu` is the roofline that binds"},
        "solver_failures": fails,
    }
    if hasattr(eng, "env_resets"):
        # resets of the device-resident REAL env (the reference raises MujocoException there; bench.py reports instead)
        out["env_resets"] = eng.env_resets()
    if hasattr(eng, "diverged_substeps"):
        # particle-substeps in which MuJoCo would have reset the simulation (NaN / > 1e10 in qpos, qvel, qacc) and the kernel did:
        # such particles go on from qpos0 with finite costs (engine.set_reset_returns('inf') gives them +inf instead)
        # in the update; apart from solver_failures (= a finite problem the iteration cap ended)
        out["diverged_particle_substeps"] = eng.diverged_substeps()
    if pipelined:
        out["pipelined"] = pipelined
    if strong:
        out["strong"] = strong
    out.update(extra)
    # N > 1, --collectives both: the same loops again with the exchanges on the OTHER path - torch.distributed's collectives in
    # a replayed hipGraph where the headline ran the library's own RCCL communicator between direct launches - so
    # that ONE invocation of the driver's scaling run says which is faster on real links.  Never `value`.
    # The headline, `strong` and everything else of the line are COMPLETE at this point; the A/B runs behind a watchdog: if it
    # has not finished after --ab-timeout seconds (a path that has never run at N > 1 on hardware may hang in a collective),
    # rank 0 prints the line without it and every rank ends its own process with code 0 - an exit, never a re-exec.
    collectives_ab = None
    if world > 1 and args.collectives == "both" and args.backend == "nccl" and getattr(comm, "lib_collectives", False):
        import threading
        printed = threading.Lock()

        def give_up():
            if rank == 0 and printed.acquire(blocking=False):
                out["collectives_ab"] = {"skipped": "the A/B of the exchange paths did not finish within %d s" % args.ab_timeout}
                print(json.dumps(out), flush=True)
            sys.stderr.write("bench.py: rank %d: the collectives A/B exceeded %d s; ending this rank\n" % (rank, args.ab_timeout))
            sys.stderr.flush()
            os._exit(0)

        watchdog = threading.Timer(args.ab_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
                from mjmpc_amd.control._device import TorchDistComm as _Comm
                comm_t = _Comm(device=torch.device("cuda", local), library_collectives=False)
                dtw, kind_w = side_loop(P_tot, comm_t)
                collectives_ab = {
                    "library RCCL": {"weak_ms_per_step": dt / args.steps * 1e3,
                                     "strong_ms_per_step": strong["ms_per_step"] if strong and "ms_per_step" in strong else None},
                    "torch.distributed": {"weak_ms_per_step": dtw / args.steps * 1e3, "strong_ms_per_step": None, "launch": kind_w},
                    "what": "the headline's loop (and `strong`) with the control iteration's exchanges on each path; "
                            "the headline `value` is the library's"}
                if strong_ok:
                    dts_t, _ = side_loop(args.particles, comm_t)
                    collectives_ab["torch.distributed"]["strong_ms_per_step"] = dts_t / args.steps * 1e3
                comm_t.close()
        except Exception as e:      # (this rank alone: the others are in collectives this one will not join - the watchdog ends them)
            collectives_ab = {"error": "%s: %s" % (type(e).__name__, e)}
        watchdog.cancel()
        if not printed.acquire(blocking=False):
            return                  # (the watchdog is printing / has printed)
    if collectives_ab:
        out["collectives_ab"] = collectives_ab
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P_loc, H, args.cpu_seconds, raw=raw, qpos=qpos, qvel=qvel,
                                               noise_scale=noise_scale)
        print(json.dumps(out), flush=True)
    if world > 1:
        # captured graphs that hold RCCL kernels go before the group does (its watchdog thread otherwise races the
        # interpreter's shutdown: an abort at exit, one run in a few, after the line has been printed)
        import gc
        ctrl._graph = None
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        if comm is not None:
            comm.close()            # (the library's own RCCL communicator, TorchDistComm.lib_collectives)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

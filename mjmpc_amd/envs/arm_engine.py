"""GPU rollout engine for arm models - the drop-in for the reference's ``SubprocVecEnv``.

The reference fans particles out to forked CPU workers that each run ``GymEnvWrapper.rollout``
(mjmpc/envs/vec_env/subproc_vec_env.py:128-186, mjmpc/envs/gym_env_wrapper.py:89-156).  Here the
whole particle set is one HIP kernel launch (mjmpc_amd/csrc/arm_rollout.hip) reached through the C
ABI; this class keeps the reference's method names, argument meaning and return layout so that
the closures of examples/example_mpc.py:112-155 work unchanged:

    sim_env = ArmRolloutEngine(reacher7dof_raw())
    controller.set_sim_state_fn = sim_env.set_env_state
    controller.rollout_fn = make_rollout_fn(sim_env)

torch is used only as the owner of device memory and streams.
"""
import ctypes
import time

import numpy as np

from .. import _lib
from ..models.compile import ArmModel, compile_arm, principal_inertia
from .seeding import np_random
from ..models.raw import RawModel

_DT = {"f32": (_lib.F32, np.float32), "f64": (_lib.F64, np.float64)}

from ._resets import EnvResetWatch, SimulationUnstableError  # noqa: F401

def _torch():
    import torch
    return torch


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class ArmRolloutEngine(EnvResetWatch):
    """One GPU's worth of particles for a compiled arm model (``reacher_7dof-v0``)."""
    _abi = "arm"

    def __init__(self, model, device=0, dtype="f64", num_shards=1):
        self.raw = model if isinstance(model, RawModel) else None
        if isinstance(model, RawModel):
            model = compile_arm(model)
        if not isinstance(model, ArmModel):
            raise TypeError("model must be a RawModel or a compiled ArmModel")
        if dtype not in _DT:
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.model = model
        self.dtype = dtype
        self._code, self._np = _DT[dtype]
        self.num_shards = int(num_shards)       # reported like the reference's num_cpu (infos['total_time'])
        self._lib = _lib.require_gpu()
        torch = _torch()
        self.device = torch.device("cuda", device)
        self._tdtype = torch.float32 if dtype == "f32" else torch.float64
        h = ctypes.c_void_p()
        blob = np.ascontiguousarray(model.blob, np.float64)
        _lib.check(self._lib.mjmpc_arm_create(blob.ctypes.data_as(_lib._dp), blob.size, device, ctypes.byref(h)))
        self._h = h
        self.d_action = model.nu
        self.d_obs = model.d_obs
        self.d_state = 3 * model.nv + 3 + 1     # qp, qv, qa, target_pos, timestep (reacher_env.py:81-85)
        self.action_lows = model.ctrl_lo.copy()
        self.action_highs = model.ctrl_hi.copy()
        self.closed = False
        self._buf = {}
        self.default_dyn_params = [dict() for _ in range(self.num_shards)]
        self.randomized_dyn_params = [dict() for _ in range(self.num_shards)]
        self.set_env_state(dict(qp=np.zeros(model.nv), qv=np.zeros(model.nv), qa=np.zeros(model.nv),
                                target_pos=model.target_default.copy(), timestep=0))

    # ------------------------------------------------------------------ reference-shaped API
    def set_env_state(self, state_dicts):
        """``SubprocVecEnv.set_env_state`` (subproc_vec_env.py:235-251): one dict, or a list holding
        one dict (every shard gets the same state).  Keys as reacher_env.py:81-85; ``qa`` and
        ``timestep`` do not influence a rollout and are ignored."""
        if isinstance(state_dicts, (list, tuple)):
            if len(state_dicts) not in (1, self.num_shards):
                raise AssertionError("num states should equal 1 (same for all envs) or 1 per env")
            if any(not _same_state(state_dicts[0], s) for s in state_dicts[1:]):
                return self._set_shard_states(state_dicts)
            state = state_dicts[0]
        else:
            state = state_dicts
        if getattr(self, "_per_shard_states", False):
            _lib.check(self._lib.mjmpc_arm_set_shard_states(self._h, None, 0, self._stream()))
            self._per_shard_states = False
        qp = np.ascontiguousarray(state["qp"], np.float64).reshape(-1)
        qv = np.ascontiguousarray(state["qv"], np.float64).reshape(-1)
        tg = np.ascontiguousarray(state["target_pos"], np.float64).reshape(-1)
        if qp.size != self.model.nv or qv.size != self.model.nv or tg.size != 3:
            raise ValueError("state has the wrong dimensions for this model")
        self._state = dict(qp=qp.copy(), qv=qv.copy(), target_pos=tg.copy())
        _lib.check(self._lib.mjmpc_arm_set_state(self._h, qp.ctypes.data_as(_lib._dp), qv.ctypes.data_as(_lib._dp),
                                                 tg.ctypes.data_as(_lib._dp), self._stream()))

    def _set_shard_states(self, state_dicts):
        nv = self.model.nv
        arr = np.zeros((self.num_shards, 19))
        for k, s in enumerate(state_dicts):
            arr[k, :nv] = np.asarray(s["qp"], float).reshape(-1)
            arr[k, 8:8 + nv] = np.asarray(s["qv"], float).reshape(-1)
            arr[k, 16:19] = np.asarray(s["target_pos"], float).reshape(-1)
        _lib.check(self._lib.mjmpc_arm_set_shard_states(self._h, arr.ctypes.data_as(_lib._dp), self.num_shards,
                                                        self._stream()))
        self._per_shard_states = True
        self._shard_state_list = [dict(qp=arr[k, :nv].copy(), qv=arr[k, 8:8 + nv].copy(), target_pos=arr[k, 16:19].copy())
                                  for k in range(self.num_shards)]

    def get_env_state(self):
        """One state dict - or, after a per-shard ``set_env_state``, one per shard (subproc_vec_env.py:253-256)."""
        states = self._shard_state_list if getattr(self, "_per_shard_states", False) else [self._state]
        return [dict(qp=st["qp"].copy(), qv=st["qv"].copy(), qa=np.zeros(self.model.nv),
                     target_pos=st["target_pos"].copy(), timestep=0) for st in states]

    def rollout(self, num_particles, horizon, mean, noise, mode="open_loop"):
        """``SubprocVecEnv.rollout``: numpy in, numpy out, reference layouts.
        Returns (obs, rew, act, done, info, next_obs); ``info`` is a list with one dict per shard."""
        t0 = time.time()
        out = self.rollout_device(num_particles, horizon, mean, noise, mode, want_obs=True)
        costs, act, obs, nobs = (x.to("cpu").numpy().astype(np.float64, copy=False) for x in out)
        done = np.zeros((num_particles, horizon))
        dt = time.time() - t0
        info = [{"total_time": dt} for _ in range(self.num_shards)]
        return obs, -costs, act, done, info, nobs

    def randomize_dynamics(self, param_dict, base_seed):
        """``SubprocVecEnv.randomize_dynamics`` (subproc_vec_env.py:304-312): shard i draws from
        ``np_random(base_seed + i*12345)`` a uniform value in ``m (1 +- noise)``, ``m = (1 + bias) * default``
        for every ``{param_id: {name: [noise_scale, bias_scale]}}`` entry (gym_env_wrapper.py:367-416) and
        from then on simulates its own model.  Supported: body_mass, body_inertia, dof_damping, geom_size
        (collision geoms), geom_friction (accepted, no effect: every contact here is frictionless condim 1),
        dof_frictionloss (the default is 0 and the randomization multiplicative: stays 0, the draw is consumed),
        sensor_noise (a known sensor's draw is consumed; no observation reads a sensor).
        Returns (default_params, randomized_params), one dict per shard."""
        if self.raw is None:
            raise ValueError("randomize_dynamics needs the engine to be built from a RawModel")
        base = self.model
        blobs = []
        for i in range(self.num_shards):
            rng, _ = np_random(int(base_seed) + i * 12345)
            defaults, rand = self.default_dyn_params[i], self.randomized_dyn_params[i]
            for param_id, entries in param_dict.items():
                for name, (noise_scale, bias_scale) in entries.items():
                    cur = defaults.setdefault(param_id, {}).get(name)
                    if cur is None:
                        cur = defaults[param_id][name] = self._default_param(param_id, name)
                    mean = (1.0 + bias_scale) * np.asarray(cur, float)
                    val = rng.uniform(mean - mean * noise_scale, mean + mean * noise_scale)
                    rand.setdefault(param_id, {})[name] = val
            if any(np.any(np.asarray(v) != 0) for v in rand.get("dof_frictionloss", {}).values()):
                raise NotImplementedError("a non-zero dof_frictionloss adds friction-loss constraint rows, which the "
                                          "arm kernel does not model")
            ov = {k: v for k, v in rand.items() if k not in ("geom_friction", "dof_frictionloss", "sensor_noise")}
            blobs.append(compile_arm(self.raw, overrides=ov, base=base).blob)
        blobs = np.ascontiguousarray(np.stack(blobs), np.float64)
        _lib.check(self._lib.mjmpc_arm_set_shard_models(self._h, blobs.ctypes.data_as(_lib._dp), self.num_shards))
        self.shard_blobs = blobs
        return self.default_dyn_params, self.randomized_dyn_params

    def _default_param(self, param_id, name):
        raw, m = self.raw, self.model
        names = [b.name for b in raw.bodies]
        if param_id == "body_mass":
            return float(m.body_mass[names.index(name)])
        if param_id == "body_inertia":
            return principal_inertia(m.body_inertia[names.index(name)])[0]
        if param_id == "dof_damping":
            return float(next(b.joint.damping for b in raw.bodies if b.joint is not None and b.joint.name == name))
        if param_id in ("geom_size", "geom_friction"):
            g = next(g for b in raw.bodies for g in b.geoms if g.name == name)
            if param_id == "geom_friction":
                return np.array([0.5, 0.1, 0.1])                      # sawyer.xml:6 default
            half = 0.5 * np.linalg.norm(np.asarray(g.b, float) - np.asarray(g.a, float)) if g.type == 2 else 0.0
            return np.array([g.radius, half, 0.0])
        if param_id == "dof_frictionloss":
            # RawJoint carries no frictionloss: every model this engine loads has MuJoCo's default 0 (sawyer.xml:5 sets
            # none).  The reference's randomization is multiplicative (gym_env_wrapper.py:409-411), so the randomized
            # value of a zero default is exactly 0: the draw is consumed, no friction-loss row ever appears.
            next(b for b in raw.bodies if b.joint is not None and b.joint.name == name)     # unknown joint -> error
            return 0.0
        if param_id == "sensor_noise":
            # (gym_env_wrapper.py:396-398 - model.sensor_noise: MuJoCo keeps the value for the user and adds no noise itself, and no
            # observation on the path reads a sensor: the draw is consumed, as in the reference, and changes nothing)
            if name not in raw.sensors:
                raise ValueError("no sensor named %r" % name)
            return float(raw.sensors[name])
        raise ValueError("Unknown dynamics field")

    def reset(self):
        self.set_env_state(dict(qp=np.zeros(self.model.nv), qv=np.zeros(self.model.nv),
                                target_pos=self.model.target_default.copy()))

    def close(self):
        if not self.closed:
            self._lib.mjmpc_arm_destroy(self._h)
            self._h = None              # later calls fail with "null engine" instead of touching freed memory
            self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ device-resident API
    def rollout_device(self, num_particles, horizon, mean, noise, mode="open_loop", want_obs=False,
                       want_actions=True):
        """Launch the fused rollout.  ``mean`` / ``noise`` may be numpy arrays or CUDA tensors.
        Returns device tensors (costs, actions, obs, next_obs); buffers are reused between calls."""
        if mode not in ("open_loop", "closed_loop_linear"):
            raise ValueError("unsupported rollout mode %r ('open_loop' or 'closed_loop_linear')" % (mode,))
        if num_particles % self.num_shards != 0:
            raise AssertionError("Number of particles must be divisible by number of shards")
        torch = _torch()
        P, H, A = int(num_particles), int(horizon), self.d_action
        closed = mode == "closed_loop_linear"
        mean_d = self._as_device(mean, torch.float64, (self.d_obs + 1, A) if closed else (H, A))
        noise_d = None if noise is None else self._as_device(noise, self._tdtype, (P, H, A))
        costs = self._buffer("costs", (P, H))
        act = self._buffer("act", (P, H, A)) if want_actions else None
        obs = self._buffer("obs", (P, H, self.d_obs)) if want_obs else None
        nobs = self._buffer("nobs", (P, H, self.d_obs)) if want_obs else None
        fn = self._lib.mjmpc_arm_rollout_cl if closed else self._lib.mjmpc_arm_rollout
        _lib.check(fn(self._h, self._code, P, H, _ptr(mean_d), _ptr(noise_d), _ptr(costs), _ptr(act), _ptr(obs),
                      _ptr(nobs), self._stream()))
        return costs, act, obs, nobs

    def rollout_fused(self, num_particles, horizon, mean, raw_noise, filter_coeffs, gamma_seq, q0_out=None):
        """Device-resident rollout with the noise filter and the discounted cost-to-go fused into the
        launch (``mjmpc_arm_rollout_fused``).  All arguments are CUDA tensors (``filter_coeffs`` may be
        None; ``q0_out``: a float64 [P] tensor the cost-to-go is written to instead of the engine's own
        buffer).  Returns (costs, actions, q0)."""
        torch = _torch()
        P, H, A = int(num_particles), int(horizon), self.d_action
        mean_d = self._as_device(mean, torch.float64, (H, A))
        noise_d = self._as_device(raw_noise, self._tdtype, (P, H, A))
        costs = self._buffer("costs", (P, H))
        act = self._buffer("act", (P, H, A))
        q0 = q0_out if q0_out is not None else self._buf.get("q0")
        if q0 is None or q0.shape[0] != P:
            q0 = self._buf["q0"] = torch.empty(P, dtype=torch.float64, device=self.device)
        _lib.check(self._lib.mjmpc_arm_rollout_fused(self._h, self._code, P, H, _ptr(mean_d), _ptr(noise_d),
                                                     _ptr(filter_coeffs), _ptr(gamma_seq), _ptr(costs), _ptr(act),
                                                     _ptr(q0), self._stream()))
        return costs, act, q0

    def mppi_step_supported(self, num_particles, horizon):
        """The fused iteration is built for launches of at most one wavefront per SIMD pair (two wavefronts per particle
        group, 4096 particles on 256 CUs) - the latency-bound regime, where the launches it saves matter; larger
        populations keep the separate sampler / rollout / update launches (their rollout kernel needs the registers and
        the LDS the fused one spends on sampling and on the action tile)."""
        torch = _torch()
        simds = 4 * torch.cuda.get_device_properties(self.device).multi_processor_count
        groups = (int(num_particles) + 7) // 8
        lds = 8 * (32 + 8 * int(horizon) * self.d_action * (8 if self.dtype == "f64" else 4) // 8)
        return 2 * groups <= simds and lds <= 40 * 1024

    def rollout_sampled(self, num_particles, horizon, mean, gamma_seq, filter_coeffs, chol, chol_full, seed, offset,
                        particle_offset, step_counter, q0_out=None):
        """A rollout whose samples are drawn in the kernel (``mjmpc_arm_rollout_sampled``): the Philox stream of
        ``DeviceUpdater.sample_noise``, coloured by the device-resident factor ``chol`` (``chol_full``: its whole lower
        triangle - CEM's adapting covariance), filtered on the fly.  Returns (costs, actions, q0) device tensors; nothing
        but the model, the state, the mean and the factor is read from memory."""
        torch = _torch()
        P, H, A = int(num_particles), int(horizon), self.d_action
        costs, act = self._buffer("costs", (P, H)), self._buffer("act", (P, H, A))
        q0 = q0_out
        if q0 is None:
            q0 = self._buf.get("q0")
            if q0 is None or q0.shape[0] != P:
                q0 = self._buf["q0"] = torch.empty(P, dtype=torch.float64, device=self.device)
        _lib.check(self._lib.mjmpc_arm_rollout_sampled(self._h, self._code, P, H, _ptr(mean), _ptr(gamma_seq), _ptr(filter_coeffs),
                                                       _ptr(chol), int(bool(chol_full)), int(seed) & (2 ** 64 - 1), int(offset),
                                                       int(particle_offset), _ptr(step_counter), _ptr(costs), _ptr(act), _ptr(q0),
                                                       self._stream()))
        return costs, act, q0

    def mppi_step(self, num_particles, horizon, mean, mean_out, gamma_seq, filter_coeffs, chol, seed, offset,
                  particle_offset, step_counter, lam, step_size, shift_mode, action_out=None, action_slots=None, record=None,
                  env_step=False, want_trajectories=False):
        """One whole control iteration in two launches (``mjmpc_arm_mppi_step``): sampling (the Philox stream of
        ``DeviceUpdater.sample_noise`` for a diagonal covariance), rollout, cost-to-go | softmax update of ``mean`` into
        ``mean_out`` (a tensor of its own), action read-out, shift, and - ``env_step`` - one step of the device-resident
        real env with that action.  All tensors are CUDA tensors (``mean`` / ``mean_out`` float64 (H,A), ``gamma_seq``
        float64 (H,), ``filter_coeffs`` float64 (3,) or None, ``chol`` float64 (A,A), ``step_counter`` int64 (1,));
        ``action_slots`` is a pinned host tensor (2, A+1).  ``record`` (float64 (2 + H*A,)): sharded runs - this GPU's
        softmax record instead of the update.  Returns (costs, actions, q0) device tensors when ``want_trajectories``."""
        launch, out = self.mppi_step_launcher(num_particles, horizon, mean, mean_out, gamma_seq, filter_coeffs, chol, seed,
                                              offset, particle_offset, step_counter, lam, step_size, shift_mode, action_out,
                                              action_slots, record, env_step, want_trajectories)
        launch()
        return out

    def mppi_step_launcher(self, num_particles, horizon, mean, mean_out, gamma_seq, filter_coeffs, chol, seed, offset,
                           particle_offset, step_counter, lam, step_size, shift_mode, action_out=None, action_slots=None,
                           record=None, env_step=False, want_trajectories=False, bind_stream=True):
        """``mppi_step`` with its arguments bound once: returns (launch, outputs) where ``launch()`` enqueues the
        iteration on the stream that is current NOW (one C call, nothing converted per step) - what a control loop
        calls every step.  ``bind_stream=False``: on the stream that is current when ``launch()`` is called (a launcher
        that is also called under stream capture must say so, or its kernels stay out of the graph).  The tensors must
        stay alive (and in place) while the launcher is in use."""
        torch = _torch()
        P, H, A = int(num_particles), int(horizon), self.d_action
        costs = act = q0 = None
        if want_trajectories:
            costs, act = self._buffer("costs", (P, H)), self._buffer("act", (P, H, A))
            q0 = self._buf.get("q0")
            if q0 is None or q0.shape[0] != P:
                q0 = self._buf["q0"] = torch.empty(P, dtype=torch.float64, device=self.device)
        scost = self._buffer("step_cost", (1,)) if env_step else None
        snobs = self._buffer("step_obs", (self.d_obs,)) if env_step else None
        keep = (mean, mean_out, gamma_seq, filter_coeffs, chol, step_counter, action_out, action_slots, record, scost, snobs,
                costs, act, q0)
        args = (self._h, self._code, P, H, _ptr(mean), _ptr(mean_out), _ptr(gamma_seq), _ptr(filter_coeffs), _ptr(chol),
                int(seed) & (2 ** 64 - 1), int(offset), int(particle_offset), _ptr(step_counter), float(lam), float(step_size),
                int(shift_mode), _ptr(action_out), _ptr(action_slots), _ptr(record), int(bool(env_step)), _ptr(scost),
                _ptr(snobs), _ptr(costs), _ptr(act), _ptr(q0))
        fn, check, stream = self._lib.mjmpc_arm_mppi_step, _lib.check, self._stream
        if bind_stream:
            args = args + (stream(),)

            def launch(_keep=keep):
                check(fn(*args))
        else:
            def launch(_keep=keep):
                check(fn(*args, stream()))

        return launch, ((costs, act, q0) if want_trajectories else None)

    def mppi_combine_launcher(self, records, n_records, horizon, mean, mean_out, step_counter, step_size, shift_mode,
                              action_out, action_slots, env_step):
        """Sharded runs: the launch behind the record all-gather (``mjmpc_arm_mppi_combine``) with its arguments bound;
        ``launch()`` enqueues it on the stream that is current WHEN IT IS CALLED (it is called under stream capture)."""
        scost = self._buffer("step_cost", (1,))         # (bound whether or not this launcher steps the env: the parameter
        snobs = self._buffer("step_obs", (self.d_obs,))  #  block of the call then never changes between launchers)
        keep = (records, mean, mean_out, step_counter, action_out, action_slots, scost, snobs)
        args = (self._h, self._code, _ptr(records), int(n_records), int(horizon), _ptr(mean), _ptr(mean_out),
                _ptr(step_counter), float(step_size), int(shift_mode), _ptr(action_out), _ptr(action_slots),
                int(bool(env_step)), _ptr(scost), _ptr(snobs))
        fn, check, stream = self._lib.mjmpc_arm_mppi_combine, _lib.check, self._stream

        def launch(_keep=keep):
            check(fn(*args, stream()))

        return launch

    def step_state(self, action):
        """Advance the engine state in place by one env step (the "real env" kept on the device).
        ``action``: numpy (A,) or CUDA float64 tensor.  Returns (cost, next_obs) device tensors."""
        torch = _torch()
        a = self._as_device(action, torch.float64, (self.d_action,))
        cost = self._buffer("step_cost", (1,))
        nobs = self._buffer("step_obs", (self.d_obs,))
        _lib.check(self._lib.mjmpc_arm_step_state(self._h, self._code, _ptr(a), _ptr(cost), _ptr(nobs),
                                                  self._stream()))
        return cost, nobs

    def solver_failures(self):
        c = ctypes.c_uint32()
        _lib.check(self._lib.mjmpc_arm_solver_failures(self._h, ctypes.byref(c)))
        return int(c.value)

    def diverged_substeps(self):
        """Resets: particle-substeps in which MuJoCo's mj_checkPos / mj_checkVel / mj_checkAcc would have called mj_resetData
        (a NaN or an entry beyond 1e10 in qpos / qvel / qacc); the kernel does the same (see TreeRolloutEngine)."""
        c = ctypes.c_uint32()
        _lib.check(self._lib.mjmpc_arm_diverged(self._h, ctypes.byref(c)))
        return int(c.value)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _buffer(self, name, shape):
        torch = _torch()
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=self._tdtype, device=self.device)
            self._buf[name] = t
        return t

    def _as_device(self, x, tdtype, shape):
        torch = _torch()
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if tuple(x.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (shape, tuple(x.shape)))
        return x.to(device=self.device, dtype=tdtype).contiguous()


def make_device_rollout_fn(sim_env):
    """Device-resident ``rollout_fn``: costs and actions come back as CUDA tensors (no observations,
    which the MPPI / CEM / DMD / random-shooting updates never read - SURVEY 8b), so one control
    iteration moves nothing across PCIe but the final action."""
    def rollout_fn(num_particles, horizon, mean, noise, mode):
        t0 = time.time()
        costs, act, _, _ = sim_env.rollout_device(num_particles, horizon, mean, noise, mode, want_obs=False)
        return dict(costs=costs, actions=act, observations=None, next_observations=None, dones=None,
                    infos={"total_time": np.array([time.time() - t0] * sim_env.num_shards)})
    rollout_fn.accepts_device = True          # controllers may hand over their device-resident mean
    rollout_fn.engine = sim_env
    if hasattr(sim_env, "rollout_fused"):   # filter + cost-to-go fused into the launch (graph fast path)
        rollout_fn.fused = sim_env.rollout_fused
    if hasattr(sim_env, "mppi_step"):       # the whole iteration in one launch (captured iterations of MPPI / DMD-MPC)
        rollout_fn.mono = sim_env.mppi_step
        rollout_fn.sampled = sim_env.rollout_sampled     # rollouts that draw their own samples (any update that reads q0 / actions)
        rollout_fn.mono_launcher = sim_env.mppi_step_launcher
        rollout_fn.combine_launcher = sim_env.mppi_combine_launcher
    return rollout_fn


def _same_state(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in ("qp", "qv", "target_pos"))


def make_rollout_fn(sim_env):
    """The ``rollout_fn`` closure of examples/example_mpc.py:112-133 over any engine with a
    reference-shaped ``rollout``: negates rewards into costs and builds the trajectory dict."""
    def rollout_fn(num_particles, horizon, mean, noise, mode):
        obs, rew, act, done, info, nobs = sim_env.rollout(num_particles, horizon, np.array(mean, copy=True),
                                                          noise, mode)
        infos = {k: np.array([d[k] for d in info]) for k in info[0]}
        return dict(observations=obs, actions=act, costs=-1.0 * rew, dones=done,
                    next_observations=nobs, infos=infos)
    return rollout_fn

"""GPU rollout engine for kinematic-tree models - ``ArmRolloutEngine``'s sibling for models the serial-chain kernel
cannot hold (a hand on an arm; the reference's vendored swimmer and half-cheetah with their floating
roots, springs, fluid forces and frictional contacts; up to 32 hinge / slide dofs).

The state dictionary follows the model's task: the reacher's ``{qp, qv, target_pos}`` (reacher_env.py:81-99) or the
locomotion envs' ``{qpos, qvel}`` (swimmer.py:33-50, half_cheetah.py:36-51).

Same reference-shaped surface (``SubprocVecEnv.rollout / set_env_state / reset / close``,
mjmpc/envs/vec_env/subproc_vec_env.py:128-186, 235-251), so ``make_rollout_fn`` / ``make_device_rollout_fn`` of
``arm_engine`` and every controller work on it unchanged:

    sim_env = TreeRolloutEngine(hand24_raw())
    controller.set_sim_state_fn = sim_env.set_env_state
    controller.rollout_fn = make_device_rollout_fn(sim_env)

One HIP launch per rollout (mjmpc_amd/csrc/tree_rollout.hip) through the C ABI (``mjmpc_tree_*``).
"""
import ctypes
import time

import numpy as np

from .. import _lib
from ..models.compile_tree import TreeModel, compile_tree
from ..models.raw import TASK_FORWARD, RawModel
from ._resets import EnvResetWatch, SimulationUnstableError  # noqa: F401
from .arm_engine import _DT, _ptr, _same_state, _torch


class TreeRolloutEngine(EnvResetWatch):
    _abi = "tree"

    def __init__(self, model, device=0, dtype="f64", num_shards=1):
        self.raw = model if isinstance(model, RawModel) else None
        if isinstance(model, RawModel):
            model = compile_tree(model)
        if not isinstance(model, TreeModel):
            raise TypeError("model must be a RawModel or a compiled TreeModel")
        if dtype not in _DT:
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.model, self.dtype = model, dtype
        self._code, self._np = _DT[dtype]
        self.num_shards = int(num_shards)
        self._lib = _lib.require_gpu()
        torch = _torch()
        self.device = torch.device("cuda", device)
        self._tdtype = torch.float32 if dtype == "f32" else torch.float64
        h = ctypes.c_void_p()
        blob = np.ascontiguousarray(model.blob, np.float64)
        _lib.check(self._lib.mjmpc_tree_create(blob.ctypes.data_as(_lib._dp), blob.size, device, ctypes.byref(h)))
        self._h = h
        self.d_action, self.d_obs = model.nu, model.d_obs
        self.forward_task = model.task == TASK_FORWARD
        self.d_state = model.nq + model.nv if self.forward_task else model.nq + 2 * model.nv + 3 + 1
        self.action_lows, self.action_highs = model.ctrl_lo.copy(), model.ctrl_hi.copy()
        self.closed = False
        self._buf = {}
        self.default_dyn_params = [dict() for _ in range(self.num_shards)]
        self.randomized_dyn_params = [dict() for _ in range(self.num_shards)]
        self.reset()

    # ------------------------------------------------------------------ reference-shaped API
    def _unpack(self, state):
        kq, kv = ("qpos", "qvel") if "qpos" in state else ("qp", "qv")
        qp = np.ascontiguousarray(state[kq], np.float64).reshape(-1)
        qv = np.ascontiguousarray(state[kv], np.float64).reshape(-1)
        tg = np.ascontiguousarray(state.get("target_pos", self.model.target_default), np.float64).reshape(-1)
        if qp.size != self.model.nq or qv.size != self.model.nv or tg.size != 3:     # (qpos in MuJoCo's layout: nq entries)
            raise ValueError("state has the wrong dimensions for this model")
        return dict(qp=qp.copy(), qv=qv.copy(), target_pos=tg.copy())

    def set_env_state(self, state_dicts):
        """``SubprocVecEnv.set_env_state`` (subproc_vec_env.py:235-251): one dict (every shard starts from it), a list
        holding one dict, or one dict per shard (shard k's particles start from states[k])."""
        if isinstance(state_dicts, (list, tuple)):
            if len(state_dicts) not in (1, self.num_shards):
                raise AssertionError("num states should equal 1 (same for all envs) or 1 per env")
            states = [self._unpack(s) for s in state_dicts]
            if any(not _same_state(states[0], s) for s in states[1:]):
                return self._set_shard_states(states)
            state = states[0]
        else:
            state = self._unpack(state_dicts)
        if getattr(self, "_per_shard_states", False):
            _lib.check(self._lib.mjmpc_tree_set_shard_states(self._h, None, 0, self._stream()))
            self._per_shard_states = False
        self._state = state
        _lib.check(self._lib.mjmpc_tree_set_state(self._h, state["qp"].ctypes.data_as(_lib._dp),
                                                  state["qv"].ctypes.data_as(_lib._dp),
                                                  state["target_pos"].ctypes.data_as(_lib._dp), self._stream()))

    def _set_shard_states(self, states):
        nv, nq = self.model.nv, self.model.nq
        arr = np.zeros((self.num_shards, 78))                   # MJMPC_TREE_STATE_LEN: qpos[40] | qvel[32] | target[3] | -
        for k, s in enumerate(states):
            arr[k, :nq], arr[k, 40:40 + nv], arr[k, 72:75] = s["qp"], s["qv"], s["target_pos"]
        _lib.check(self._lib.mjmpc_tree_set_shard_states(self._h, arr.ctypes.data_as(_lib._dp), self.num_shards,
                                                         self._stream()))
        self._per_shard_states = True
        self._shard_state_list = states
        self._state = states[0]

    def get_env_state(self):
        """One state dict - or, after a per-shard ``set_env_state``, one per shard (subproc_vec_env.py:253-256)."""
        states = self._shard_state_list if getattr(self, "_per_shard_states", False) else [self._state]
        if self.forward_task:
            return [dict(qpos=st["qp"].copy(), qvel=st["qv"].copy()) for st in states]
        return [dict(qp=st["qp"].copy(), qv=st["qv"].copy(), qa=np.zeros(self.model.nv),
                     target_pos=st["target_pos"].copy(), timestep=0) for st in states]

    def reset(self):
        self.set_env_state(dict(qp=self.model.qpos0.copy(), qv=np.zeros(self.model.nv),
                                target_pos=self.model.target_default.copy()))
        return self.get_env_state()

    def close(self):
        if not self.closed:
            self._lib.mjmpc_tree_destroy(self._h)
            self._h = None              # later calls fail with "null engine" instead of touching freed memory
            self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rollout(self, num_particles, horizon, mean, noise, mode="open_loop"):
        """``SubprocVecEnv.rollout``: numpy in, numpy out -> (obs, rew, act, done, info, next_obs)."""
        t0 = time.time()
        out = self.rollout_device(num_particles, horizon, mean, noise, mode, want_obs=True)
        costs, act, obs, nobs = (x.to("cpu").numpy().astype(np.float64, copy=False) for x in out)
        done = np.zeros((num_particles, horizon))
        info = [{"total_time": time.time() - t0} for _ in range(self.num_shards)]
        return obs, -costs, act, done, info, nobs

    def rollout_device(self, num_particles, horizon, mean, noise, mode="open_loop", want_obs=False, want_actions=True):
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
        fn = self._lib.mjmpc_tree_rollout_cl if closed else self._lib.mjmpc_tree_rollout
        _lib.check(fn(self._h, self._code, P, H, _ptr(mean_d), _ptr(noise_d), _ptr(costs), _ptr(act), _ptr(obs), _ptr(nobs),
                      self._stream()))
        return costs, act, obs, nobs

    def rollout_fused(self, num_particles, horizon, mean, raw_noise, filter_coeffs, gamma_seq, q0_out=None):
        """Device-resident rollout with the noise filter and the discounted cost-to-go fused into the launch
        (``mjmpc_tree_rollout_fused``, as ``ArmRolloutEngine.rollout_fused``).  All arguments are CUDA tensors
        (``filter_coeffs`` may be None; ``q0_out``: a float64 [P] tensor the cost-to-go is written to instead of the engine's
        own buffer).  Returns (costs, actions, q0)."""
        if num_particles % self.num_shards != 0:
            raise AssertionError("Number of particles must be divisible by number of shards")
        torch = _torch()
        P, H, A = int(num_particles), int(horizon), self.d_action
        mean_d = self._as_device(mean, torch.float64, (H, A))
        noise_d = self._as_device(raw_noise, self._tdtype, (P, H, A))
        costs = self._buffer("costs", (P, H))
        act = self._buffer("act", (P, H, A))
        q0 = q0_out if q0_out is not None else self._buf.get("q0")
        if q0 is None or q0.shape[0] != P:
            q0 = self._buf["q0"] = torch.empty(P, dtype=torch.float64, device=self.device)
        _lib.check(self._lib.mjmpc_tree_rollout_fused(self._h, self._code, P, H, _ptr(mean_d), _ptr(noise_d), _ptr(filter_coeffs),
                                                      _ptr(gamma_seq), _ptr(costs), _ptr(act), _ptr(q0), self._stream()))
        return costs, act, q0

    def step(self, action):
        """Advance the engine's own state by one env step (a one-particle rollout, state round trip through the host:
        the tree engine keeps no device-resident "real env").  Returns (next_obs, reward)."""
        if self.forward_task and self.model.obs_skip:
            raise ValueError("the observation leaves out qpos[:%d]; step an engine compiled with obs_skip = 0"
                             % self.model.obs_skip)
        with self.real_step_guard("TreeRolloutEngine.step"):
            _, rew, _, _, _, nobs = self.rollout(1, 1, np.asarray(action, np.float64).reshape(1, -1), None)
        nv, nq = self.model.nv, self.model.nq
        self.set_env_state(dict(qp=nobs[0, 0, :nq], qv=nobs[0, 0, nq:nq + nv], target_pos=self._state["target_pos"]))
        return nobs[0, 0].copy(), float(rew[0, 0])

    def step_state(self, action):
        """Advance the engine state in place by one env step (the "real env" kept on the device, as
        ``ArmRolloutEngine.step_state``).  ``action``: numpy (A,) or CUDA float64 tensor.  Returns (cost, next_obs)
        device tensors; ``get_state_device()`` reads the state back."""
        torch = _torch()
        a = self._as_device(action, torch.float64, (self.d_action,))
        cost = self._buffer("step_cost", (1,))
        nobs = self._buffer("step_obs", (self.d_obs,))
        _lib.check(self._lib.mjmpc_tree_step_state(self._h, self._code, _ptr(a), _ptr(cost), _ptr(nobs), self._stream()))
        return cost, nobs

    def get_state_device(self):
        """The device-resident state as the task's state dictionary (one D2H copy; synchronises the stream)."""
        qp, qv = np.zeros(self.model.nq), np.zeros(self.model.nv)
        _lib.check(self._lib.mjmpc_tree_get_state(self._h, qp.ctypes.data_as(_lib._dp), qv.ctypes.data_as(_lib._dp), self._stream()))
        if self.on_env_reset != "ignore":
            self.check_env_resets("the device-resident env (step_state)")     # (the host has synchronised anyway)
        if self.forward_task:
            return dict(qpos=qp, qvel=qv)
        return dict(qp=qp, qv=qv, qa=np.zeros(self.model.nv), target_pos=self._state["target_pos"].copy(), timestep=0)

    def randomize_dynamics(self, param_dict, base_seed):
        """``SubprocVecEnv.randomize_dynamics`` (subproc_vec_env.py:304-312), as ``ArmRolloutEngine.randomize_dynamics``:
        shard i draws from ``np_random(base_seed + i*12345)`` a uniform value in ``m (1 +- noise)``, ``m = (1 + bias) *
        default`` for every ``{param_id: {name: [noise_scale, bias_scale]}}`` entry (gym_env_wrapper.py:367-416) and from
        then on simulates its own model block (``mjmpc_tree_set_shard_models``).  Supported: body_mass, body_inertia,
        dof_damping, dof_frictionloss (friction-loss constraint rows), geom_size and geom_friction of colliding geoms,
        sensor_noise (a known sensor's draw is consumed; no observation reads a sensor).
        Returns (default_params, randomized_params), one dict per shard."""
        if self.raw is None:
            raise ValueError("randomize_dynamics needs the engine to be built from a RawModel")
        from .seeding import np_random
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
                    rand.setdefault(param_id, {})[name] = rng.uniform(mean - mean * noise_scale, mean + mean * noise_scale)
            # (dof_frictionloss: friction-loss constraint rows - the general kernel instantiation, which the engine's own model
            # must already run: a model whose defaults are all zero keeps them zero under a multiplicative draw)
            blobs.append(compile_tree(self.raw, overrides=rand, base=self.model).blob)
        blobs = np.ascontiguousarray(np.stack(blobs), np.float64)
        _lib.check(self._lib.mjmpc_tree_set_shard_models(self._h, blobs.ctypes.data_as(_lib._dp), self.num_shards))
        self.shard_blobs = blobs
        return self.default_dyn_params, self.randomized_dyn_params

    def _default_param(self, param_id, name):
        from ..models.compile import principal_inertia
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
                return np.array([g.friction, 0.005, 0.0001])          # (torsional / rolling: MuJoCo's defaults, unused here)
            half = 0.5 * np.linalg.norm(np.asarray(g.b, float) - np.asarray(g.a, float)) if g.type == 2 else 0.0
            return np.array([g.radius, half, 0.0])
        if param_id == "dof_frictionloss":
            return float(next(b.joint.frictionloss for b in raw.bodies if b.joint is not None and b.joint.name == name))
        if param_id == "sensor_noise":
            # (gym_env_wrapper.py:396-398 - model.sensor_noise: MuJoCo keeps the value for the user and adds no noise itself, and no
            # observation on the path reads a sensor: the draw is consumed, as in the reference, and changes nothing)
            if name not in raw.sensors:
                raise ValueError("no sensor named %r" % name)
            return float(raw.sensors[name])
        raise ValueError("Unknown dynamics field")

    def solver_failures(self):
        c = ctypes.c_uint32()
        _lib.check(self._lib.mjmpc_tree_solver_failures(self._h, ctypes.byref(c)))
        return int(c.value)

    def diverged_substeps(self):
        """Resets: particle-substeps in which MuJoCo's mj_checkPos / mj_checkVel / mj_checkAcc would have called mj_resetData
        (a NaN or an entry beyond 1e10 in qpos / qvel / qacc); the kernel does the same and the particle rolls on from
        qpos0 with finite costs - counted apart from solver_failures()."""
        c = ctypes.c_uint32()
        _lib.check(self._lib.mjmpc_tree_diverged(self._h, ctypes.byref(c)))
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

"""Controller base classes with the reference's public surface.

``Controller`` keeps mjmpc/control/controller.py's contract: settable ``rollout_fn`` /
``set_sim_state_fn`` plug points (:152-175) and ``optimize(state, calc_val, hotstart) -> (action,
value)`` (:207-257).  ``OLGaussianMPC`` keeps mjmpc/control/olgaussian_mpc.py's: a mean ``(H,A)`` and
a covariance ``(A,A)``, ``sample_noise`` (:88-93), ``generate_rollouts`` (:95-114), ``_shift``
(:116-129), ``reset`` (:131-135).

What is new: the distribution lives on the GPU (``DeviceUpdater``) and ``_update_distribution`` runs
HIP reductions; ``mean_action`` / ``cov_action`` are host mirrors refreshed after every update, so
code that reads them keeps working.  ``rollout_fn`` may return numpy arrays (any user callback, as
in the reference) or CUDA tensors (``make_device_rollout_fn``: nothing leaves the GPU).

Extra constructor keywords (all optional, defaults reproduce the reference bit for bit):
  noise_mode  'host'   legacy numpy stream, identical seeds -> identical noise (parity mode)
              'device' Philox sampler on the GPU (same distribution, not the same bits)
              'device_mt19937'  the reference's own stream (MT19937 + polar method) regenerated on the GPU:
                       identical seeds -> the same particles (to the last bit or two), isotropic covariance;
                       sharded runs keep their block of the one stream; the serial twister recurrence is cut into 32 jumped-ahead segments
                       (mt_jump.py), ~0.1 ms per 4096x32x7 draw
  noise_dtype 'f64' | 'f32'  storage type of device-sampled noise
  device      CUDA device ordinal;  comm  particle-sharding communicator (see _device.py)
"""
import copy
import ctypes
import inspect
from abc import ABC, abstractmethod

import numpy as np

from .. import _lib
from ._device import DeviceUpdater
from .sharding import local_block
from .control_utils import generate_noise


def _seed_value(seed):
    """gym.utils.seeding.np_random(seed) semantics: the given seed is the seed value."""
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0])
    if not isinstance(seed, (int, np.integer)) or seed < 0:
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    return int(seed)


class Controller(ABC):
    def __init__(self, d_state, d_obs, d_action, action_lows, action_highs, horizon, gamma, n_iters,
                 set_sim_state_fn=None, rollout_fn=None, sample_mode='mean', batch_size=1, seed=0,
                 device=0, comm=None):
        self.d_state = d_state
        self.d_obs = d_obs
        self.d_action = d_action
        self.action_lows = action_lows
        self.action_highs = action_highs
        self.horizon = horizon
        self.gamma = gamma
        self.gamma_seq = np.cumprod([1.0] + [self.gamma] * (horizon - 1)).reshape(1, horizon)
        self.n_iters = n_iters
        self._set_sim_state_fn = set_sim_state_fn
        self._rollout_fn = rollout_fn
        self.sample_mode = sample_mode
        self.batch_size = batch_size
        self.num_steps = 0
        self.seed_val = self.seed(seed)
        self.dev = DeviceUpdater(horizon, d_action, self.gamma_seq, device=device, comm=comm)

    # -- plug points (controller.py:152-175) ------------------------------------------------
    @property
    def set_sim_state_fn(self):
        return self._set_sim_state_fn

    @set_sim_state_fn.setter
    def set_sim_state_fn(self, fn):
        self._set_sim_state_fn = fn

    @property
    def rollout_fn(self):
        return self._rollout_fn

    @rollout_fn.setter
    def rollout_fn(self, fn):
        self._rollout_fn = fn

    # -- hooks ------------------------------------------------------------------------------
    @abstractmethod
    def _get_next_action(self, state, mode='mean'):
        pass

    @abstractmethod
    def _update_distribution(self, trajectories):
        pass

    @abstractmethod
    def _shift(self):
        pass

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def _calc_val(self, trajectories):
        pass

    @abstractmethod
    def generate_rollouts(self, state):
        pass

    def sample_actions(self):
        raise NotImplementedError('sample_actions funtion not implemented')

    def check_convergence(self):
        return False

    # -- the MPC iteration (controller.py:207-257) ---------------------------------------------
    def optimize(self, state, calc_val=False, hotstart=True):
        for _ in range(self.n_iters):
            trajectory = self.generate_rollouts(copy.deepcopy(state))
            self._update_distribution(trajectory)
            if self.check_convergence():
                break
        curr_action = self._get_next_action(state, mode=self.sample_mode)
        value = 0.0
        if calc_val:
            trajectories = self.generate_rollouts(copy.deepcopy(state))
            value = self._calc_val(trajectories)
        self.num_steps += 1
        if hotstart:
            self._shift()
        return curr_action, value

    def get_optimal_value(self, state):
        self.reset()
        _, value = self.optimize(state, calc_val=True, hotstart=False)
        return value

    def seed(self, seed=None):
        seed = _seed_value(seed)
        self.np_random = np.random.RandomState(seed)
        return seed


_SHIFT_MODES = {'null': 0, 'repeat': 1, 'random': 2}


def resident_state(state):
    """``set_sim_state_fn`` of a closed loop whose real env lives on the device (``engine.step_state`` advances it inside
    the captured iteration): there is nothing to copy or upload, and ``optimize()`` skips the deep copy of the state
    dictionary the reference makes for the callback (controller.py:217: 2 us of the host's turn-around per control step)."""
    return None


resident_state.ignores_state = True


def _keep_graph_supported(torch):
    """``torch.cuda.CUDAGraph(keep_graph=True)`` + ``raw_cuda_graph()`` arrived together (what the launch tape's acceptance
    test reads); a torch without them replays the plain hipGraph instead of losing the captured path altogether."""
    return hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph")


class _AlternatingGraphs:
    """Two captured iterations, one per direction of the mean's double buffer: ``replay()`` runs the one that reads the
    buffer that is the mean now, then makes the buffer it wrote the mean."""

    def __init__(self, dev, graphs):
        self.dev, self.graphs = dev, graphs

    def replay(self):
        self.graphs[id(self.dev.mean)].replay()
        self.dev.mean, self.dev.mean_alt = self.dev.mean_alt, self.dev.mean


class _LaunchTape:
    """The captured control iteration as the list of library calls that make it up, replayed call by call on the current
    stream.  On this runtime consecutive hipGraph replays are 9-13 us apart on the device whatever the host does
    (tools/kernel_gaps.py), plain launches follow each other at once - so an iteration whose launches are ALL calls into
    the library (checked by the owner: a capture of the tape's replay has as many kernel nodes as the captured iteration
    itself, and they are the same kernels in the same launch shapes) runs from its tape.  Arguments are the ones recorded under capture; the stream argument (last by this
    library's convention) is replaced by the stream that is current at replay."""

    def __init__(self, calls, recorded_stream, dev):
        self.dev = dev
        self.calls = []
        for fn, args in calls:          # (calls that take no stream are host-side queries - sizes, addresses - not launches)
            tail = args[-1] if args else None
            tail = getattr(tail, "value", tail)
            if tail is not None and tail == recorded_stream:
                self.calls.append((fn, args[:-1]))

    def replay(self):
        s = self.dev.torch.cuda.current_stream(self.dev.device).cuda_stream
        for fn, args in self.calls:
            rc = fn(*args, s)
            if rc:
                _lib.check(rc)


class OLGaussianMPC(Controller):
    """Open-loop Gaussian MPC: N(mean_action[t], cov_action) per horizon step."""

    def __init__(self, d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov, init_mean,
                 base_action, num_particles, gamma, n_iters, step_size, filter_coeffs, set_sim_state_fn=None,
                 rollout_fn=None, cov_type='diagonal', sample_mode='mean', batch_size=1, seed=0,
                 use_zero_control_seq=False, noise_mode='host', noise_dtype='f64', device=0, comm=None):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, gamma, n_iters,
                         set_sim_state_fn, rollout_fn, sample_mode, batch_size, seed, device=device, comm=comm)
        if noise_mode not in ('host', 'device', 'device_mt19937'):
            raise ValueError("noise_mode must be 'host', 'device' or 'device_mt19937'")
        self._mean_stale = self._cov_stale = False
        self._mean_seen = self._cov_seen = None
        self.init_cov = np.array([init_cov] * self.d_action)
        self.init_mean = init_mean.copy()
        self.mean_action = init_mean
        self.base_action = base_action
        self.num_particles = num_particles
        self.cov_type = cov_type
        self.cov_action = np.diag(self.init_cov)
        self.step_size = step_size
        self.filter_coeffs = filter_coeffs
        self.use_zero_control_seq = use_zero_control_seq
        self.noise_mode = noise_mode
        self.noise_dtype = noise_dtype
        self._graph = None
        self._graph_post = None
        self._graph_on = False
        self._want_mono = True
        self._mono = False
        self._lookahead = False
        self._ahead = 0
        self._push()

    # -- host <-> device mirrors ---------------------------------------------------------------
    # ``mean_action`` / ``cov_action`` are lazy host mirrors of the device tensors: a device-side
    # update only marks them stale, the D2H copy happens when somebody reads them.
    @property
    def mean_action(self):
        if self._mean_stale:
            self._mean_host = self.dev.get_mean()
            self._mean_seen = self._mean_host.copy()
            self._mean_stale = False
        return self._mean_host

    @mean_action.setter
    def mean_action(self, value):
        self._mean_host = value
        self._mean_seen = None            # unknown to the device: _sync_in will upload it
        self._mean_stale = False

    @property
    def cov_action(self):
        if self._cov_stale:
            self._cov_host = self.dev.get_cov()
            self._cov_seen = self._cov_host.copy()
            self._cov_stale = False
        return self._cov_host

    @cov_action.setter
    def cov_action(self, value):
        self._cov_host = value
        self._cov_seen = None
        self._cov_stale = False

    def _push(self):
        self.dev.set_mean(self.mean_action)
        self.dev.set_cov(self.cov_action)
        self._mean_seen = np.array(self._mean_host, copy=True)
        self._cov_seen = np.array(self._cov_host, copy=True)

    def _pull(self, cov=False):
        """Called after a device-side update: the host mirrors are now out of date."""
        self._mean_stale = True
        if cov:
            self._cov_stale = True

    def _sync_in(self):
        """The host arrays are the public truth whenever they are fresh (users assign or edit
        ``mean_action`` directly, as the reference allows): upload them if they differ from what the
        device last saw."""
        if not self._mean_stale and (self._mean_seen is None or not np.array_equal(self._mean_host, self._mean_seen)):
            self.dev.set_mean(self._mean_host)
            self._mean_seen = np.array(self._mean_host, copy=True)
        if not self._cov_stale and (self._cov_seen is None or not np.array_equal(self._cov_host, self._cov_seen)):
            self.dev.set_cov(self._cov_host)
            self._cov_seen = np.array(self._cov_host, copy=True)
            return True         # a covariance went up: samples drawn ahead / a factor left on the device are stale
        return False

    @property
    def local_particles(self):
        return local_block(self.num_particles, self.dev.comm.rank, self.dev.comm.world_size)[1]

    # -- sampling (olgaussian_mpc.py:69-93) ------------------------------------------------------
    def _get_next_action(self, state, mode='mean'):
        if mode == 'mean':
            return self.mean_action[0].copy()
        if mode == 'sample':
            delta = generate_noise(self.cov_action, self.filter_coeffs, shape=(1, 1),
                                   base_seed=self.seed_val + 123 * self.num_steps)
            return self.mean_action[0].copy() + delta.reshape(self.d_action).copy()
        raise ValueError('Unidentified sampling mode in get_next_action')

    def sample_noise(self):
        """(P_local, H, A) perturbations.  'host': the reference's stream, sliced to this rank's
        contiguous particle block; 'device': Philox, keyed by the GLOBAL particle index."""
        n_loc, rank = self.local_particles, self.dev.comm.rank
        if self.noise_mode == 'host':
            delta = generate_noise(self.cov_action, self.filter_coeffs, shape=(self.num_particles, self.horizon),
                                   base_seed=self.seed_val + self.num_steps)
            return delta[rank * n_loc:(rank + 1) * n_loc] if self.dev.comm.world_size > 1 else delta
        if self.noise_mode == 'device_mt19937':
            return self.dev.sample_noise_mt19937(n_loc, self.cov_action, self.filter_coeffs, self.seed_val,
                                                 self.num_steps, dtype=self.noise_dtype, particle_offset=rank * n_loc)
        if self._device_cov():
            self._sync_in()             # a covariance the user assigned on the host is uploaded first
            return self.dev.sample_noise(n_loc, None, self.filter_coeffs, self.seed_val, self.num_steps,
                                         dtype=self.noise_dtype, particle_offset=rank * n_loc,
                                         device_cov_diagonal=self.cov_type == 'diagonal')
        return self.dev.sample_noise(n_loc, self.cov_action, self.filter_coeffs, self.seed_val, self.num_steps,
                                     dtype=self.noise_dtype, particle_offset=rank * n_loc)

    def generate_rollouts(self, state):
        self._sync_in()
        self._set_sim_state_fn(copy.deepcopy(state))
        delta = self.sample_noise()
        if self.use_zero_control_seq and self.dev.comm.rank == self.dev.comm.world_size - 1:
            if isinstance(delta, np.ndarray):
                delta[-1, :] = -1.0 * np.array(self.mean_action, copy=True)
            else:
                delta[-1] = (-self.dev.mean).to(delta.dtype)
        mean = self.dev.mean if getattr(self._rollout_fn, "accepts_device", False) else self.mean_action
        return self._rollout_fn(self.local_particles, self.horizon, mean, delta, mode="open_loop")

    # -- hipGraph fast path ---------------------------------------------------------------------
    def enable_graph(self, post_step=None, lookahead=False, mono=True, tape=True):
        """Capture one whole control iteration (noise -> rollout -> update -> action -> shift) as a
        hipGraph and replay it from ``optimize()``: one launch and one stream sync per step instead of
        ~12 launches.  Needs a device-resident pipeline: ``noise_mode='device'``, a rollout_fn made by
        ``make_device_rollout_fn``, a static covariance, and a shift that needs no host RNG.
        ``post_step(action_tensor)`` is captured too (e.g. stepping the real env on the device).

        ``mono``: where the engine offers it (``rollout_fn.mono``: MPPI / DMD-MPC with a static diagonal covariance on
        the arm engine) the iteration is ONE kernel - sampling, rollout, update, action, shift and, if ``post_step`` is
        that engine's own ``step_state``, the env step (``mjmpc_arm_mppi_step``).

        ``tape`` (one GPU): an iteration of several launches runs from the recorded list of its library calls instead
        of a hipGraph replay when that list is the whole iteration (``_LaunchTape``: consecutive graph replays are
        ~10 us apart on the device, plain launches are not); ``launch_mode`` says which one is in use.

        ``lookahead``: the caller promises that the states it passes to ``optimize()`` carry no information - the real
        env lives on the device and ``post_step`` advances it - so iteration k + 1 depends on nothing the host provides.
        ``optimize()`` then enqueues iteration k + 1 BEFORE it waits for the action of iteration k: the GPU goes from
        one iteration to the next without waiting for the host's round trip.  The device then runs one iteration
        ahead of the actions the host has seen; host mirrors (``mean_action``) read the device's latest.  Whatever the
        host assigns between two ``optimize()`` calls (``mean_action``, ``cov_action``, ``num_steps``; an engine
        ``set_env_state``) cannot reach the iteration that is already in flight: ``optimize()`` raises when it finds
        such an assignment; call ``reset()`` (which drains the queue) before editing the distribution or the engine
        state from the host."""
        if not self._graph_capable():
            raise ValueError("this controller configuration cannot run as a captured graph "
                             "(needs noise_mode='device', a device rollout_fn, static covariance, "
                             "base_action != 'random'; sharded runs additionally the fused MPPI path over RCCL)")
        self._graph_on = True
        self._graph_post = post_step
        self._graph = None
        self._noise_valid = False
        self._want_mono = bool(mono)
        self._want_tape = bool(tape)
        self.launch_mode = "hipGraph replay"
        self._mono = False
        self._lookahead = bool(lookahead)
        self._ahead = 0                 # iterations enqueued beyond the ones whose action the host has taken

    def _host_uploads_per_step(self):
        return False            # subclasses whose _device_update uploads host tables on every call say True

    def _graph_capable(self):
        return (not self._host_uploads_per_step() and self.noise_mode in ('device', 'device_mt19937') and getattr(self._rollout_fn, "accepts_device", False)
                and self.base_action in ('null', 'repeat') and self.sample_mode == 'mean'
                and (self._static_cov() or (self._device_cov() and self.noise_mode == 'device'))
                and (self.dev.comm.world_size == 1 or getattr(self.dev.comm, "backend", "") == "nccl"))
        # (sharded runs: RCCL collectives are captured with the iteration - MPPI / DMD-MPC one record all-gather, CEM its
        # two exchanges, random shooting one; if the runtime refuses, every rank drops to eager launches together)

    def _static_cov(self):
        return False            # subclasses whose update leaves cov_action alone say True

    def _device_cov(self):
        return False            # subclasses that adapt the covariance entirely on the device say True

    def _q0_kw(self, n_loc):
        """``q0_out=`` for fused rollouts that take it: the launch writes the cost-to-go where the update reads it."""
        fused = self._rollout_fn.fused
        cached = getattr(self, "_q0_kw_for", None)
        if cached is None or cached[0] is not fused:            # (inspect.signature costs tens of microseconds)
            cached = self._q0_kw_for = (fused, "q0_out" in inspect.signature(fused).parameters)
        return dict(q0_out=self.dev.q0_destination(n_loc)) if cached[1] else {}

    def _shift_cov_args(self):
        """(diag or None = identity, scale) of the covariance growth after the shift (cem.py:94, gaussian_dmd.py:111);
        None: the covariance does not grow."""
        return None

    def _device_shift_cov(self):
        grow = self._shift_cov_args()
        if grow is not None:
            self.dev.add_cov_diag(*grow)

    def _device_update(self, trajectories):
        raise NotImplementedError

    def _fused_capable(self):
        return False            # MPPI overrides: filter + cost-to-go + update/shift fusions

    def _wants_q0(self):
        return False            # CEM / random shooting: the update reads nothing of the costs but q0

    def _draw_raw(self, n_loc, steps_ahead):
        """Unfiltered device noise of control step (device step counter + steps_ahead)."""
        if self.noise_mode == 'device_mt19937':
            return self.dev.sample_noise_mt19937(n_loc, self._cov_host, self.filter_coeffs, self.seed_val, steps_ahead,
                                                 dtype=self.noise_dtype, d_step=self._step_dev, filtered=False,
                                                 particle_offset=self.dev.comm.rank * n_loc)
        return self.dev.sample_noise(n_loc, self._noise_cov(), self.filter_coeffs, self.seed_val, steps_ahead,
                                     dtype=self.noise_dtype, d_step=self._step_dev, filtered=False,
                                     particle_offset=self.dev.comm.rank * n_loc,
                                     device_cov_diagonal=self.cov_type == 'diagonal')

    def _noise_cov(self):
        """Host covariance for the sampler, or None when the covariance adapts on the device."""
        return self._cov_host if self._static_cov() else None

    def _bind_mono(self, env_step, direct=False):
        """Bind the fused iteration's arguments once (``rollout_fn.mono_launcher``): per step it is one C call.
        ``direct``: the launchers are never called under stream capture (the stream is bound too)."""
        n_loc = self.local_particles
        chol, coeffs, _ = self.dev.prepare_noise(self._cov_host, self.filter_coeffs)
        fc = np.asarray(self.filter_coeffs, np.float64)
        coeffs = None if (fc[0] == 1.0 and fc[1] == 0.0 and fc[2] == 0.0) else coeffs
        eng = getattr(self._rollout_fn, "engine", None)
        post = self._graph_post
        own_step = post is not None and getattr(post, "__self__", None) is eng and getattr(post, "__name__", "") == "step_state"
        sharded = self.dev.comm.world_size > 1
        self._mono_rec = self.dev.record("fused_rec", 2 + self.horizon * self.d_action) if sharded else None
        # the launch that finishes the iteration also steps the real env when that env is this engine's resident state
        self._mono_steps_env = bool(env_step and own_step)
        # one bound launcher per direction of the mean's double buffer (the iteration reads one tensor and writes the
        # other; _device_iteration swaps them afterwards and picks the launcher by the tensor that is the mean then)
        self._mono_launch, self._mono_combine = {}, {}
        recs = self.dev.comm.all_gather(self._mono_rec) if sharded else None   # (the persistent receive buffer, [G][2 + H A])
        self._mono_gather = None
        if sharded:
            mk = getattr(self.dev.comm, "all_gather_launcher", None)
            self._mono_gather = mk(self._mono_rec) if (mk and direct) else (lambda: self.dev.comm.all_gather(self._mono_rec))
        for src, dst in ((self.dev.mean, self.dev.mean_alt), (self.dev.mean_alt, self.dev.mean)):
            self._mono_launch[id(src)], _ = self._rollout_fn.mono_launcher(
                n_loc, self.horizon, src, dst, self.dev.gseq, coeffs, chol, self.seed_val, 0, self.dev.comm.rank * n_loc,
                self._step_dev, self.lam, self.step_size, _SHIFT_MODES[self.base_action], action_out=self._action_dev,
                action_slots=None if sharded else self._action_pin, record=self._mono_rec,
                env_step=self._mono_steps_env and not sharded,
                bind_stream=direct or not sharded)     # (sharded, captured: the launcher runs under stream capture)
            if sharded:         # behind the all-gather: merge the G records, update + shift, action, env step - one launch
                self._mono_combine[id(src)] = self._rollout_fn.combine_launcher(
                    recs, recs.shape[0], self.horizon, src, dst, self._step_dev, self.step_size,
                    _SHIFT_MODES[self.base_action], self._action_dev, self._action_pin, self._mono_steps_env)

    def _noise_ahead(self):
        """Captured fused iterations with the Philox sampler draw the next step's samples inside the update."""
        return ((not self._mono) and self._graph_on and self.noise_mode == 'device'
                and (self._fused_capable() or (self._cem_fused() and not self._cem_in_kernel())))

    def _cem_in_kernel(self):
        """The fused CEM step on launches the engine can sample for itself (``rollout_fn.sampled``: at most one wavefront
        per SIMD pair, as the one-launch MPPI iteration): no sample buffer at all - the rollout kernel colours its own
        Philox draws with the factor the finish launch of the previous step left on the device."""
        eng = getattr(self._rollout_fn, "engine", None)
        return (self._cem_fused() and getattr(self, "_want_cem_in_kernel", True) and hasattr(self._rollout_fn, "sampled")
                and hasattr(eng, "mppi_step_supported") and eng.mppi_step_supported(self.local_particles, self.horizon)
                and getattr(eng, "num_shards", 1) == 1 and not getattr(eng, "_per_shard_states", False)
                and not hasattr(eng, "shard_blobs") and getattr(eng, "dtype", self.noise_dtype) == self.noise_dtype)

    def _cem_fused(self):
        return False            # CEM overrides: selection + moments and refit + tail + next samples, two launches

    def _mono_capable(self):
        """The whole iteration in one launch (``rollout_fn.mono``): the fused MPPI / DMD-MPC update, the Philox sampler,
        a static DIAGONAL covariance, one iteration per step."""
        if not (getattr(self, "_want_mono", True) and self._graph_on and self.noise_mode == 'device' and self.n_iters == 1
                and self._fused_capable() and self._static_cov() and hasattr(self._rollout_fn, "mono")):
            return False
        cov = np.asarray(self._cov_host, np.float64)
        eng = getattr(self._rollout_fn, "engine", None)
        if hasattr(eng, "mppi_step_supported") and not eng.mppi_step_supported(self.local_particles, self.horizon):
            return False
        return (np.count_nonzero(cov - np.diag(np.diag(cov))) == 0 and getattr(eng, "num_shards", 1) == 1
                and not getattr(eng, "_per_shard_states", False) and not hasattr(eng, "shard_blobs")
                and getattr(eng, "dtype", self.noise_dtype) == self.noise_dtype)

    def _device_iteration(self):
        """The control iteration without any host synchronisation (capturable)."""
        if self._mono:              # (first: this branch is the host leg of the headline's control step)
            key = id(self.dev.mean)
            self._mono_launch[key]()
            if self.dev.comm.world_size > 1:
                self._mono_gather()                             # (into the buffer the combine launch is bound to)
                self._mono_combine[key]()
            # the new mean was written to the other buffer: it is the mean now
            self.dev.mean, self.dev.mean_alt = self.dev.mean_alt, self.dev.mean
            if self._graph_post is not None and not self._mono_steps_env:
                self._graph_post(self._action_dev)
            return
        n_loc = self.local_particles
        if self._fused_capable():
            # noise (raw) -> rollout (filters the noise, emits the cost-to-go) -> update + action + shift
            coeffs = self.dev.record("coeffs", 3)
            # In a captured iteration the sampler rides in the update's first launch: the raw samples of step k+1 are
            # drawn (into the same buffer) once the rollout of step k no longer needs them - see _optimize_graphed
            # for the first step.  Every iteration of one optimize() uses the same base seed (olgaussian_mpc.py:91).
            ahead = self._noise_ahead()
            for it in range(self.n_iters):
                raw = self.dev._rec[("noise", self.noise_dtype)] if ahead else self._draw_raw(n_loc, 0)
                costs, actions, q0 = self._rollout_fn.fused(n_loc, self.horizon, self.dev.mean, raw, coeffs,
                                                            self.dev.gseq)
                last = it == self.n_iters - 1
                nxt = None
                if ahead and last:
                    nxt = dict(seed=self.seed_val, offset=1, particle_offset=self.dev.comm.rank * n_loc,
                               d_step=self._step_dev)
                self.dev.mppi_fused_update(q0, actions, self.lam, self.step_size,
                                           _SHIFT_MODES[self.base_action] if last else -1,
                                           self._action_dev if last else None,
                                           self._action_pin if last else None, self._step_dev if last else None,
                                           draw_next=nxt)
            if self._graph_post is not None:
                self._graph_post(self._action_dev)
            return
        if self._cem_fused():
            # CEM (cem.py:65-95) beside the rollout in TWO launches: selection + elite list + moments, then refit + covariance
            # growth + Cholesky factor + action + shift + step counter + the raw samples of the NEXT step, drawn with the new
            # factor into the buffer this step's rollout has finished reading
            in_kernel = self._cem_in_kernel()
            if in_kernel:
                fc = np.asarray(self.filter_coeffs, np.float64)
                coeffs = None if (fc[0] == 1.0 and fc[1] == 0.0 and fc[2] == 0.0) else self.dev.record("coeffs", 3)
                costs, actions, q0 = self._rollout_fn.sampled(n_loc, self.horizon, self.dev.mean, self.dev.gseq, coeffs,
                                                              self.dev.record("chol", self.d_action * self.d_action), True,
                                                              self.seed_val, 0, self.dev.comm.rank * n_loc, self._step_dev,
                                                              q0_out=self.dev.q0_destination(n_loc))
                raw = None
            else:
                raw = self.dev._rec[("noise", self.noise_dtype)]
                costs, actions, q0 = self._rollout_fn.fused(n_loc, self.horizon, self.dev.mean, raw, self.dev.record("coeffs", 3),
                                                            self.dev.gseq, **self._q0_kw(n_loc))
                if self._q0_kw(n_loc) == {}:
                    self.dev._take_q0(self.dev.workspace(n_loc), n_loc, q0)
            self.dev.cem_fused_step(actions, self.num_elite, self.step_size, self.cov_type == 'full',
                                    _SHIFT_MODES[self.base_action], self._action_dev, self._action_pin, self._step_dev,
                                    self._shift_cov_args(), raw, self.seed_val, self.dev.comm.rank * n_loc)
            if self._graph_post is not None:
                self._graph_post(self._action_dev)
            return
        # Updates that only need q0 = cost_to_go(costs)[:, 0] (CEM, random shooting) take it from the rollout launch, which
        # also applies the noise filter on the fly: one filter pass and one cost-to-go pass less per iteration
        q0_fused = (self._wants_q0() and self.noise_mode == 'device' and hasattr(self._rollout_fn, "fused")
                    and not self.dev.gamma_zero and not self.use_zero_control_seq)
        for _ in range(self.n_iters):
            if q0_fused:
                raw = self._draw_raw(n_loc, 0)
                costs, actions, q0 = self._rollout_fn.fused(n_loc, self.horizon, self.dev.mean, raw,
                                                            self.dev.record("coeffs", 3), self.dev.gseq, **self._q0_kw(n_loc))
                self._device_update(dict(costs=costs, actions=actions, q0=q0))
                continue
            if self.noise_mode == 'device_mt19937':
                delta = self.dev.sample_noise_mt19937(n_loc, self._cov_host, self.filter_coeffs, self.seed_val, 0,
                                                      dtype=self.noise_dtype, d_step=self._step_dev,
                                                      particle_offset=self.dev.comm.rank * n_loc)
            else:
                delta = self.dev.sample_noise(n_loc, self._noise_cov(), self.filter_coeffs, self.seed_val, 0,
                                              dtype=self.noise_dtype, d_step=self._step_dev,
                                              particle_offset=self.dev.comm.rank * n_loc,
                                              device_cov_diagonal=self.cov_type == 'diagonal')
            if self.use_zero_control_seq and self.dev.comm.rank == self.dev.comm.world_size - 1:
                delta[-1] = (-self.dev.mean).to(delta.dtype)    # the LAST particle of the whole set (olgaussian_mpc.py:110-111)
            traj = self._rollout_fn(n_loc, self.horizon, self.dev.mean, delta, mode="open_loop")
            self._device_update(traj)
        # action read-out (device copy + pinned host copy), shift, covariance growth, step counter: one launch
        self.dev.step_tail(_SHIFT_MODES[self.base_action], self._action_dev, self._action_pin, self._step_dev,
                           self._shift_cov_args())
        if self._graph_post is not None:
            self._graph_post(self._action_dev)

    def _host_dirty(self):
        """A host-assigned mean / covariance the device has not seen (what ``_sync_in`` would upload)."""
        return ((not self._mean_stale and (self._mean_seen is None or not np.array_equal(self._mean_host, self._mean_seen)))
                or (not self._cov_stale and (self._cov_seen is None or not np.array_equal(self._cov_host, self._cov_seen))))

    def _optimize_graphed(self, state):
        torch = self.dev.torch
        if self._ahead > 0 and self._host_dirty():
            # lookahead: iteration k + 1 is already in the queue, computed from the device's mean; an upload now would land
            # behind it and silently take effect one step late (like num_steps below)
            raise RuntimeError("mean_action / cov_action were assigned while an iteration enqueued ahead "
                               "(enable_graph(lookahead=True)) was in flight; call reset() first, or run without lookahead")
        if self._sync_in() and self._graph is not None:
            # A covariance assigned on the host between two captured steps: this step's samples were already drawn (by
            # the previous update / finish launch) or would be coloured with the factor that launch left behind - draw
            # them again from the covariance just uploaded.  A static covariance is also baked into the capture (bound
            # sampler parameters, the one-launch iteration's diagonal test): capture again.
            self._noise_valid = False
            if self._static_cov():
                self._graph = None
        if not getattr(self._set_sim_state_fn, "ignores_state", False):
            self._set_sim_state_fn(copy.deepcopy(state) if state is not None else None)
        if self._graph is None:
            self._mono = self._mono_capable()           # (decided once per capture: the test reads host arrays)
            self._step_dev = torch.full((1,), self.num_steps, dtype=torch.int64, device=self.dev.device)
            self._step_host = self.num_steps
            self._action_dev = torch.zeros(self.d_action, dtype=torch.float64, device=self.dev.device)
            # action | step flag, two slots: one-launch iterations publish into slot (step & 1), the others into slot 0
            self._action_pin = torch.zeros(2 * (self.d_action + 1), dtype=torch.float64).pin_memory()
            self._action_np = self._action_pin.numpy()
            if self._mono and (self.dev.comm.world_size == 1 or getattr(self.dev.comm, "lib_collectives", False)):
                # two launches per iteration - sharded: rollout + record, the library's all-gather, the combine launch -
                # and nothing to capture: they are enqueued directly (a hipGraph replay costs 9-13 us between replays on
                # the device side, plain launches next to nothing)
                self._bind_mono(env_step=True, direct=True)
                self._graph = "direct"
                self.launch_mode = "launched directly"
            else:
                self._capture_iteration()
        if self._graph is None:                 # capture failed: every rank has dropped to eager launches
            return self._optimize_eager_after_fallback(state)
        if self._step_host != self.num_steps:
            if self._ahead:
                raise RuntimeError("num_steps was changed while an iteration enqueued ahead was in flight")
            self._step_dev.fill_(self.num_steps)
            self._noise_valid = False
        if not self._noise_valid:               # (first step / after a jump of the step counter; the tests below cost ~20 us)
            if self._noise_ahead():
                self._draw_raw(self.local_particles, 0)         # the current step's samples
            elif self._cem_in_kernel():
                self.dev.factor_cov(self.filter_coeffs)         # the factor the first rollout colours its draws with
            self._noise_valid = True
        replay = self._device_iteration if self._graph == "direct" else self._graph.replay
        if self._ahead == 0:
            self._action_np[self._slot(self.num_steps) + self.d_action] = -1.0      # completion flag (see _wait_action)
            replay()
            self._ahead = 1
        if self._lookahead and self._mono and self.dev.comm.world_size == 1:
            # iteration k + 1 goes into the queue before the host waits for the action of iteration k
            self._action_np[self._slot(self.num_steps + 1) + self.d_action] = -1.0
            replay()
            self._ahead += 1
        action = self._wait_action()
        self._ahead -= 1
        self.num_steps += 1
        self._step_host = self.num_steps
        self._mean_stale = True
        if not self._static_cov():
            self._cov_stale = True
        return action, 0.0

    def _capture_iteration(self):
        """Capture the control iteration as a hipGraph (``self._graph``; None if the runtime refused and every rank
        agreed to run eagerly)."""
        torch = self.dev.torch
        sharded_mono = self._mono and self.dev.comm.world_size > 1
        if self._mono:
            self._bind_mono(env_step=False)         # (the dry run below must not step the real env)
        # eager dry run on a side stream (allocates every buffer), with the state it must not consume
        keep = (self.dev.mean.clone(), self._step_dev.clone(), self.dev.cov.clone())
        side = torch.cuda.Stream(self.dev.device)
        side.wait_stream(torch.cuda.current_stream(self.dev.device))
        post, self._graph_post = self._graph_post, None
        with torch.cuda.stream(side):
            if self._noise_ahead():
                self._draw_raw(self.local_particles, 0)     # buffer + sampler parameters for the dry run
            if self._cem_in_kernel():
                self.dev.factor_cov(self.filter_coeffs)
            self._device_iteration()
        torch.cuda.current_stream(self.dev.device).wait_stream(side)
        torch.cuda.synchronize(self.dev.device)
        self._graph_post = post
        self.dev.mean.copy_(keep[0])
        self._step_dev.copy_(keep[1])
        self.dev.cov.copy_(keep[2])
        self._noise_valid = False       # the dry run left the samples of step + 1 behind
        err = None
        try:
            if sharded_mono:
                # the iteration reads one buffer of the mean and writes the other: one graph per direction, replayed in
                # turn; its last launch (the combine behind the all-gather) steps the real env when that is the engine's
                self._bind_mono(env_step=True)
                graphs = {}
                for _ in range(2):
                    g = torch.cuda.CUDAGraph()
                    key = id(self.dev.mean)
                    with torch.cuda.graph(g):
                        self._device_iteration()        # (swaps dev.mean / dev.mean_alt: the second pass is the way back)
                    graphs[key] = g
                self._graph = _AlternatingGraphs(self.dev, graphs)
            elif (getattr(self, "_want_tape", True) and _keep_graph_supported(torch)
                  and (getattr(self.dev.comm, "lib_collectives", False)
                       or (self.dev.comm.world_size == 1 and not getattr(self.dev.comm, "always_collective", False)))):
                # (sharded runs: the exchange is a library call too when the communicator issues it through the C ABI -
                # TorchDistComm.lib_collectives; a torch.distributed collective is not, and such an iteration stays a graph)
                # the iteration's library calls are recorded while it is captured; it then runs from that tape if the
                # tape is the whole iteration (as many kernel nodes in a capture of its replay as in the capture itself)
                tape = []
                g = torch.cuda.CUDAGraph(keep_graph=True)
                with torch.cuda.graph(g), _lib.recording(tape):
                    cap_stream = torch.cuda.current_stream(self.dev.device).cuda_stream
                    self._device_iteration()
                self._graph = self._tape_or_graph(g, tape, cap_stream)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._device_iteration()
                self._graph = g
        except Exception as e:          # e.g. a collective that cannot be captured
            err = e
        # Sharded runs decide TOGETHER (one all-reduce outside the capture): a rank that fell back alone would
        # issue a different collective sequence than its peers.  Every rank then runs eagerly, or none does.
        if not self.dev.comm.all_agree(err is None, self.dev.device):
            import warnings
            warnings.warn("hipGraph capture of the control iteration failed (%s); running eagerly"
                          % (err if err is not None else "on another rank",))
            torch.cuda.synchronize(self.dev.device)
            self.dev.mean.copy_(keep[0])
            self.dev.cov.copy_(keep[2])
            self._graph_on = False
            self._graph = None
            self.graph_fallback = True
            # a later enable_graph() / reset() starts clean: no half-bound launchers keyed by the (possibly swapped)
            # mean buffers, no one-launch mode
            self._mono = False
            self._mono_launch, self._mono_combine = {}, {}

    def _tape_or_graph(self, g, tape, cap_stream):
        """The launch tape of the iteration just captured in ``g`` if it reproduces the capture, else ``g``."""
        torch = self.dev.torch
        self.launch_mode = "hipGraph replay"
        try:
            t = _LaunchTape(tape, cap_stream, self.dev)
            g2 = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(g2):
                t.replay()
            n1, n2 = (ctypes.c_int64 * 2)(), (ctypes.c_int64 * 2)()
            _lib.check(self.dev.lib.mjmpc_graph_kernel_nodes(ctypes.c_void_p(int(g.raw_cuda_graph())), n1))
            _lib.check(self.dev.lib.mjmpc_graph_kernel_nodes(ctypes.c_void_p(int(g2.raw_cuda_graph())), n2))
            # ... and the same kernels in the same launch shapes, not merely as many (mjmpc_graph_signature)
            s1, s2 = (ctypes.c_uint64 * 2)(), (ctypes.c_uint64 * 2)()
            _lib.check(self.dev.lib.mjmpc_graph_signature(ctypes.c_void_p(int(g.raw_cuda_graph())), s1))
            _lib.check(self.dev.lib.mjmpc_graph_signature(ctypes.c_void_p(int(g2.raw_cuda_graph())), s2))
            del g2
            # (kernel nodes AND nodes of any kind: a collective or a tensor copy inside the iteration is not a library call)
            if n1[0] > 0 and n1[0] == n2[0] and n1[1] == n2[1] and s1[0] == s2[0] and s1[1] == s2[1]:
                self.launch_mode = "launch tape (%d calls, %d kernels)" % (len(t.calls), n1[0])
                return t
        except Exception as e:         # (an older runtime without raw graph access, a call that cannot be re-captured ...)
            import warnings
            warnings.warn("launch tape not available (%s); replaying the hipGraph" % (e,))
        g.instantiate()
        return g

    def _wait_action(self):
        """The action of the replayed iteration.  The fused update writes it into mapped pinned memory followed by
        the new step count; polling that flag returns the action as soon as it exists, while the rest of the
        graph (e.g. the captured env step) is still running - the next replay is enqueued behind it, so the GPU
        never waits for the host round trip.  (Sharded runs publish from the combine kernel after the all-gather.)
        Every tail of a captured iteration publishes this way: the fused MPPI / DMD-MPC update, the fused CEM step's
        finish launch, and ``mjmpc_step_tail`` for everything else."""
        A = self.d_action
        o = self._slot(self.num_steps)
        flag, want, spins = self._action_np, float(self.num_steps + 1), 0
        while flag[o + A] != want:
            spins += 1
            if spins > 2000000:                         # ~1 s without an answer: let the runtime report what happened
                self.dev.torch.cuda.current_stream(self.dev.device).synchronize()
                if flag[o + A] != want:
                    raise RuntimeError("captured control iteration finished without publishing its action")
        return flag[o:o + A].copy()

    def _slot(self, step):
        """Offset of the pinned slot the iteration of ``step`` publishes into (see _optimize_graphed)."""
        return (step & 1) * (self.d_action + 1) if self._mono else 0

    def _optimize_eager_after_fallback(self, state):
        action, value = Controller.optimize(self, state, False, True)
        if self._graph_post is not None:
            self._graph_post(self.dev.torch.from_numpy(np.ascontiguousarray(action)).to(self.dev.device))
        return action, value

    def optimize(self, state, calc_val=False, hotstart=True):
        if getattr(self, "graph_fallback", False) and not calc_val and hotstart:
            out = self._optimize_eager_after_fallback(state)
        elif self._graph_on and not calc_val and hotstart:
            out = self._optimize_graphed(state)
        else:
            out = super().optimize(state, calc_val, hotstart)
        self.dev.check_status()         # sampler error flags (host-visible memory: no copy, no synchronisation)
        return out

    # -- shift / reset (olgaussian_mpc.py:116-135) -------------------------------------------------
    def _shift(self):
        if self.base_action not in _SHIFT_MODES:
            raise NotImplementedError("invalid option for base action during shift")
        self._sync_in()
        row = None
        if self.base_action == 'random':
            row = np.random.normal(0, self.init_cov, self.d_action)
        self.dev.shift(_SHIFT_MODES[self.base_action], row)
        self._pull()

    def reset(self):
        if self._ahead:                 # an iteration enqueued ahead is still running: let it finish before the state goes
            self.dev.torch.cuda.synchronize(self.dev.device)
            self._ahead = 0
        self.num_steps = 0
        self.mean_action = np.zeros(shape=(self.horizon, self.d_action))
        self.cov_action = np.diag(self.init_cov)
        self.gamma_seq = np.cumprod([1.0] + [self.gamma] * (self.horizon - 1)).reshape(1, self.horizon)
        self._push()
        self._graph = None

    def _calc_val(self, trajectories):
        raise NotImplementedError("_calc_val not implemented")

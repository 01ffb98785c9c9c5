"""Device-side state and kernel calls shared by the controllers (torch = memory + streams only).

``DeviceUpdater`` owns the float64 device copies of the sampling distribution (mean ``(H,A)``,
cov ``(A,A)``), the discount sequence and the update workspace, and wraps the C-ABI update entry
points of include/mjmpc_amd.h.  Multi-GPU: every rank holds a contiguous block of particles (the
reference's worker mapping, subproc_vec_env.py:163-167); the small per-GPU *records* are exchanged
with ONE all-gather per reduction through ``comm`` and combined identically on every rank.
"""
import ctypes

import numpy as np

from .. import _lib
from . import mt_jump


def _vp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class SingleProcessComm:
    """World of one: the all-gather is the identity."""
    rank, world_size = 0, 1

    collectives, fell_back, why_fell_back, init_seconds, lib_collectives = "none", False, "", 0.0, False

    def all_gather(self, t):
        return t.reshape(1, -1)

    def all_gather_flat(self, t):
        return t

    def all_agree(self, ok, device=None):
        return bool(ok)


class TorchDistComm:
    """One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

    ``library_collectives`` (default: on where the backend is RCCL; ``MJMPC_TORCH_COLLECTIVES=1`` in the environment turns it
    off): the float64 all-gathers of the control iteration are issued by libmjmpc_amd.so itself on a communicator of its
    own (``mjmpc_comm_*``: the ranks of this group, the id handed out through one torch broadcast) - the same RCCL
    collective, but a LIBRARY call, so that a sharded iteration runs from the launch tape / as direct launches like the
    one-GPU loop instead of a hipGraph replay (controller.py).  Call ``close()`` before the process group
    is destroyed."""

    def __init__(self, group=None, library_collectives=None, device=None, init_timeout_s=None):
        """``device``: the GPU this rank's communicator lives on (default: the current CUDA device WHEN THIS IS CALLED - call
        ``torch.cuda.set_device`` first); ``init_timeout_s``: bound on the wait in ncclCommInitRank (default 60 s,
        ``MJMPC_COMM_INIT_TIMEOUT``): a rank stuck longer prints why and EXITS the process with code 3 - its peers are waiting
        for it inside a collective, there is nothing to fall back to from there."""
        import os
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._out = {}
        self._lib_comm = None
        self._gather_ranks = self.world_size
        self.device = None
        self.init_seconds = 0.0             # what making the library's communicator took (0: none was made)
        self.fell_back = False              # the library's communicator was asked for and is not in use
        self.why_fell_back = ""
        self.init_timeout_s = float(init_timeout_s if init_timeout_s is not None
                                    else os.environ.get("MJMPC_COMM_INIT_TIMEOUT", "60"))
        if library_collectives is None:
            library_collectives = self.backend == "nccl" and not os.environ.get("MJMPC_TORCH_COLLECTIVES")
        want = bool(library_collectives) and self.backend == "nccl"
        if self.backend == "nccl":
            import torch
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            # ONE all-reduce decides for the whole group: ranks that disagree about the switch (an environment variable set
            # on some of them) would otherwise enter different collectives and hang
            agreed = self.all_agree(want, self.device)
            if want and not agreed:
                self.fell_back, self.why_fell_back = True, "the ranks disagree about library_collectives / MJMPC_TORCH_COLLECTIVES"
            want = agreed
        self.lib_collectives = want
        if self.lib_collectives:
            self._open_library_comm()

    @property
    def collectives(self):
        """Which exchange path the control iterations of this communicator take (bench.py reports it)."""
        return "library RCCL" if self.lib_collectives else "torch.distributed"

    def _open_library_comm(self):
        """The library's communicator, made when this object is (the controllers read ``lib_collectives`` when they choose how
        to launch).  Every rank first checks that the library can reach RCCL at all; the ranks then go on TOGETHER or fall
        back to torch.distributed's collectives together (a rank that could not bind RCCL would otherwise leave the others
        waiting in ncclCommInitRank)."""
        import time
        import warnings
        dev = self.device
        ok, why = True, ""
        try:
            probe = (ctypes.c_ubyte * 128)()
            _lib.check(_lib.load().mjmpc_comm_unique_id(probe))
        except Exception as e:      # (no librccl the library can bind, an older library ...)
            ok, why = False, str(e)
        if self.all_agree(ok, dev):
            t0 = time.perf_counter()
            try:
                self._library_comm(dev)
            except Exception as e:
                ok, why = False, str(e)
            self.init_seconds = time.perf_counter() - t0
            ok = self.all_agree(ok, dev)
        else:
            ok = False
        if not ok:
            self.close()
            self.lib_collectives = False
            self.fell_back, self.why_fell_back = True, why or "another rank failed"
            warnings.warn("mjmpc_amd: the library's own RCCL communicator is not available (%s); the control iterations' exchanges go "
                          "through torch.distributed (hipGraph replay instead of direct launches)" % self.why_fell_back)

    def _library_comm(self, device):
        """ncclCommInitRank of the library's own communicator over the ranks of this group (collective; outside any
        stream capture: the controllers' dry run / first eager iteration gets here first).  The call runs on a helper thread
        so that the wait can be bounded (``init_timeout_s``)."""
        import os
        import sys
        import threading
        import torch
        lib = _lib.load()
        ident = torch.zeros(128, dtype=torch.uint8, device=device)
        if self.rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            _lib.check(lib.mjmpc_comm_unique_id(buf))
            ident.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
        src = self._dist.get_global_rank(self._group, 0) if self._group is not None else 0
        self._dist.broadcast(ident, src=src, group=self._group)
        raw = bytes(ident.cpu().numpy().tobytes())
        h = ctypes.c_void_p()
        box = {}

        def init():
            try:
                box["rc"] = lib.mjmpc_comm_create(ctypes.c_char_p(raw), self._dist.get_world_size(self._group),
                                                  self._dist.get_rank(self._group),
                                                  device.index if device.index is not None else torch.cuda.current_device(),
                                                  ctypes.byref(h))
            except BaseException as e:      # noqa: B036 - handed to the caller's thread
                box["exc"] = e

        th = threading.Thread(target=init, name="mjmpc-comm-init", daemon=True)
        th.start()
        th.join(self.init_timeout_s)
        if th.is_alive():
            # ncclCommInitRank has not returned: some rank never arrived (or the fabric is down).  Nothing can be unwound from
            # here - the peers sit in the same call - so say so and end THIS process (an exit, never a re-exec: this process
            # has initialised the GPU)
            sys.stderr.write("mjmpc_amd: rank %d waited %.0f s in ncclCommInitRank for the library's communicator (world size %d); "
                             "giving up. Set MJMPC_TORCH_COLLECTIVES=1 to run the exchanges through torch.distributed.\n"
                             % (self.rank, self.init_timeout_s, self.world_size))
            sys.stderr.flush()
            os._exit(3)
        if "exc" in box:
            raise box["exc"]
        _lib.check(box["rc"])
        self._lib_comm = (lib, h)

    def close(self):
        if self._lib_comm is not None:
            lib, h = self._lib_comm
            self._lib_comm = None
            lib.mjmpc_comm_destroy(h)

    def __del__(self):
        # (not at interpreter teardown, and not once the process group is gone: ncclCommDestroy after HIP / RCCL shut down
        # is undefined - call close() before destroy_process_group())
        try:
            import sys
            if self._lib_comm is not None and not sys.is_finalizing() and self._dist.is_initialized():
                self.close()
        except Exception:
            pass

    def all_gather(self, t):
        import torch
        key = (t.numel(), t.dtype, t.device)
        out = self._out.get(key)            # persistent receive buffer: a captured graph replays into it
        if out is None:
            out = self._out[key] = torch.empty(self.world_size * t.numel(), dtype=t.dtype, device=t.device)
        if self.lib_collectives and t.is_cuda and t.dtype == torch.float64:
            if t.device != self.device:
                raise ValueError("all_gather of a tensor on %s through a communicator made on %s" % (t.device, self.device))
            lib, h = self._lib_comm
            src = t.reshape(-1).contiguous()
            if src.data_ptr() != t.data_ptr():
                self._keep = src            # (a copy made for contiguity must outlive the asynchronous collective)
            _lib.check(lib.mjmpc_comm_all_gather_f64(h, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), src.numel(),
                                                     ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))
            return out.reshape(self.world_size, t.numel())
        self._dist.all_gather_into_tensor(out, t.reshape(-1).contiguous(), group=self._group)   # flat: gloo insists
        return out.reshape(self.world_size, t.numel())

    def all_gather_launcher(self, t):
        """``launch()`` = ``all_gather(t)`` into its persistent buffer with the arguments bound once (the host leg of a
        sharded control step: one C call on the stream that is current when it runs)."""
        out = self.all_gather(t)            # (makes the buffer and, the first time, the library's communicator)
        if not (self.lib_collectives and self._lib_comm is not None and t.is_contiguous()):
            return lambda: self.all_gather(t)
        import torch
        lib, h = self._lib_comm
        fn, check, dev = lib.mjmpc_comm_all_gather_f64, _lib.check, t.device
        args = (h, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()), t.numel())
        stream = torch.cuda.current_stream

        def launch(_keep=(t, out)):
            check(fn(*args, ctypes.c_void_p(stream(dev).cuda_stream)))
        return launch

    def all_gather_flat(self, t):
        return self.all_gather(t).reshape(-1)

    def all_agree(self, ok, device=None):
        """True iff ``ok`` holds on EVERY rank (an all-reduce(MIN) outside any capture): ranks use it to take a
        code path together - e.g. to drop from graph replay to eager launches as one, never alone."""
        import torch
        dev = device if self.backend == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self._group)
        return bool(t.item())


class DeviceUpdater:
    def __init__(self, horizon, d_action, gamma_seq, device=0, comm=None):
        import torch
        self.lib = _lib.require_gpu()
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.H, self.A = int(horizon), int(d_action)
        self.comm = comm or SingleProcessComm()
        g = np.asarray(gamma_seq, np.float64).reshape(-1)
        self.gamma_zero = int(bool(np.any(g == 0)))
        self.gseq = torch.from_numpy(g.copy()).to(self.device)
        self.mean = torch.zeros((self.H, self.A), dtype=torch.float64, device=self.device)
        self.mean_alt = torch.zeros_like(self.mean)       # the fused iteration writes the new mean here, then the two swap
        self.cov = torch.zeros((self.A, self.A), dtype=torch.float64, device=self.device)
        self.covinv = torch.zeros((self.A, self.A), dtype=torch.float64, device=self.device)
        self.value = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.wnorm = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._ws, self._ws_P = None, -1
        self._rec = {}
        # kernel-side error flags live in mapped pinned host memory (like the published action): the kernels store
        # to them directly, the host reads them without a copy or a synchronisation (check_status)
        self._status = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._status_np = self._status.numpy()
        self.chol_status = self._status[0:1]
        self.mt_status = self._status[1:2]
        self.mt_segments = 32            # workgroups generating the MT19937 stream in parallel (0/1 = serial)

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, P):
        if self._ws is None or self._ws_P < P:
            nbytes = self.lib.mjmpc_update_workspace_bytes(P, self.H, self.A)
            self._ws = self.torch.empty((nbytes + 7) // 8, dtype=self.torch.float64, device=self.device)
            self._ws_P = P
        return self._ws

    def record(self, name, n):
        t = self._rec.get(name)
        if t is None or t.numel() != n:
            t = self.torch.empty(n, dtype=self.torch.float64, device=self.device)
            self._rec[name] = t
        return t

    def to_device(self, x, name):
        """numpy or tensor -> contiguous CUDA tensor (f32 stays f32, everything else f64)."""
        torch = self.torch
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float64)
        return x.to(self.device).contiguous()

    @staticmethod
    def code(t):
        import torch
        return _lib.F32 if t.dtype == torch.float32 else _lib.F64

    def set_mean(self, mean):
        self.mean.copy_(self.torch.from_numpy(np.ascontiguousarray(mean, np.float64)))

    def set_cov(self, cov):
        self.cov.copy_(self.torch.from_numpy(np.ascontiguousarray(cov, np.float64)))

    def get_mean(self):
        return self.mean.cpu().numpy()

    def get_cov(self):
        return self.cov.cpu().numpy()

    def _pair(self, costs, actions):
        costs = self.to_device(costs, "costs")
        actions = self.to_device(actions, "actions")
        if costs.dtype != actions.dtype:
            costs = costs.to(actions.dtype)
        P = costs.shape[0]
        if tuple(costs.shape) != (P, self.H) or tuple(actions.shape) != (P, self.H, self.A):
            raise ValueError("costs must be (P,H) and actions (P,H,A)")
        return costs, actions, P

    # ------------------------------------------------------------------ softmax family (MPPI / DMD / PFMPC)
    def td_lambda_returns(self, costs, actions, qvals, beta, alpha, gamma, td_lam, covinv=None):
        """MPPIQ.calculate_returns on the device (mppiq.py:104-136) -> (P,H) tensor of the costs' dtype."""
        torch = self.torch
        costs, actions, P = self._pair(costs, actions)
        if qvals is not None:
            qvals = self.to_device(qvals, "qvals").to(costs.dtype).contiguous()
            if tuple(qvals.shape) != (P, self.H):
                raise ValueError("qvals must have shape (P, H)")
        wseq = np.cumprod([1.0] + [gamma * td_lam] * (self.H - 2)) if self.H > 1 else np.array([1.0])
        wd = self.record("td_wseq", max(self.H - 1, 1))
        wd.copy_(torch.from_numpy(np.ascontiguousarray(wseq, np.float64)))
        if alpha == 0:
            self.covinv.copy_(torch.from_numpy(np.ascontiguousarray(covinv, np.float64)))
        key = ("td_returns", costs.dtype)
        out = self._rec.get(key)
        if out is None or tuple(out.shape) != (P, self.H):
            out = self._rec[key] = torch.empty((P, self.H), dtype=costs.dtype, device=self.device)
        _lib.check(self.lib.mjmpc_td_lambda_returns(self.code(costs), P, self.H, self.A, _vp(costs), _vp(actions),
                                                    _vp(qvals), _vp(self.mean), _vp(self.covinv), _vp(wd),
                                                    int(bool(np.any(wseq == 0))), float(beta), int(alpha), float(gamma),
                                                    float(td_lam), _vp(out), _vp(self.workspace(P)), self.stream()))
        return out

    def softmax_update(self, costs, actions, lam, step_size, alpha=1, time_based_weights=False, cov_mode=0,
                       covinv=None, want_value=False, update_mean=True, costs_are_returns=False, replicated=False):
        """``costs_are_returns``: the (P,H) input already holds per-step returns - no cost_to_go (MPPIQ).
        ``replicated``: every rank passes ALL particles (PFMPC after its cost gather) - no record exchange."""
        costs, actions, P = self._pair(costs, actions)
        tbw = int(bool(time_based_weights))
        n = self.lib.mjmpc_softmax_record_len(self.H, self.A, tbw)
        rec = self.record("softmax", n)
        if alpha == 0:
            self.covinv.copy_(self.torch.from_numpy(np.ascontiguousarray(covinv, np.float64)))
        ws = self.workspace(P)
        _lib.check(self.lib.mjmpc_softmax_stats(self.code(costs), P, self.H, self.A, _vp(costs), _vp(actions),
                                                _vp(self.mean), _vp(self.covinv), _vp(self.gseq),
                                                1 if costs_are_returns else self.gamma_zero,
                                                float(lam), int(alpha), tbw, int(cov_mode != 0), _vp(rec), _vp(ws),
                                                self.stream()))
        recs = rec.reshape(1, -1) if replicated else self.comm.all_gather(rec)
        G = recs.shape[0]
        mean_out = self.mean if update_mean else self.record("mean_scratch", self.H * self.A).copy_(self.mean.reshape(-1))
        _lib.check(self.lib.mjmpc_softmax_combine(_vp(recs), G, self.H, self.A, tbw, float(lam), float(step_size),
                                                  int(cov_mode), float(P * G), _vp(mean_out),
                                                  _vp(self.cov) if cov_mode else None,
                                                  _vp(self.value) if want_value else None, _vp(self.wnorm),
                                                  self.stream()))
        return P

    def zero_actions(self, P, like):
        """(P,H,A) zeros of ``like``'s dtype on the device (weights-only calls need no actions)."""
        key = ("zero_actions", like.dtype)
        t = self._rec.get(key)
        if t is None or t.shape[0] != P:
            t = self.torch.zeros((P, self.H, self.A), dtype=like.dtype, device=self.device)
            self._rec[key] = t
        return t

    def softmax_weights(self, P):
        w = self.record("weights", P)
        _lib.check(self.lib.mjmpc_softmax_weights(P, self.H, self.A, _vp(self.wnorm), _vp(self.workspace(P)), _vp(w),
                                                  self.stream()))
        return w

    def mppi_fused_update(self, q0, actions, lam, step_size, shift_mode, action_out, action_pinned=None,
                          step_counter=None, draw_next=None):
        """q0 (float64 [P], device) + actions -> mean update, action read-out and shift in two launches.
        Sharded: the fused kernels only produce this GPU's record; one all-gather; then the combine.
        ``draw_next`` = dict(seed, offset, particle_offset, d_step): the same launches also draw the next control
        step's raw samples into the sampler's buffer (parameters as for ``sample_noise``, set up by its last call)."""
        P = q0.shape[0]

        def fused(step, shift, mean_io, act, rec, pinned, counter):
            args = [self.code(actions), P, self.H, self.A, _vp(q0), _vp(actions), float(lam), float(step), int(shift),
                    _vp(mean_io), _vp(act), _vp(rec), None, _vp(pinned), _vp(counter), _vp(self.workspace(P))]
            if draw_next is None:
                _lib.check(self.lib.mjmpc_mppi_fused_update(*args, self.stream()))
            else:
                buf = self._rec[("noise", "f32" if actions.dtype == self.torch.float32 else "f64")]
                _lib.check(self.lib.mjmpc_mppi_fused_update_draw_next(
                    *args, _vp(buf), _vp(self._rec["chol"]), int(draw_next["seed"]) & (2 ** 64 - 1),
                    int(draw_next["offset"]), int(draw_next["particle_offset"]), _vp(draw_next["d_step"]),
                    self._rec["chol_diag"], self.stream()))

        if self.comm.world_size == 1:
            fused(step_size, shift_mode, self.mean, action_out, None, action_pinned, step_counter)
            return
        rec = self.record("fused_rec", 2 + self.H * self.A)        # [xmax | S | W[H*A]]: the layout of a partial
        fused(0.0, -1, self.mean, None, rec, None, None)
        recs = self.comm.all_gather(rec)
        G = recs.shape[0]
        _lib.check(self.lib.mjmpc_mppi_fused_combine(_vp(recs), G, float(P * G), self.H, self.A, float(lam),
                                                     float(step_size), int(shift_mode), _vp(self.mean),
                                                     _vp(action_out), None, _vp(action_pinned), _vp(step_counter),
                                                     self.stream()))

    # ------------------------------------------------------------------ CEM
    def cem_update(self, costs, actions, num_elite, step_size, full_cov, q0=None):
        """``q0``: cost_to_go(costs)[:, 0] when the rollout launch has already produced it (float64 [P], device)."""
        costs, actions, P = self._pair(costs, actions)
        ws = self.workspace(P)
        code = self.code(costs)
        G, rank = self.comm.world_size, self.comm.rank
        if q0 is not None:
            self._take_q0(ws, P, q0)
        else:
            _lib.check(self.lib.mjmpc_traj_cost(code, P, self.H, self.A, _vp(costs), _vp(self.gseq), self.gamma_zero,
                                                _vp(ws), self.stream()))
        q_all_ptr, P_all = None, P
        if G > 1 or getattr(self.comm, "always_collective", False):    # (the flag: one-GPU tests of the sharded path)
            q0 = self._q0_view(ws, P)
            q_all = self.comm.all_gather_flat(q0)
            q_all_ptr, P_all = _vp(q_all), P * G
        # one record per GPU: { n_g | sum of elite actions | scatter of ITS elite deltas about ITS OWN mean } - the
        # second (and last) exchange of the iteration; mjmpc_cem_combine pools the scatters
        HA = self.H * self.A
        rec = self.record("cem_rec", 1 + HA + self.A * self.A)
        srec, crec = rec[:1 + HA], rec[1 + HA:]
        _lib.check(self.lib.mjmpc_cem_elite_sums(code, P, self.H, self.A, _vp(actions), q_all_ptr, P_all, rank * P,
                                                 int(num_elite), _vp(srec), _vp(ws), self.stream()))
        _lib.check(self.lib.mjmpc_cem_elite_cov(code, P, self.H, self.A, _vp(actions), _vp(self.mean), _vp(srec), 1,
                                                _vp(crec), _vp(ws), self.stream()))
        recs = self.comm.all_gather(rec)
        _lib.check(self.lib.mjmpc_cem_combine(_vp(recs), G, self.H, self.A, float(num_elite), int(full_cov),
                                              float(step_size), _vp(self.mean), _vp(self.cov), self.stream()))

    def factor_cov(self, filter_coeffs):
        """Lower Cholesky factor of the device-resident covariance into the ``chol`` record, and the filter coefficients
        into theirs (what ``sample_noise(cov=None)`` sets up before it draws)."""
        torch = self.torch
        chol = self.record("chol", self.A * self.A)
        _lib.check(self.lib.mjmpc_cholesky_lower(_vp(self.cov), self.A, _vp(chol), _vp(self.chol_status), self.stream()))
        fc = np.asarray(filter_coeffs, np.float64)
        cached = self._rec.get("noise_params")
        if cached is None or cached[0] is not None or not np.array_equal(cached[1], fc):
            self.record("coeffs", 3).copy_(torch.from_numpy(fc.copy()))
            self._rec["noise_params"] = (None, fc.copy())
        return chol

    def cem_fused_supported(self, P, num_elite):
        """The two-launch CEM step (``cem_fused_step``) covers this shape (A <= 8, P x world <= 32768, ...)."""
        return bool(self.lib.mjmpc_cem_fused_supported(P * self.comm.world_size, P, int(num_elite), self.H, self.A))

    def cem_fused_step(self, actions, num_elite, step_size, full_cov, shift_mode, action_out, action_pinned, step_counter,
                       grow, next_noise, seed, particle_offset):
        """One CEM update + the tail of the control step in two launches (three and two exchanges when sharded):
        selection + elite list + moments (``mjmpc_cem_select_moments``; q0 is where the rollout launch left it), then refit +
        covariance growth + Cholesky factor + action + shift + step counter + the NEXT step's raw samples into
        ``next_noise`` (``mjmpc_cem_finish``).  ``grow`` = (diag tensor or None, scale) or None (cem.py:94)."""
        P = actions.shape[0]
        ws = self.workspace(P)
        code = self.code(actions)
        G, rank = self.comm.world_size, self.comm.rank
        q_all_ptr, P_all = None, P
        if G > 1 or getattr(self.comm, "always_collective", False):
            q_all = self.comm.all_gather_flat(self._q0_view(ws, P))
            q_all_ptr, P_all = _vp(q_all), P * G
        _lib.check(self.lib.mjmpc_cem_select_moments(code, P, self.H, self.A, _vp(actions), q_all_ptr, P_all, rank * P,
                                                     int(num_elite), _vp(self.mean), _vp(self.cov), _vp(step_counter),
                                                     _vp(ws), self.stream()))
        recs_ptr, n_rec = None, 1
        if q_all_ptr is not None:
            rec = self.record("cem_rec", 1 + self.H * self.A + self.A * self.A)
            _lib.check(self.lib.mjmpc_cem_record(P, self.H, self.A, int(num_elite), _vp(self.mean), _vp(rec), _vp(ws), self.stream()))
            recs = self.comm.all_gather(rec)
            recs_ptr, n_rec = _vp(recs), recs.shape[0]
        gd, gs = (None, 0.0) if grow is None else (self._cov_diag(grow[0]), grow[1])
        chol = self.record("chol", self.A * self.A)
        _lib.check(self.lib.mjmpc_cem_finish(code, P, self.H, self.A, int(num_elite), recs_ptr, n_rec, float(num_elite),
                                             int(full_cov), float(step_size), int(shift_mode), _vp(self.mean), _vp(self.cov),
                                             _vp(chol), _vp(self.chol_status), _vp(gd), float(gs), _vp(action_out),
                                             _vp(action_pinned), _vp(step_counter), _vp(next_noise),
                                             int(seed) & (2 ** 64 - 1), 0, int(particle_offset), _vp(ws), self.stream()))

    def check_status(self):
        """Raise if a sampler kernel has flagged an error: an indefinite covariance would otherwise turn into NaN
        samples, mean and action without a word, and a short MT19937 stream would leave stale samples behind.  The kernels
        only ever SET their flag (sticky: a later successful draw does not erase an earlier failure); the host clears
        exactly the flags it reports, and reports both when both are up.  Eager iterations: read behind every kernel of
        the control step that has just returned its action.  Captured iterations return the action while the graph's
        tail (the next step's Cholesky factor and samples) is still running, so a failure of that tail raises one
        ``optimize()`` later - on the step whose samples it spoiled."""
        st = self._status_np
        chol, mt = int(st[0]), int(st[1])
        if not (chol or mt):
            return
        msgs = []
        if chol:
            st[0] = 0
            msgs.append("the action covariance on the device is indefinite or not finite: its Cholesky factor (sampler "
                        "colouring) does not exist")
        if mt:
            st[1] = 0
            msgs.append("device MT19937 sampler: the generated stream was too short for the requested draw (status %d); "
                        "samples are incomplete" % mt)
        raise _lib.MjmpcError("; ".join(msgs))

    def q0_destination(self, P):
        """Where the updates expect cost_to_go(costs)[:, 0] (float64 [P], a view of the workspace): a rollout launch that
        writes its q0 there saves the copy."""
        return self._q0_view(self.workspace(P), P)

    def _take_q0(self, ws, P, q0):
        view = self._q0_view(ws, P)
        if q0.data_ptr() != view.data_ptr():
            view.copy_(q0)

    def _q0_view(self, ws, P):
        addr = self.lib.mjmpc_workspace_q0(_vp(ws), P, self.H, self.A)
        off = (addr - ws.data_ptr()) // 8
        return ws[off:off + P]

    # ------------------------------------------------------------------ random shooting
    def rs_update(self, costs, actions, step_size, q0=None):
        costs, actions, P = self._pair(costs, actions)
        ws = self.workspace(P)
        code = self.code(costs)
        if q0 is not None:
            self._take_q0(ws, P, q0)
        else:
            _lib.check(self.lib.mjmpc_traj_cost(code, P, self.H, self.A, _vp(costs), _vp(self.gseq), self.gamma_zero,
                                                _vp(ws), self.stream()))
        rec = self.record("rs", 2 + self.H * self.A)
        _lib.check(self.lib.mjmpc_rs_best(code, P, self.H, self.A, _vp(actions), self.comm.rank * P, _vp(rec), _vp(ws),
                                          self.stream()))
        recs = self.comm.all_gather(rec)
        _lib.check(self.lib.mjmpc_rs_combine(_vp(recs), recs.shape[0], self.H, self.A, float(step_size),
                                             _vp(self.mean), self.stream()))

    def mean_q0(self, costs):
        """CEM / RandomShooting _calc_val: average cost-to-go over ALL particles."""
        costs = self.to_device(costs, "costs")
        P = costs.shape[0]
        ws = self.workspace(P)
        _lib.check(self.lib.mjmpc_traj_cost(self.code(costs), P, self.H, self.A, _vp(costs), _vp(self.gseq),
                                            self.gamma_zero, _vp(ws), self.stream()))
        s = self.record("q0sum", 1)
        _lib.check(self.lib.mjmpc_q0_sum(P, self.H, self.A, _vp(s), _vp(ws), self.stream()))
        tot = self.comm.all_gather(s)
        return float(tot.sum().item()) / (P * tot.shape[0])

    # ------------------------------------------------------------------ shift / noise
    def shift(self, mode, row=None):
        row_d = None
        if mode == 2:
            row_d = self.record("shift_row", self.A)
            row_d.copy_(self.torch.from_numpy(np.ascontiguousarray(row, np.float64)))
        _lib.check(self.lib.mjmpc_shift_mean(_vp(self.mean), self.H, self.A, int(mode), _vp(row_d), self.stream()))

    def _cov_diag(self, diag):
        """The device copy of a covariance-growth diagonal (None = identity), uploaded when it changes."""
        if diag is None:
            return None
        d = self.record("cov_diag", self.A)
        cached = self._rec.get("cov_diag_host")
        diag = np.ascontiguousarray(diag, np.float64)
        if cached is None or not np.array_equal(cached, diag):
            d.copy_(self.torch.from_numpy(diag.copy()))
            self._rec["cov_diag_host"] = diag.copy()
        return d

    def add_cov_diag(self, diag, scale):
        """cov += scale * diag(diag) on the device (diag None: identity)."""
        _lib.check(self.lib.mjmpc_cov_add_diag(_vp(self.cov), self.A, _vp(self._cov_diag(diag)), float(scale),
                                               self.stream()))

    def step_tail(self, mode, action_out, action_pinned, step_counter, grow_cov=None):
        """The end of a device-resident control step in one launch (``mjmpc_step_tail``): action = mean[0] to the device
        and the pinned host buffer, shift (mode 0 'null' / 1 'repeat'), step counter + 1, and - ``grow_cov`` = (diag or
        None, scale) - the covariance growth of the shift."""
        d, scale = (self._cov_diag(grow_cov[0]), float(grow_cov[1])) if grow_cov is not None else (None, 0.0)
        _lib.check(self.lib.mjmpc_step_tail(_vp(self.mean), self.H, self.A, int(mode), None, _vp(action_out),
                                            _vp(action_pinned), _vp(step_counter),
                                            _vp(self.cov) if grow_cov is not None else None, _vp(d), scale, self.stream()))

    def sample_noise_mt19937(self, P, cov, filter_coeffs, seed, offset, dtype="f64", d_step=None, filtered=True,
                             particle_offset=0):
        """The reference's own noise (legacy numpy stream of ``np.random.seed(seed + offset)``) regenerated
        on the device.  Isotropic covariance c*I: bit-identical up to libm's last bit.  General covariance: numpy
        colours the standard-normal stream with ``B = sqrt(s)[:, None] * v`` from ``svd(cov)`` (LAPACK, on the host
        here as there) - the same stream, the same B, a fixed-order product on the device instead of BLAS' ``dot``
        (agreement ~1e-15 relative, not bit-for-bit).  Returns the (P,H,A) tensor, filtered unless told not to.
        ``particle_offset``: global index of local particle 0 - a rank of a sharded run keeps its own block of the
        one stream (and regenerates the stream up to the end of that block: rejections make positions data dependent)."""
        torch = self.torch
        cov = np.asarray(cov, np.float64)
        c = float(cov[0, 0])
        general = np.count_nonzero(cov - c * np.eye(self.A)) != 0
        if general:
            cached = self._rec.get("mt_cov")
            if cached is None or not np.array_equal(cached, cov):
                _, sv, vt = np.linalg.svd(cov)                       # what np.random.multivariate_normal does
                B = np.sqrt(sv)[:, None] * vt
                self.record("mt_B", self.A * self.A).copy_(torch.from_numpy(np.ascontiguousarray(B).reshape(-1)))
                self._rec["mt_cov"] = cov.copy()
            c = 1.0
        tdt = torch.float32 if dtype == "f32" else torch.float64
        key = ("noise_mt", dtype)
        buf = self._rec.get(key)
        n = P * self.H * self.A
        first = int(particle_offset) * self.H * self.A
        if buf is None or tuple(buf.shape) != (P, self.H, self.A) or self._rec.get("mt_first") != first:
            buf = self._rec[key] = torch.empty((P, self.H, self.A), dtype=tdt, device=self.device)
            self._rec["mt_first"] = first
            nbytes = self.lib.mjmpc_mt19937_workspace_bytes(first + n)
            self._rec["mt_ws"] = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.float64, device=self.device)
            self._rec["mt_status"] = self.mt_status
            # jump-ahead plan: serial head + MT_SEGMENTS workgroups (tables are host-computed once per size)
            head, seg, nseg = mt_jump.plan_segments(int(self.lib.mjmpc_mt19937_stream_words(first + n)), self.mt_segments)
            if nseg:
                idx, starts = mt_jump.jump_tables(seg, nseg, head)
                self._rec["mt_jump"] = (torch.from_numpy(idx.copy()).to(self.device),
                                        torch.from_numpy(starts.copy()).to(self.device), head, seg, nseg)
            else:
                self._rec["mt_jump"] = (None, None, 0, 0, 0)
        fc = np.asarray(filter_coeffs, np.float64)
        co = self.record("coeffs", 3)
        cached = self._rec.get("mt_coeffs")
        if cached is None or not np.array_equal(cached, fc):
            co.copy_(torch.from_numpy(fc.copy()))
            self._rec["mt_coeffs"] = fc.copy()
        jidx, jstarts, head, seg, nseg = self._rec["mt_jump"]
        _lib.check(self.lib.mjmpc_sample_noise_mt19937_jump(
            _lib.F32 if dtype == "f32" else _lib.F64, _vp(buf), n, float(np.sqrt(c)),
            (int(seed) + int(offset)) & (2 ** 64 - 1), _vp(d_step), _vp(jidx), _vp(jstarts), head, seg, nseg, first,
            _vp(self._rec["mt_ws"]), _vp(self._rec["mt_status"]), self.stream()))
        if general:
            _lib.check(self.lib.mjmpc_color_noise(_lib.F32 if dtype == "f32" else _lib.F64, _vp(buf), P * self.H, self.A,
                                                  _vp(self._rec["mt_B"]), self.stream()))
        if filtered and not (fc[0] == 1.0 and fc[1] == 0.0 and fc[2] == 0.0):
            _lib.check(self.lib.mjmpc_filter_noise(_lib.F32 if dtype == "f32" else _lib.F64, _vp(buf), P, self.H, self.A,
                                                   _vp(co), self.stream()))
        return buf

    def prepare_noise(self, cov, filter_coeffs):
        """Upload the sampler's parameters for a HOST covariance (Cholesky factor, filter coefficients) without drawing:
        what ``sample_noise`` does ahead of its launch.  Returns (chol, coeffs, chol_is_diagonal)."""
        torch = self.torch
        fc = np.asarray(filter_coeffs, np.float64)
        cov = np.asarray(cov, np.float64)
        cached = self._rec.get("noise_params")
        if cached is None or cached[0] is None or not (np.array_equal(cached[0], cov) and np.array_equal(cached[1], fc)):
            self.record("chol", self.A * self.A).copy_(torch.from_numpy(np.linalg.cholesky(cov).reshape(-1).copy()))
            self.record("coeffs", 3).copy_(torch.from_numpy(fc.copy()))
            self._rec["noise_params"] = (cov.copy(), fc.copy())
            self._rec["chol_diag"] = int(np.count_nonzero(cov - np.diag(np.diag(cov))) == 0)
        return self._rec["chol"], self._rec["coeffs"], self._rec["chol_diag"]

    def sample_noise(self, P, cov, filter_coeffs, seed, offset, dtype="f64", particle_offset=0, d_step=None,
                     filtered=True, device_cov_diagonal=False):
        """Philox noise coloured by ``cov`` (host array), or - ``cov=None`` - by the device-resident ``self.cov``
        (``device_cov_diagonal`` promises it has no off-diagonal entries)."""
        torch = self.torch
        tdt = torch.float32 if dtype == "f32" else torch.float64
        key = ("noise", dtype)
        buf = self._rec.get(key)
        if buf is None or tuple(buf.shape) != (P, self.H, self.A):
            buf = torch.empty((P, self.H, self.A), dtype=tdt, device=self.device)
            self._rec[key] = buf
        fc = np.asarray(filter_coeffs, np.float64)
        if cov is None:
            # the covariance lives on the device (self.cov): factor it there - nothing crosses PCIe
            chol = self.record("chol", self.A * self.A)
            _lib.check(self.lib.mjmpc_cholesky_lower(_vp(self.cov), self.A, _vp(chol), _vp(self.chol_status),
                                                     self.stream()))
            cached = self._rec.get("noise_params")
            if cached is None or cached[0] is not None or not np.array_equal(cached[1], fc):
                self.record("coeffs", 3).copy_(torch.from_numpy(fc.copy()))
                self._rec["noise_params"] = (None, fc.copy())
            self._rec["chol_diag"] = int(bool(device_cov_diagonal))
        else:
            self.prepare_noise(cov, fc)
        chol, co = self._rec["chol"], self._rec["coeffs"]
        _lib.check(self.lib.mjmpc_sample_noise(_lib.F32 if dtype == "f32" else _lib.F64, _vp(buf), P, self.H, self.A,
                                               _vp(chol), _vp(co) if filtered else None, int(seed) & (2 ** 64 - 1),
                                               int(offset),
                                               int(particle_offset), _vp(d_step), self._rec["chol_diag"],
                                               self.stream()))
        return buf

"""GPU rollout engines for the reference's two analytic environments (Pendulum, LQR).

Same reference-shaped interface as ``ArmRolloutEngine`` (``set_env_state`` / ``rollout`` / ``reset`` /
``close``); state dicts are the envs' own ``{'state': ndarray}`` (pendulum.py:106-110, lqr.py:76-80).
"""
import ctypes
import time

import numpy as np

from .. import _lib

KIND_PENDULUM, KIND_LQR = 0, 1


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class AnalyticRolloutEngine:
    def __init__(self, kind, params, d_state, d_obs, d_action, device=0, dtype="f64", num_shards=1):
        import torch
        self._lib = _lib.require_gpu()
        self.torch = torch
        self.kind = kind
        self.device = torch.device("cuda", device)
        self.dtype = dtype
        self._code = _lib.F32 if dtype == "f32" else _lib.F64
        self._tdtype = torch.float32 if dtype == "f32" else torch.float64
        self.d_state, self.d_obs, self.d_action = d_state, d_obs, d_action
        self.num_shards = num_shards
        self._params = torch.from_numpy(np.ascontiguousarray(params, np.float64).reshape(-1).copy()).to(self.device)
        self._state = torch.zeros(d_state, dtype=torch.float64, device=self.device)
        self.closed = False

    @classmethod
    def pendulum(cls, g=10.0, **kw):
        """PendulumEnv.__init__ constants (pendulum.py:13-19)."""
        return cls(KIND_PENDULUM, [8.0, 2.0, 0.05, g, 1.0, 1.0], d_state=2, d_obs=3, d_action=1, **kw)

    @classmethod
    def lqr(cls, A, B, Q, R, **kw):
        A, B, Q, R = (np.asarray(x, np.float64) for x in (A, B, Q, R))
        n, m = A.shape[0], B.shape[1]
        if n > 8 or m > 8:
            raise ValueError("LQR kernel supports state / action dimensions up to 8")
        return cls(KIND_LQR, np.concatenate([A.reshape(-1), B.reshape(-1), Q.reshape(-1), R.reshape(-1)]),
                   d_state=n, d_obs=n, d_action=m, **kw)

    def set_env_state(self, state_dicts):
        s = state_dicts[0] if isinstance(state_dicts, (list, tuple)) else state_dicts
        x = np.ascontiguousarray(s["state"], np.float64).reshape(-1)
        if x.size != self.d_state:
            raise ValueError("state has %d entries, expected %d" % (x.size, self.d_state))
        self._state.copy_(self.torch.from_numpy(x.copy()))

    def rollout_device(self, num_particles, horizon, mean, noise, mode="open_loop", want_obs=False):
        if mode not in ("open_loop", "closed_loop_linear"):
            raise ValueError("unsupported rollout mode %r ('open_loop' or 'closed_loop_linear')" % (mode,))
        closed = mode == "closed_loop_linear"
        if num_particles % self.num_shards != 0:
            raise AssertionError("Number of particles must be divisible by number of shards")
        torch = self.torch
        P, H, A = int(num_particles), int(horizon), self.d_action

        def dev(x, dt, shape):
            if not isinstance(x, torch.Tensor):
                x = torch.from_numpy(np.ascontiguousarray(x))
            if tuple(x.shape) != shape:
                raise ValueError("expected shape %s, got %s" % (shape, tuple(x.shape)))
            return x.to(device=self.device, dtype=dt).contiguous()

        mean_d = dev(mean, torch.float64, (self.d_obs + 1, A) if closed else (H, A))
        noise_d = None if noise is None else dev(noise, self._tdtype, (P, H, A))
        costs = torch.empty((P, H), dtype=self._tdtype, device=self.device)
        act = torch.empty((P, H, A), dtype=self._tdtype, device=self.device)
        obs = torch.empty((P, H, self.d_obs), dtype=self._tdtype, device=self.device) if want_obs else None
        nobs = torch.empty((P, H, self.d_obs), dtype=self._tdtype, device=self.device) if want_obs else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.mjmpc_analytic_rollout(self.kind, _ptr(self._params), self.d_state, A, _ptr(self._state),
                                                    self._code, P, H, _ptr(mean_d), _ptr(noise_d), _ptr(costs), _ptr(act),
                                                    _ptr(obs), _ptr(nobs), int(closed), stream))
        return costs, act, obs, nobs

    def rollout(self, num_particles, horizon, mean, noise, mode="open_loop"):
        t0 = time.time()
        costs, act, obs, nobs = (x.cpu().numpy().astype(np.float64, copy=False)
                                 for x in self.rollout_device(num_particles, horizon, mean, noise, mode, want_obs=True))
        info = [{"total_time": time.time() - t0} for _ in range(self.num_shards)]
        return obs, -costs, act, np.zeros((num_particles, horizon)), info, nobs

    def reset(self):
        pass

    def close(self):
        self.closed = True

"""MuJoCo's reset on instability, seen from the host.

The kernels emulate ``mj_checkPos / mj_checkVel / mj_checkAcc -> mj_resetData`` [EXT]: a simulation whose qpos / qvel / qacc
holds a NaN or an entry beyond 1e10 goes on from qpos0.  Inside ``GymEnvWrapper.rollout`` (gym_env_wrapper.py:125-153) that
is all the reference does too.  For the env that is being CONTROLLED it is not: under mujoco-py's default warning callback
``sim.step()`` raises ``MujocoException`` in the worker that steps it (reacher_env.py:29-39 via gym_env_wrapper.py:64-66).
The engines therefore count the real env's resets apart (``mjmpc_*_env_resets``) and surface them:

* ``engine.on_env_reset`` = ``"raise"`` (default: ``SimulationUnstableError``, the stand-in for ``MujocoException``),
  ``"warn"`` or ``"ignore"``;
* host-synchronous env steps (``Reacher7DOFEnv.step`` ..., ``TreeRolloutEngine.step``) check right away;
* the device-resident env (``step_state``, the env step inside the fused control iteration) is checked wherever the host
  synchronises anyway - ``get_state_device()``, ``check_env_resets()`` - never inside the asynchronous launch path;
* ``engine.set_reset_returns("inf")``: rollout particles that reset cost +inf from that env step on (no weight in the
  updates) instead of the finite costs of the reset state.
"""
import contextlib
import ctypes
import warnings

from .. import _lib


class SimulationUnstableError(RuntimeError):
    """The controlled env's simulation left MuJoCo's bounds and was reset (the reference raises MujocoException here)."""


class EnvResetWatch:
    """Mixin of the rollout engines (``_abi`` = "arm" / "tree", ``_h`` the handle, ``_lib`` the library)."""
    on_env_reset = "raise"
    _env_resets_seen = 0
    _abi = "arm"

    def env_resets(self):
        """Resets of the device-resident real env since the engine was made (synchronises the device)."""
        c = ctypes.c_uint32()
        _lib.check(getattr(self._lib, "mjmpc_%s_env_resets" % self._abi)(self._h, ctypes.byref(c)))
        return int(c.value)

    def check_env_resets(self, where="the device-resident env"):
        """Raise / warn (``on_env_reset``) if the real env has reset since the last check; returns the new resets."""
        n = self.env_resets()
        new, self._env_resets_seen = n - self._env_resets_seen, n
        if new > 0:
            self._env_reset_event(new, where)
        return new

    def set_reset_returns(self, mode):
        """"finite" (default: the costs of MuJoCo's reset state, what the reference's workers return) or "inf" (+inf from the
        env step of the reset on).  Applies to launches issued afterwards (a captured iteration keeps what it was made with)."""
        if mode not in ("finite", "inf"):
            raise ValueError("reset_returns must be 'finite' or 'inf'")
        _lib.check(getattr(self._lib, "mjmpc_%s_set_reset_returns" % self._abi)(self._h, int(mode == "inf")))
        self.reset_returns = mode

    @contextlib.contextmanager
    def real_step_guard(self, where):
        """Around a host-synchronous step of the controlled env (a one-particle rollout): any reset in it is the real env's."""
        if self.on_env_reset == "ignore":
            yield
            return
        before = self.diverged_substeps()
        yield
        new = self.diverged_substeps() - before
        if new > 0:
            self._env_reset_event(new, where)

    def _env_reset_event(self, n, where):
        msg = ("%s: the simulation left MuJoCo's bounds (a NaN or an entry beyond 1e10 in qpos / qvel / qacc) and was reset "
               "to qpos0 with zero controls %d time(s) - the reference raises MujocoException here; set engine.on_env_reset = "
               "'warn' / 'ignore' to go on" % (where, n))
        if self.on_env_reset == "raise":
            raise SimulationUnstableError(msg)
        if self.on_env_reset == "warn":
            warnings.warn(msg)

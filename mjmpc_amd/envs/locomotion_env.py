"""``Swimmer-v0`` and ``HalfCheetah-v0`` as environment objects, stepped by the HIP tree engine at P = 1.

Mirror the reference env classes (mjmpc/envs/basic/swimmer.py, half_cheetah.py; registered in
mjmpc/envs/__init__.py:11-19): ``step`` (reward = forward progress of qpos[0] / dt + control cost, never done),
``_get_obs`` (qpos without its leading entries, then qvel), ``reset_model`` (initial state + noise from the env's own
generator), ``get_env_state`` / ``set_env_state`` (a ``{qpos, qvel}`` dictionary).  They play the "real" environment
of a closed-loop run when no MuJoCo is installed; the models are the reference's vendored XMLs restated in
mjmpc_amd/models/swimmer.py / half_cheetah.py (physics: MuJoCo's published algorithm, parity unpinned - DESIGN 4.6).
"""
import dataclasses

import numpy as np

from ..models.half_cheetah import half_cheetah_raw
from ..models.swimmer import swimmer_raw
from .tree_engine import TreeRolloutEngine


class _ForwardEnv:
    raw_fn = None
    fwd_key = "reward_fwd"

    def __init__(self, device=0, dtype="f64", engine=None):
        raw = type(self).raw_fn()
        # the stepping engine observes the whole state (obs_skip = 0); rewards do not depend on the observation
        self.engine = engine or TreeRolloutEngine(dataclasses.replace(raw, obs_skip=0), device=device, dtype=dtype)
        self.raw = raw
        self.nv = self.engine.model.nv
        self.obs_skip = raw.obs_skip
        self.frame_skip = raw.frame_skip
        self.dt = raw.timestep * raw.frame_skip
        self.ctrl_cost = raw.ctrl_cost
        self.d_obs, self.d_state, self.d_action = 2 * self.nv - self.obs_skip, 2 * self.nv, self.engine.d_action
        self.action_lows, self.action_highs = self.engine.action_lows, self.engine.action_highs
        self.np_random = np.random.RandomState(0)
        self.init_qpos, self.init_qvel = np.zeros(self.nv), np.zeros(self.nv)
        self._qpos, self._qvel = self.init_qpos.copy(), self.init_qvel.copy()
        self.real_step = True

    # -- gym-like surface ---------------------------------------------------------------------
    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def reset(self, seed=None):
        if seed is not None:
            self.seed(seed)
        return self.reset_model()

    def set_state(self, qpos, qvel):
        self._qpos, self._qvel = np.array(qpos, float).copy(), np.array(qvel, float).copy()

    def step(self, a):
        a = np.asarray(a, float).reshape(-1)
        xposbefore = self._qpos[0]
        self.engine.set_env_state(dict(qpos=self._qpos, qvel=self._qvel))
        with self.engine.real_step_guard("%s.step" % type(self).__name__):
            _, rew, _, _, _, nobs = self.engine.rollout(1, 1, a.reshape(1, -1), None)
        self._qpos, self._qvel = nobs[0, 0, :self.nv].copy(), nobs[0, 0, self.nv:].copy()
        reward_fwd = (self._qpos[0] - xposbefore) / self.dt
        reward_ctrl = -self.ctrl_cost * np.square(a).sum()
        return self._get_obs(), float(rew[0, 0]), False, {self.fwd_key: reward_fwd, "reward_ctrl": reward_ctrl,
                                                          "goal_achieved": False}

    def _get_obs(self):
        return np.concatenate([self._qpos[self.obs_skip:], self._qvel])

    def get_obs(self):
        return self._get_obs()

    def real_env_step(self, flag):
        self.real_step = bool(flag)

    def get_env_state(self):
        return dict(qpos=self._qpos.copy(), qvel=self._qvel.copy())

    def set_env_state(self, state_dict):
        self.set_state(state_dict["qpos"], state_dict["qvel"])

    def evaluate_success(self, paths):
        """The reference's locomotion envs define no success metric; the driver prints the mean forward progress
        per step instead (m/s over the episode)."""
        return float(np.mean([np.mean(p["rewards"]) for p in paths]))


class SwimmerEnv(_ForwardEnv):
    """mjmpc/envs/basic/swimmer.py: frame_skip 4, ctrl cost 1e-4, obs = [qpos[2:], qvel]."""
    raw_fn = staticmethod(swimmer_raw)
    fwd_key = "reward_fwd"
    max_episode_steps = 1000

    def reset_model(self):
        self.set_state(self.init_qpos + self.np_random.uniform(low=-.1, high=.1, size=self.nv),
                       self.init_qvel + self.np_random.uniform(low=-.1, high=.1, size=self.nv))     # swimmer.py:26-31
        return self._get_obs()


class HalfCheetahEnv(_ForwardEnv):
    """mjmpc/envs/basic/half_cheetah.py: frame_skip 5, ctrl cost 0.1, obs = [qpos[1:], qvel]."""
    raw_fn = staticmethod(half_cheetah_raw)
    fwd_key = "reward_run"
    max_episode_steps = 1000

    def reset_model(self):
        qpos = self.init_qpos + self.np_random.uniform(low=-.1, high=.1, size=self.nv)              # half_cheetah.py:27-31
        qvel = self.init_qvel + self.np_random.randn(self.nv) * .1
        self.set_state(qpos, qvel)
        return self._get_obs()

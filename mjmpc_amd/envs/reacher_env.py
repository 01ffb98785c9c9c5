"""``reacher_7dof-v0`` as an environment object, stepped by the same HIP engine at P = 1.

Mirrors the reference env's state interface (mjmpc/envs/basic/reacher_env.py): ``step`` (:29-39),
``get_obs`` (:41-47), ``reset_model`` / ``target_reset`` (:56-70), ``get_env_state`` /
``set_env_state`` (:81-99), ``evaluate_success`` (:117-125).  Used as the "real" environment of a
closed-loop run (examples/example_mpc.py:165-168) when no MuJoCo is installed.
"""
import numpy as np

from .arm_engine import ArmRolloutEngine
from ..models.reacher7dof import reacher7dof_raw


class Reacher7DOFEnv:
    max_episode_steps = 75                       # mjmpc/envs/__init__.py:24

    def __init__(self, device=0, dtype="f64", engine=None):
        self.engine = engine or ArmRolloutEngine(reacher7dof_raw(), device=device, dtype=dtype)
        self.nv = self.engine.model.nv
        self.d_obs, self.d_state, self.d_action = self.engine.d_obs, self.engine.d_state, self.engine.d_action
        self.action_lows, self.action_highs = self.engine.action_lows, self.engine.action_highs
        self.np_random = np.random.RandomState(0)
        self.env_timestep = 0
        self.real_step = True
        self._qp, self._qv = np.zeros(self.nv), np.zeros(self.nv)
        self._qa = np.zeros(self.nv)
        self._target = self.engine.model.target_default.copy()
        self._hand = self._fresh_hand()

    # -- internals --------------------------------------------------------------------------
    def _push(self):
        self.engine.set_env_state(dict(qp=self._qp, qv=self._qv, target_pos=self._target))

    def _fresh_hand(self):
        """site_xpos right after set_env_state's sim.forward(): read from a zero-length-lag rollout
        (the pre-step observation of a 1-step rollout is computed from the current qpos)."""
        self._push()
        obs, _, _, _, _, _ = self.engine.rollout(1, 1, np.zeros((1, self.d_action)), None)
        return obs[0, 0, 2 * self.nv:2 * self.nv + 3].copy()

    # -- gym-like surface ---------------------------------------------------------------------
    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def reset(self, seed=None):
        if seed is not None:
            self.seed(seed)
        self._qp, self._qv = np.zeros(self.nv), np.zeros(self.nv)
        self.target_reset()
        self.env_timestep = 0
        self._hand = self._fresh_hand()
        return self.get_obs()

    def step(self, a):
        self._push()
        with self.engine.real_step_guard("%s.step" % type(self).__name__):
            obs, rew, act, done, info, nobs = self.engine.rollout(1, 1, np.asarray(a, float).reshape(1, -1), None)
        o = nobs[0, 0]
        self._qp, self._qv = o[:self.nv].copy(), o[self.nv:2 * self.nv].copy()
        self._hand = o[2 * self.nv:2 * self.nv + 3].copy()
        ob = self.get_obs()
        self.env_timestep += 1                      # reacher_env.py:37-38: the count, then the timed events
        self.trigger_timed_events()
        return ob, float(rew[0, 0]), False, self.get_env_infos()

    def target_reset(self):
        """reacher_env.py:56-62: a new target from the env's own generator."""
        t = np.array([0.1, 0.1, 0.1])
        t[0] = self.np_random.uniform(low=-0.3, high=0.3)
        t[1] = self.np_random.uniform(low=-0.2, high=0.2)
        t[2] = self.np_random.uniform(low=-0.25, high=0.25)
        self._target = t

    def trigger_timed_events(self):
        pass                                        # used by the continual version (reacher_env.py:73-75)

    def real_env_step(self, flag):
        self.real_step = bool(flag)

    def get_obs(self):
        return np.concatenate([self._qp, self._qv, self._hand, self._hand - self._target])

    def get_env_state(self):
        return dict(qp=self._qp.copy(), qv=self._qv.copy(), qa=self._qa.copy(), target_pos=self._target.copy(),
                    timestep=self.env_timestep)

    def set_env_state(self, state):
        self._qp = np.array(state['qp'], float).copy()
        self._qv = np.array(state['qv'], float).copy()
        self._qa = np.array(state.get('qa', np.zeros(self.nv)), float).copy()
        self._target = np.array(state['target_pos'], float).copy()
        self.env_timestep = state.get('timestep', 0)
        self._hand = self._fresh_hand()

    def get_env_infos(self):
        l2 = np.linalg.norm(self._hand - self._target)
        return dict(state=self.get_env_state(), goal_achieved=(l2 < 0.025))

    def evaluate_success(self, paths):
        n = sum(1 for p in paths if np.sum(p['env_infos']['goal_achieved']) > 10)
        return n * 100.0 / len(paths)


class ContinualReacher7DOFEnv(Reacher7DOFEnv):
    """``continual_reacher-v0`` (mjmpc/envs/__init__.py:27-31, reacher_env.py:128-132): the real environment
    draws a new target every 50 steps; rollouts (real_step False) never do."""
    max_episode_steps = 250

    def trigger_timed_events(self):
        if self.env_timestep % 50 == 0 and self.env_timestep > 0 and self.real_step is True:
            self.target_reset()


class HandTreeEnv(Reacher7DOFEnv):
    """``hand_tree-v0``: the synthetic 24-dof hand-on-an-arm tree (mjmpc_amd/models/hand24.py) with the reacher task -
    bring the index fingertip to a target above the table - stepped by the TREE engine at P = 1.  Not a reference
    environment: it gives the tree kernel (pen-v0's shape of work) a closed loop to run in."""
    max_episode_steps = 100

    def __init__(self, device=0, dtype="f64", engine=None):
        if engine is None:
            from .tree_engine import TreeRolloutEngine
            from ..models.hand24 import hand24_raw
            engine = TreeRolloutEngine(hand24_raw(), device=device, dtype=dtype)
        super().__init__(device=device, dtype=dtype, engine=engine)

    def target_reset(self):
        t = np.zeros(3)                              # a box in front of the hand's rest pose, above the table
        t[0] = self.np_random.uniform(low=0.45, high=0.65)
        t[1] = self.np_random.uniform(low=-0.55, high=-0.25)
        t[2] = self.np_random.uniform(low=-0.05, high=0.15)
        self._target = t

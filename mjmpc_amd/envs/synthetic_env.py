"""Environment objects over the round-4 synthetic MJCF models (mjmpc_amd/models/synthetic.py: cart-pole with friction
loss, a free-jointed glass on a tray carried by an arm, a door with a latch) - the KINDS of environment the reference's
experiment files name beyond its vendored three (examples/configs/classic_control/cartpole*.yml, panda/tray_glass-v0.yml,
sawyer/door-v0.yml), whose own assets are absent.  Same state interface as ``Reacher7DOFEnv`` (reacher_env.py:29-99):
``{qp, qv, qa, target_pos, timestep}`` with ``qp`` in MuJoCo's qpos layout (a free joint: position + quaternion), the
reacher task's reward and observation on the model's tracked site, stepped by the TREE engine at P = 1."""
import numpy as np

from .reacher_env import Reacher7DOFEnv
from ..models.synthetic import start_state, synthetic_raw


class SyntheticEnv(Reacher7DOFEnv):
    model_name = None
    max_episode_steps = 100
    goal_radius = 0.05
    reset_noise = 0.0

    def __init__(self, device=0, dtype="f64", engine=None, num_shards=1):
        if engine is None:
            from .tree_engine import TreeRolloutEngine
            engine = TreeRolloutEngine(synthetic_raw(self.model_name), device=device, dtype=dtype, num_shards=num_shards)
        self.engine = engine
        self.nv, self.nq = engine.model.nv, engine.model.nq
        self.d_obs, self.d_state, self.d_action = engine.d_obs, engine.d_state, engine.d_action
        self.action_lows, self.action_highs = engine.action_lows, engine.action_highs
        self.np_random = np.random.RandomState(0)
        self.env_timestep = 0
        self.real_step = True
        st = start_state(self.model_name, engine.raw)
        self._start = st
        self._qp, self._qv, self._qa = st["qp"].copy(), st["qv"].copy(), np.zeros(self.nv)
        self._target = st["target_pos"].copy()
        self._hand = self._fresh_hand()

    def _fresh_hand(self):
        self._push()
        obs, _, _, _, _, _ = self.engine.rollout(1, 1, np.zeros((1, self.d_action)), None)
        return obs[0, 0, self.nq + self.nv:self.nq + self.nv + 3].copy()

    def reset(self, seed=None):
        if seed is not None:
            self.seed(seed)
        self._qp, self._qv = self._start["qp"].copy(), self._start["qv"].copy()
        if self.reset_noise > 0 and self.nq == self.nv:         # (hinge / slide models; quaternions are left alone)
            self._qp = self._qp + self.np_random.uniform(-self.reset_noise, self.reset_noise, self.nq)
        self.target_reset()
        self.env_timestep = 0
        self._hand = self._fresh_hand()
        return self.get_obs()

    def step(self, a):
        self._push()
        with self.engine.real_step_guard("%s.step" % type(self).__name__):
            obs, rew, act, done, info, nobs = self.engine.rollout(1, 1, np.asarray(a, float).reshape(1, -1), None)
        o = nobs[0, 0]
        self._qp, self._qv = o[:self.nq].copy(), o[self.nq:self.nq + self.nv].copy()
        self._hand = o[self.nq + self.nv:self.nq + self.nv + 3].copy()
        ob = self.get_obs()
        self.env_timestep += 1
        return ob, float(rew[0, 0]), False, self.get_env_infos()

    def target_reset(self):
        self._target = self._start["target_pos"].copy()

    def get_env_infos(self):
        l2 = np.linalg.norm(self._hand - self._target)
        return dict(state=self.get_env_state(), goal_achieved=(l2 < self.goal_radius))


class CartPoleEnv(SyntheticEnv):
    """Swing the pole up: the tracked site is the pole's tip, the target the point above the rail's centre."""
    model_name = "cartpole"
    max_episode_steps = 200
    goal_radius = 0.15
    reset_noise = 0.05


class TrayEnv(SyntheticEnv):
    """Carry the glass (a free-jointed body standing on the tray) to the target without dropping it."""
    model_name = "tray"
    max_episode_steps = 100
    goal_radius = 0.05


class DoorEnv(SyntheticEnv):
    """Unlatch the door (turn the handle: a joint equality retracts the bolt) and swing it open: the tracked site is
    the handle, the target its place with the door open."""
    model_name = "door"
    max_episode_steps = 150
    goal_radius = 0.1


class GripperEnv(SyntheticEnv):
    """Lift the pen (a free capsule lying across the gripper's two box fingers) to the target above it without dropping it."""
    model_name = "gripper"
    max_episode_steps = 100
    goal_radius = 0.03

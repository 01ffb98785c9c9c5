"""Random-shooting MPC on the GPU (reference mjmpc/control/random_shooting.py)."""
import numpy as np

from .controller import OLGaussianMPC


class RandomShooting(OLGaussianMPC):
    def __init__(self, d_state, d_obs, d_action, horizon, init_cov, base_action, num_particles, step_size, gamma,
                 n_iters, action_lows, action_highs, set_sim_state_fn=None, rollout_fn=None, sample_mode='mean',
                 filter_coeffs=[1.0, 0.0, 0.0], batch_size=1, seed=0, **device_kw):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov,
                         np.zeros(shape=(horizon, d_action)), base_action, num_particles, gamma, n_iters, step_size,
                         filter_coeffs, set_sim_state_fn, rollout_fn, 'diagonal', sample_mode, batch_size, seed,
                         **device_kw)

    def _static_cov(self):
        return True

    def _wants_q0(self):
        return True

    def _device_update(self, trajectories):
        self.dev.rs_update(trajectories["costs"], trajectories["actions"], self.step_size, q0=trajectories.get("q0"))

    def _update_distribution(self, trajectories):
        """random_shooting.py:52-62: move the mean towards the single best action sequence."""
        self._sync_in()
        self.dev.rs_update(trajectories["costs"], trajectories["actions"], self.step_size)
        self._pull()

    def _calc_val(self, trajectories):
        return self.dev.mean_q0(trajectories["costs"])

"""MPC over closed-loop linear-Gaussian policies (reference mjmpc/control/clgaussian_mpc.py).

The sampling distribution is a policy ``a = W^T [obs; 1] + eps``: ``mean_weights`` is the (d_obs+1, d_action)
matrix the engine's ``mode="closed_loop_linear"`` rollouts take (gym_env_wrapper.py:135-136); the rollouts
return ``observations``, which concrete subclasses (policy-gradient updates such as the reference's
``Reinforce``) consume.  This base class leaves ``_update_distribution`` abstract, like the reference.
"""
import copy

import numpy as np

from .control_utils import generate_noise
from .controller import Controller


class CLGaussianMPC(Controller):
    def __init__(self, d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov, init_mean,
                 num_particles, gamma, n_iters, filter_coeffs, set_sim_state_fn=None, rollout_fn=None,
                 cov_type='diagonal', sample_mode='mean', batch_size=1, seed=0, device=0, comm=None):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, gamma, n_iters,
                         set_sim_state_fn, rollout_fn, sample_mode, batch_size, seed, device=device, comm=comm)
        self.init_cov = np.array([init_cov] * self.d_action)
        self.init_mean = init_mean.copy()
        self.mean_weights = init_mean
        self.num_particles = num_particles
        self.cov_type = cov_type
        self.cov_action = np.diag(self.init_cov)
        self.filter_coeffs = filter_coeffs
        self.curr_obs = None

    def _get_next_action(self, state, mode='mean'):
        """clgaussian_mpc.py:63-73: the policy at the observation the last rollouts started from."""
        mean_action = self.mean_weights.T @ np.append(self.curr_obs, 1.0)
        if mode == 'mean':
            return mean_action.copy()
        if mode == 'sample':
            delta = generate_noise(self.cov_action, self.filter_coeffs, shape=(1, 1),
                                   base_seed=self.seed_val + 123 * self.num_steps)
            return mean_action.copy() + delta.reshape(self.d_action).copy()
        raise ValueError('Unidentified sampling mode in get_next_action')

    def sample_noise(self):
        """clgaussian_mpc.py:84-89."""
        return generate_noise(self.cov_action, self.filter_coeffs, shape=(self.num_particles, self.horizon),
                              base_seed=self.seed_val + self.num_steps)

    def generate_rollouts(self, state):
        """clgaussian_mpc.py:91-116: closed-loop rollouts under the current weights; remembers the start obs."""
        self._set_sim_state_fn(copy.deepcopy(state))
        delta = self.sample_noise()
        trajectories = self._rollout_fn(self.num_particles, self.horizon, self.mean_weights, delta,
                                        mode="closed_loop_linear")
        first = trajectories["observations"][0, 0]
        self.curr_obs = first.cpu().numpy().astype(np.float64) if hasattr(first, "cpu") else np.array(first)
        return trajectories

    def _shift(self):
        pass                    # a state-feedback policy needs no time shift (clgaussian_mpc.py:118-132)

    def reset(self, seed=None):
        if seed is not None:
            self.seed_val = self.seed(seed)
        self.num_steps = 0
        self.mean_weights = self.init_mean.copy()
        self.cov_action = np.diag(self.init_cov)
        self.gamma_seq = np.cumprod([1.0] + [self.gamma] * (self.horizon - 1)).reshape(1, self.horizon)
        self.converged = False

    def _calc_val(self, cost_seq, act_seq):
        raise NotImplementedError("_calc_val not implemented")

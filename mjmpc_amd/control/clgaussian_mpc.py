"""MPC over closed-loop linear-Gaussian policies (reference mjmpc/control/clgaussian_mpc.py).

The sampling distribution is a state-feedback policy ``a = W^T [obs; 1] + eps`` instead of an open-loop action
sequence: ``mean_weights`` is the (d_obs+1, d_action) matrix that the engines' ``mode="closed_loop_linear"``
rollouts take (gym_env_wrapper.py:135-136).  Those rollouts return ``observations``, which a concrete subclass
(a policy-gradient update such as the reference's ``Reinforce``) consumes; like the reference, this base class
leaves ``_update_distribution`` to the subclass.
"""
import copy

import numpy as np

from .control_utils import generate_noise
from .controller import Controller


class CLGaussianMPC(Controller):
    # positional order as in the reference constructor (clgaussian_mpc.py:11-29)
    def __init__(self, d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov, init_mean,
                 num_particles, gamma, n_iters, filter_coeffs, set_sim_state_fn=None, rollout_fn=None,
                 cov_type='diagonal', sample_mode='mean', batch_size=1, seed=0, device=0, comm=None):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, gamma, n_iters,
                         set_sim_state_fn, rollout_fn, sample_mode, batch_size, seed, device=device, comm=comm)
        self.num_particles, self.cov_type, self.filter_coeffs = num_particles, cov_type, filter_coeffs
        self.init_mean, self.mean_weights = init_mean.copy(), init_mean
        self.init_cov = np.full(self.d_action, init_cov, dtype=float)
        self.cov_action = np.diag(self.init_cov)
        self.curr_obs = None                    # observation the last rollouts started from

    # -- the policy ---------------------------------------------------------------------------------------
    def _policy_mean(self, obs):
        return self.mean_weights.T @ np.append(obs, 1.0)

    def _get_next_action(self, state, mode='mean'):
        """The policy evaluated at ``curr_obs`` (clgaussian_mpc.py:63-73); 'sample' adds one seeded draw."""
        action = self._policy_mean(self.curr_obs)
        if mode == 'sample':
            eps = generate_noise(self.cov_action, self.filter_coeffs, shape=(1, 1),
                                 base_seed=self.seed_val + 123 * self.num_steps)
            action = action + eps.reshape(self.d_action)
        elif mode != 'mean':
            raise ValueError('Unidentified sampling mode in get_next_action')
        return action.copy()

    # -- rollouts -------------------------------------------------------------------------------------------
    def sample_noise(self):
        """(P, H, A) action perturbations from the reference's stream, seed + num_steps (clgaussian_mpc.py:84-89)."""
        return generate_noise(self.cov_action, self.filter_coeffs, shape=(self.num_particles, self.horizon),
                              base_seed=self.seed_val + self.num_steps)

    def generate_rollouts(self, state):
        """Closed-loop rollouts under the current weights (clgaussian_mpc.py:91-116)."""
        self._set_sim_state_fn(copy.deepcopy(state))
        trajectories = self._rollout_fn(self.num_particles, self.horizon, self.mean_weights, self.sample_noise(),
                                        mode="closed_loop_linear")
        first = trajectories["observations"][0, 0]
        self.curr_obs = first.cpu().numpy().astype(np.float64) if hasattr(first, "cpu") else np.array(first)
        return trajectories

    # -- bookkeeping ------------------------------------------------------------------------------------------
    def _shift(self):
        """A state-feedback policy is not indexed by time: nothing to shift (clgaussian_mpc.py:118-132)."""

    def reset(self, seed=None):
        if seed is not None:
            self.seed_val = self.seed(seed)
        self.num_steps, self.converged = 0, False
        self.mean_weights = self.init_mean.copy()
        self.cov_action = np.diag(self.init_cov)
        self.gamma_seq = np.cumprod([1.0] + [self.gamma] * (self.horizon - 1)).reshape(1, self.horizon)

    def _calc_val(self, cost_seq, act_seq):
        raise NotImplementedError("_calc_val not implemented")

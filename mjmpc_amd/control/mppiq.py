"""MPPI over Q-function estimates, update on the GPU (reference mjmpc/control/mppiq.py).

The trajectory score is a TD(lambda) blend of the per-step costs and optional terminal Q estimates
(``trajectories["qvals"]``, shape (P,H)); weights are a softmax over particles, per horizon step by default.
Both stages are HIP kernels: ``mjmpc_td_lambda_returns`` then the softmax record / combine MPPI uses.
"""
import numpy as np

from .controller import OLGaussianMPC


class MPPIQ(OLGaussianMPC):
    """Same constructor as the reference (mppiq.py:20-70): ``beta`` is the temperature, ``td_lam`` the TD mix."""

    def __init__(self, d_state, d_obs, d_action, horizon, init_cov, base_action, beta, num_particles, step_size, alpha,
                 gamma, n_iters, td_lam, action_lows, action_highs, time_based_weights=True, set_sim_state_fn=None,
                 get_sim_state_fn=None, sim_step_fn=None, sim_reset_fn=None, rollout_fn=None, sample_mode='mean',
                 batch_size=1, filter_coeffs=[1., 0., 0.], seed=0, **device_kw):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov,
                         np.zeros(shape=(horizon, d_action)), base_action, num_particles, gamma, n_iters, step_size,
                         filter_coeffs, set_sim_state_fn, rollout_fn, 'diagonal', sample_mode, batch_size, seed,
                         **device_kw)
        self.beta = beta
        self.td_lam = td_lam
        self.alpha = alpha
        self.time_based_weights = time_based_weights

    def _static_cov(self):
        return self.alpha == 1

    def _host_uploads_per_step(self):
        return True             # td_lambda_returns uploads its weight table (and cov^-1) from the host on every update:
                                # not capturable - enable_graph() refuses instead of failing inside the capture

    def _returns(self, trajectories):
        """mppiq.py:91-126: per-step total cost -> TD(lambda) returns, on the device."""
        qvals = trajectories.get("qvals") if hasattr(trajectories, "get") else None
        covinv = np.linalg.inv(self.cov_action) if self.alpha != 1 else None
        return self.dev.td_lambda_returns(trajectories["costs"], trajectories["actions"], qvals, self.beta,
                                          0 if self.alpha != 1 else 1, self.gamma, self.td_lam, covinv)

    def _device_update(self, trajectories):
        self.dev.softmax_update(self._returns(trajectories), trajectories["actions"], self.beta, self.step_size,
                                time_based_weights=self.time_based_weights, costs_are_returns=True)

    def _update_distribution(self, trajectories):
        """mppiq.py:73-102: w = softmax(-q_hat / beta) over particles; mean <- (1-step) mean + step * sum w a."""
        self._sync_in()
        self._device_update(trajectories)
        self._pull()

    def _calc_val(self, trajectories):
        """mppiq.py:138-165: -beta * logsumexp(-q_hat[:,0] / beta, b = 1/P)."""
        self._sync_in()
        self.dev.softmax_update(self._returns(trajectories), trajectories["actions"], self.beta, 0.0,
                                costs_are_returns=True, want_value=True, update_mean=False)
        return float(self.dev.value.item())

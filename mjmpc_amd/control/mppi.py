"""MPPI with the update on the GPU (reference mjmpc/control/mppi.py)."""
import numpy as np

from .controller import OLGaussianMPC


class MPPI(OLGaussianMPC):
    """Same constructor as the reference (mppi.py:16-66); ``alpha`` = 1 switches the control cost off."""

    def __init__(self, d_state, d_obs, d_action, horizon, init_cov, base_action, lam, num_particles, step_size,
                 alpha, gamma, n_iters, action_lows, action_highs, time_based_weights=False, set_sim_state_fn=None,
                 get_sim_state_fn=None, sim_step_fn=None, sim_reset_fn=None, rollout_fn=None, sample_mode='mean',
                 batch_size=1, filter_coeffs=[1., 0., 0.], seed=0, use_zero_control_seq=False, **device_kw):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov,
                         np.zeros(shape=(horizon, d_action)), base_action, num_particles, gamma, n_iters, step_size,
                         filter_coeffs, set_sim_state_fn, rollout_fn, 'diagonal', sample_mode, batch_size, seed,
                         use_zero_control_seq, **device_kw)
        self.lam = lam
        self.alpha = alpha
        self.time_based_weights = time_based_weights

    def _covinv(self):
        return np.linalg.inv(self.cov_action) if self.alpha != 1 else None

    def _static_cov(self):
        return self.alpha == 1          # alpha == 0 uploads cov^-1 every update (host work)

    def _fused_capable(self):
        return (self.alpha == 1 and not self.time_based_weights and not self.use_zero_control_seq
                and not self.dev.gamma_zero and hasattr(self._rollout_fn, "fused"))

    def _device_update(self, trajectories):
        self.dev.softmax_update(trajectories["costs"], trajectories["actions"], self.lam, self.step_size,
                                time_based_weights=self.time_based_weights)

    def _update_distribution(self, trajectories):
        """mppi.py:69-111: w = softmax(-(cost-to-go + lam * control cost)/lam) over particles (per
        horizon step when time_based_weights); mean <- (1-step) mean + step * sum_p w_p actions_p."""
        self._sync_in()
        self.dev.softmax_update(trajectories["costs"], trajectories["actions"], self.lam, self.step_size,
                                alpha=0 if self.alpha != 1 else 1, time_based_weights=self.time_based_weights,
                                covinv=self._covinv())
        self._pull()

    def _calc_val(self, trajectories):
        """mppi.py:113-131: -lam * logsumexp(-total/lam, b = 1/P)."""
        if self.time_based_weights:
            raise ValueError("MPPI._calc_val is undefined with time_based_weights (the reference raises too)")
        self._sync_in()
        self.dev.softmax_update(trajectories["costs"], trajectories["actions"], self.lam, 0.0,
                                alpha=0 if self.alpha != 1 else 1, covinv=self._covinv(), want_value=True,
                                update_mean=False)
        return float(self.dev.value.item())

"""DMD-MPC (Gaussian, exponential utility, optional covariance adaptation) on the GPU
(reference mjmpc/control/gaussian_dmd.py)."""
import numpy as np

from .controller import OLGaussianMPC


class DMDMPC(OLGaussianMPC):
    def __init__(self, d_state, d_obs, d_action, horizon, init_cov, beta, base_action, lam, num_particles,
                 step_size, gamma, n_iters, action_lows, action_highs, set_sim_state_fn=None, rollout_fn=None,
                 update_cov=False, cov_type='diagonal', sample_mode='mean', batch_size=1,
                 filter_coeffs=[1., 0., 0.], seed=0, **device_kw):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov,
                         np.zeros(shape=(horizon, d_action)), base_action, num_particles, gamma, n_iters, step_size,
                         filter_coeffs, set_sim_state_fn, rollout_fn, cov_type, sample_mode, batch_size, seed,
                         **device_kw)
        self.lam = lam
        self.beta = beta
        self.update_cov = update_cov

    def _static_cov(self):
        return not self.update_cov

    def _fused_capable(self):
        # without covariance adaptation the update is MPPI's with alpha = 1 (gaussian_dmd.py:65-104): same fused
        # rollout (filter + cost-to-go) and two-launch update
        return (not self.update_cov and not self.dev.gamma_zero and hasattr(self._rollout_fn, "fused"))

    def _device_cov(self):
        return self.update_cov and self.cov_type in ('diagonal', 'full')

    def _cov_mode(self):
        if not self.update_cov:
            return 0
        if self.cov_type == 'diagonal':
            return 1
        if self.cov_type == 'full':
            return 2
        raise ValueError('Unidentified covariance type in update_distribution')

    def _device_update(self, trajectories):
        self.dev.softmax_update(trajectories["costs"], trajectories["actions"], self.lam, self.step_size,
                                cov_mode=self._cov_mode())

    def _shift_cov_args(self):
        return (None, self.beta) if self.update_cov else None

    def _update_distribution(self, trajectories):
        """gaussian_dmd.py:65-104: softmax weights; weighted mean; if update_cov the weighted scatter,
        diagonal = mean_t sum_p w delta^2, full = (sum_{p,t} w delta delta^T) / H."""
        self._sync_in()
        cov_mode = self._cov_mode()
        self.dev.softmax_update(trajectories["costs"], trajectories["actions"], self.lam, self.step_size,
                                cov_mode=cov_mode)
        self._pull(cov=bool(cov_mode))

    def _shift(self):
        """gaussian_dmd.py:106-113: shift the mean; grow the covariance by beta * I when it adapts."""
        super()._shift()
        if self.update_cov:
            self._sync_in()
            self._device_shift_cov()    # cov += beta * I on the device
            self._pull(cov=True)

    def _calc_val(self, trajectories):
        """gaussian_dmd.py:126-139."""
        self._sync_in()
        self.dev.softmax_update(trajectories["costs"], trajectories["actions"], self.lam, 0.0, want_value=True,
                                update_mean=False)
        return float(self.dev.value.item())

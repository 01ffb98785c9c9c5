"""Cross-entropy-method MPC on the GPU (reference mjmpc/control/cem.py)."""
import numpy as np

from .controller import OLGaussianMPC


class CEM(OLGaussianMPC):
    def __init__(self, d_state, d_obs, d_action, horizon, init_cov, base_action, elite_frac, num_particles,
                 step_size, gamma, n_iters, action_lows, action_highs, set_sim_state_fn=None, rollout_fn=None,
                 beta=0.0, cov_type='diagonal', sample_mode='mean', batch_size=1, filter_coeffs=[1., 0., 0.],
                 seed=0, **device_kw):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, init_cov,
                         np.zeros(shape=(horizon, d_action)), base_action, num_particles, gamma, n_iters, step_size,
                         filter_coeffs, set_sim_state_fn, rollout_fn, cov_type, sample_mode, batch_size, seed,
                         **device_kw)
        self.elite_frac = elite_frac
        self.beta = beta
        self.num_elite = int(self.num_particles * self.elite_frac)

    def _device_cov(self):
        return self.cov_type in ('diagonal', 'full')

    def _wants_q0(self):
        return True

    def _cem_fused(self):
        """The two-launch CEM step (``DeviceUpdater.cem_fused_step``): a captured / device-resident iteration with the
        Philox sampler, one iteration per step, a rollout launch that emits q0 and filters raw samples."""
        return (self._graph_on and getattr(self, "_want_cem_fused", True) and self.noise_mode == 'device' and self.n_iters == 1
                and hasattr(self._rollout_fn, "fused") and not self.dev.gamma_zero and not self.use_zero_control_seq
                and self.cov_type in ('diagonal', 'full') and self.num_elite >= 1
                and self.dev.cem_fused_supported(self.local_particles, self.num_elite))

    def _device_update(self, trajectories):
        self.dev.cem_update(trajectories["costs"], trajectories["actions"], self.num_elite, self.step_size,
                            self.cov_type == 'full', q0=trajectories.get("q0"))

    def _shift_cov_args(self):
        return self.init_cov, self.beta

    def _update_distribution(self, trajectories):
        """cem.py:65-86: the num_elite particles of least cost-to-go (ties by particle index) refit
        mean and covariance - np.var (ddof 0) on the diagonal, np.cov (ddof 1) for 'full'."""
        if self.cov_type not in ('diagonal', 'full'):
            raise ValueError("cov_type must be 'diagonal' or 'full'")
        self._sync_in()
        self.dev.cem_update(trajectories["costs"], trajectories["actions"], self.num_elite, self.step_size,
                            self.cov_type == 'full')
        self._pull(cov=True)

    def _shift(self):
        """cem.py:89-95."""
        super()._shift()
        self._sync_in()
        self._device_shift_cov()        # cov += beta * diag(init_cov), where the covariance lives
        self._pull(cov=True)

    def _calc_val(self, trajectories):
        """cem.py:107-112."""
        return self.dev.mean_q0(trajectories["costs"])

"""Particle-filter MPC (reference mjmpc/control/particle_filter_controller.py).

The rollout and the exponentiated-cost weights run on the GPU; the systematic resampling walk
(:159-174) is a serial prefix scan whose float summation order decides which particle survives,
so it stays on the host in the reference's order ("replicas + host gather").
"""
import copy
import random

import numpy as np

from .controller import Controller
from .control_utils import generate_noise


class PFMPC(Controller):
    def __init__(self, d_state, d_obs, d_action, horizon, cov_shift, cov_resample, base_action, lam,
                 num_particles, gamma, n_iters, action_lows, action_highs, set_sim_state_fn=None, rollout_fn=None,
                 sample_mode="mean", batch_size=1, filter_coeffs=[1., 0., 0.], seed=0, device=0, comm=None):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, gamma, n_iters,
                         set_sim_state_fn, rollout_fn, sample_mode, batch_size, seed, device=device, comm=comm)
        if self.dev.comm.world_size != 1:
            raise NotImplementedError("PFMPC runs as per-GPU replicas; particle sharding is not supported")
        self.lam = lam
        self.num_particles = num_particles
        self.cov_shift = np.diag(np.array([cov_shift] * self.d_action))
        self.cov_resample = np.diag(np.array([cov_resample] * self.d_action))
        self.base_action = base_action
        self.filter_coeffs = filter_coeffs
        random.seed(self.seed_val)
        self.mean_action = np.zeros(shape=(horizon, d_action))
        self.action_samples = generate_noise(self.cov_resample, self.filter_coeffs,
                                             shape=(self.num_particles, self.horizon), base_seed=self.seed_val)

    def generate_rollouts(self, state):
        self._set_sim_state_fn(copy.deepcopy(state))
        delta = self.action_samples - self.mean_action
        return self._rollout_fn(self.num_particles, self.horizon, self.mean_action, delta, mode="open_loop")

    def _exp_util(self, costs):
        """particle_filter_controller.py:104-113, evaluated by the softmax kernels."""
        costs = self.dev.to_device(costs, "costs")
        P = self.dev.softmax_update(costs, self.dev.zero_actions(costs.shape[0], costs), self.lam, 0.0,
                                    update_mean=False)
        return self.dev.softmax_weights(P).cpu().numpy()

    def _update_distribution(self, trajectories):
        w = self._exp_util(trajectories["costs"])
        random.seed(self.seed_val + self.num_steps)
        np.random.seed(self.seed_val + self.num_steps)
        self.action_samples = self._resampling(self.action_samples, w, low_variance=True)
        self.mean_action = np.mean(self.action_samples, axis=0)

    def sample_actions(self):
        return self.action_samples

    def _get_next_action(self, state, mode='mean'):
        return np.mean(self.action_samples, axis=0)[0].copy()

    def _shift(self):
        """particle_filter_controller.py:127-150: roll the samples, add fresh filtered noise, append."""
        self.action_samples[:, :-1] = self.action_samples[:, 1:]
        delta = generate_noise(self.cov_shift, self.filter_coeffs, shape=(self.num_particles, self.horizon),
                               base_seed=self.seed_val + self.num_steps)
        self.action_samples = self.action_samples + delta
        if self.base_action == 'random':
            self.action_samples[:, -1] = np.random.normal(0, self.cov_resample, self.d_action)
        elif self.base_action == 'null':
            self.action_samples[:, -1] = np.zeros((self.num_particles, self.d_action))
        elif self.base_action == 'repeat':
            self.action_samples[:, -1] = self.action_samples[:, -2]
        else:
            raise NotImplementedError("invalid option for base action during shift")

    def reset(self):
        self.num_steps = 0
        self.mean_action = np.zeros(shape=(self.horizon, self.d_action))
        self.action_samples = generate_noise(self.cov_resample, self.filter_coeffs,
                                             shape=(self.num_particles, self.horizon), base_seed=self.seed_val)

    def _resampling(self, act_seq, weights, low_variance=True):
        if not low_variance:
            return np.array(random.choices(self.action_samples, weights=weights, k=self.num_particles))
        M = act_seq.shape[0]
        out = np.zeros_like(act_seq)
        r = random.uniform(0.0, 1.0 / M * 1.0)
        c, i = 0.0, 0
        for m in range(M):
            u = r + m * 1.0 / M * 1.0
            while c < u and i < M:
                c += weights[i]
                i += 1
            out[m] = act_seq[i - 1]
        return out

    def _calc_val(self, trajectories):
        raise NotImplementedError("_calc val not implemented yet")

"""Particle-filter MPC with the rollout and the weights on the GPU.

Counterpart of the reference's ``PFMPC`` (mjmpc/control/particle_filter_controller.py): the control
distribution is a set of P action sequences; every iteration rolls them out, turns the discounted
costs into softmax weights, resamples systematically and (on shift) diffuses the survivors.

Split of work: rollouts and the exponentiated-cost weights are HIP kernels; the resampling itself is
a cumulative-sum search whose float summation order decides which particle survives, so it is done
on the host on the gathered weights, in the reference's order.

Sharded runs (one process per GPU): every rank holds the whole particle set, rolls out only its contiguous
block, all-gathers the (P,H) costs - the path's one exchange - and then computes the same weights and the
same resampling as every other rank (identical seeds), so the replicas stay bit-identical without a broadcast.
"""
import copy
import random

import numpy as np

from .control_utils import generate_noise
from .controller import Controller
from .sharding import local_block


def systematic_resample_indices(weights, first_pointer):
    """Low-variance resampling: pointers ``first_pointer + m/M`` walk the running sum of ``weights``.

    Equivalent to the serial walk of particle_filter_controller.py:159-171 (``while c < u: c += w[i]``):
    ``np.cumsum`` accumulates in the same order, a pointer selects the first particle whose running sum
    reaches it, pointers beyond the total select the last particle, and a pointer at 0 selects index -1
    exactly as the reference's ``act_seq[i - 1]`` with ``i == 0`` does."""
    M = weights.shape[0]
    pointers = first_pointer + np.arange(M) * 1.0 / M * 1.0
    running = np.cumsum(weights)
    idx = np.minimum(np.searchsorted(running, pointers, side="left"), M - 1)
    idx[pointers <= 0.0] = -1
    return idx


class PFMPC(Controller):
    def __init__(self, d_state, d_obs, d_action, horizon, cov_shift, cov_resample, base_action, lam,
                 num_particles, gamma, n_iters, action_lows, action_highs, set_sim_state_fn=None, rollout_fn=None,
                 sample_mode="mean", batch_size=1, filter_coeffs=[1., 0., 0.], seed=0, device=0, comm=None):
        super().__init__(d_state, d_obs, d_action, action_lows, action_highs, horizon, gamma, n_iters,
                         set_sim_state_fn, rollout_fn, sample_mode, batch_size, seed, device=device, comm=comm)
        if num_particles % self.dev.comm.world_size != 0:
            raise AssertionError("Number of particles must be divisible by number of shards")
        self.lam = lam
        self.num_particles = num_particles
        self.base_action = base_action
        self.filter_coeffs = filter_coeffs
        self.cov_shift = np.diag(np.full(self.d_action, float(cov_shift)))
        self.cov_resample = np.diag(np.full(self.d_action, float(cov_resample)))
        random.seed(self.seed_val)                       # the reference seeds the global `random` here (:66)
        self.mean_action = np.zeros((horizon, d_action))
        self.action_samples = self._fresh_samples()

    # -- sampling ---------------------------------------------------------------------------------
    def _fresh_samples(self):
        """Initial particle set: filtered N(0, cov_resample) draws from the controller seed (:68-70)."""
        return generate_noise(self.cov_resample, self.filter_coeffs, shape=(self.num_particles, self.horizon),
                              base_seed=self.seed_val)

    def sample_actions(self):
        return self.action_samples

    def generate_rollouts(self, state):
        """:74-90 - the particles are passed as deviations from their mean, the only form rollout_fn takes."""
        self._set_sim_state_fn(copy.deepcopy(state))
        off, n = local_block(self.num_particles, self.dev.comm.rank, self.dev.comm.world_size)
        return self._rollout_fn(n, self.horizon, self.mean_action,
                                (self.action_samples - self.mean_action)[off:off + n], mode="open_loop")

    # -- update -------------------------------------------------------------------------------------
    def _exp_util(self, costs):
        """softmax(-cost_to_go[:, 0] / lam) (:104-113), by the HIP softmax kernels."""
        costs = self.dev.to_device(costs, "costs")
        if self.dev.comm.world_size > 1:                  # the exchange: every rank sees all P cost rows
            costs = self.dev.comm.all_gather_flat(costs.reshape(-1)).reshape(self.num_particles, self.horizon)
        n = self.dev.softmax_update(costs, self.dev.zero_actions(costs.shape[0], costs), self.lam, 0.0,
                                    update_mean=False, replicated=True)
        return self.dev.softmax_weights(n).cpu().numpy()

    def _resampling(self, act_seq, weights, low_variance=True):
        if low_variance:
            M = act_seq.shape[0]
            return act_seq[systematic_resample_indices(weights, random.uniform(0.0, 1.0 / M * 1.0))]
        return np.array(random.choices(self.action_samples, weights=weights, k=self.num_particles))

    def _update_distribution(self, trajectories):
        """:92-102 - weights, reseed both global generators with seed + step, resample, recentre."""
        w = self._exp_util(trajectories["costs"])
        step_seed = self.seed_val + self.num_steps
        random.seed(step_seed)
        np.random.seed(step_seed)
        self.action_samples = self._resampling(self.action_samples, w, low_variance=True)
        self.mean_action = self.action_samples.mean(axis=0)

    def _get_next_action(self, state, mode='mean'):
        return self.action_samples.mean(axis=0)[0].copy()

    # -- time shift ---------------------------------------------------------------------------------
    def _shift(self):
        """:127-150 - advance every sequence one step, diffuse with fresh filtered noise, append the base action."""
        moved = self.action_samples
        moved[:, :-1] = moved[:, 1:]
        jitter = generate_noise(self.cov_shift, self.filter_coeffs, shape=(self.num_particles, self.horizon),
                                base_seed=self.seed_val + self.num_steps)
        moved = moved + jitter
        tails = {'null': lambda: np.zeros((self.num_particles, self.d_action)),
                 'repeat': lambda: moved[:, -2],
                 'random': lambda: np.random.normal(0, self.cov_resample, self.d_action)}
        if self.base_action not in tails:
            raise NotImplementedError("invalid option for base action during shift")
        moved[:, -1] = tails[self.base_action]()
        self.action_samples = moved

    def reset(self):
        self.num_steps = 0
        self.mean_action = np.zeros((self.horizon, self.d_action))
        self.action_samples = self._fresh_samples()

    def _calc_val(self, trajectories):
        raise NotImplementedError("_calc val not implemented yet")

"""numpy restatement of the controller half of the hot path  --  TEST ORACLE, NOT PRODUCT CODE.

Pinned: every function here is checked against golden vectors produced by running the
reference itself (tests/golden/*.npz, generator tests/golden/make_fixtures.py).
Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.

Written as pure functions on (mean, cov) rather than as classes; each cites the reference
lines it follows (paths relative to /root/reference).
"""
import random

import numpy as np


# ---------------------------------------------------------------------------- a3 / a4
def generate_noise(cov, filter_coeffs, shape, base_seed):
    """mjmpc/utils/control_utils.py:24-34.  Legacy global seeding + multivariate normal, then the
    IN-PLACE recursive 3-tap filter for t >= 2 (t-1, t-2 already filtered)."""
    np.random.seed(base_seed)
    b0, b1, b2 = filter_coeffs
    n = cov.shape[0]
    eps = np.random.multivariate_normal(mean=np.zeros((n,)), cov=cov, size=shape)
    for t in range(2, eps.shape[1]):
        eps[:, t, :] = b0 * eps[:, t, :] + b1 * eps[:, t - 1, :] + b2 * eps[:, t - 2, :]
    return eps


def gamma_seq(gamma, horizon):
    """mjmpc/control/controller.py:71."""
    return np.cumprod([1.0] + [gamma] * (horizon - 1)).reshape(1, horizon)


def cost_to_go(cost_seq, gseq):
    """mjmpc/utils/control_utils.py:37-46."""
    if np.any(gseq == 0):
        return cost_seq
    c = gseq * cost_seq
    c = np.cumsum(c[:, ::-1], axis=-1)[:, ::-1]
    return c / gseq


def softmax0(x):
    """scipy.special.softmax(x, axis=0): max-subtracted, no epsilon (mppi.py:96)."""
    e = np.exp(x - np.max(x, axis=0, keepdims=True))
    return e / np.sum(e, axis=0, keepdims=True)


def logsumexp_b(x, b):
    """scipy.special.logsumexp(x, b=b) for scalar b > 0."""
    m = np.max(x)
    return m + np.log(np.sum(b * np.exp(x - m)))


# ---------------------------------------------------------------------------- a16
def shift_mean(mean, base_action, init_cov=None):
    """OLGaussianMPC._shift, mjmpc/control/olgaussian_mpc.py:116-129 ('repeat' copies the row
    that was last BEFORE the roll, because mean[-2] is read after mean[:-1] = mean[1:])."""
    mean = mean.copy()
    mean[:-1] = mean[1:]
    if base_action == "random":
        mean[-1] = np.random.normal(0, init_cov, mean.shape[1])
    elif base_action == "null":
        mean[-1] = 0.0
    elif base_action == "repeat":
        mean[-1] = mean[-2]
    else:
        raise NotImplementedError("invalid option for base action during shift")
    return mean


# ---------------------------------------------------------------------------- a11 / a17 (MPPI)
def mppi_control_costs(mean, cov, delta, gseq, alpha, time_based_weights):
    """MPPI._control_costs, mjmpc/control/mppi.py:99-111."""
    if alpha == 1:
        return np.zeros(delta.shape[:2] if time_based_weights else delta.shape[0])
    u_n = mean.dot(np.linalg.inv(cov))[None]
    cc = np.sum(0.5 * u_n * (mean[None] + 2.0 * delta), axis=-1)
    cc = cost_to_go(cc, gseq)
    return cc if time_based_weights else cc[:, 0]


def mppi_weights(costs, actions, mean, cov, gseq, lam, alpha, time_based_weights):
    """MPPI._exp_util, mjmpc/control/mppi.py:84-97."""
    delta = actions - mean[None]
    tc = cost_to_go(costs.copy(), gseq)
    if not time_based_weights:
        tc = tc[:, 0]
    total = tc + lam * mppi_control_costs(mean, cov, delta, gseq, alpha, time_based_weights)
    return softmax0((-1.0 / lam) * total)


def mppi_update(costs, actions, mean, cov, gseq, lam, alpha, step_size, time_based_weights=False):
    """MPPI._update_distribution, mjmpc/control/mppi.py:69-82."""
    w = mppi_weights(costs, actions, mean, cov, gseq, lam, alpha, time_based_weights)
    weighted = np.sum((w.T * actions.T).T, axis=0)
    return (1.0 - step_size) * mean + step_size * weighted


def mppi_value(costs, actions, mean, cov, gseq, lam, alpha):
    """MPPI._calc_val, mjmpc/control/mppi.py:113-131 (time_based_weights=False only: the
    reference raises for True, see make_fixtures.py)."""
    delta = actions - mean[None]
    tc = cost_to_go(costs.copy(), gseq)[:, 0]
    total = tc + lam * mppi_control_costs(mean, cov, delta, gseq, alpha, False)
    return -lam * logsumexp_b((-1.0 / lam) * total, 1.0 / total.shape[0])


# ---------------------------------------------------------------------------- MPPIQ
def mppiq_control_costs(mean, cov, delta, alpha):
    """MPPIQ._control_costs, mjmpc/control/mppiq.py:128-136: per-step, NOT accumulated."""
    if alpha == 1:
        return np.zeros(delta.shape[:2])
    u_n = mean.dot(np.linalg.inv(cov))[None]
    return np.sum(0.5 * u_n * (mean[None] + 2.0 * delta), axis=-1)


def mppiq_returns(total_costs, qvals, gamma, td_lam):
    """MPPIQ.calculate_returns, mjmpc/control/mppiq.py:104-126: TD(lambda) blend of the per-step costs and
    the Q estimates; without estimates Q is zero except for the last step's own cost."""
    P, H = total_costs.shape
    if qvals is None:
        qvals = np.zeros((P, H))
        qvals[:, -1] = total_costs[:, -1]
    td = total_costs[:, :-1] + gamma * qvals[:, 1:] - qvals[:, :-1]
    if H == 1:
        wseq = np.array([1.0])
    else:
        wseq = np.cumprod([1.0] + [gamma * td_lam] * (H - 2)).reshape(1, H - 1)
    q_lam = qvals[:, :-1] + td_lam * cost_to_go(td, wseq)
    return np.hstack([q_lam, qvals[:, [-1]]])


def mppiq_update(costs, actions, qvals, mean, cov, beta, alpha, gamma, td_lam, step_size, time_based_weights):
    """MPPIQ._update_distribution / _exp_util, mjmpc/control/mppiq.py:73-102."""
    delta = actions - mean[None]
    q_hat = mppiq_returns(costs + beta * mppiq_control_costs(mean, cov, delta, alpha), qvals, gamma, td_lam)
    if not time_based_weights:
        q_hat = q_hat[:, 0]
    w = softmax0((-1.0 / beta) * q_hat)
    return (1.0 - step_size) * mean + step_size * np.sum((w.T * actions.T).T, axis=0)


def mppiq_value(costs, actions, qvals, mean, cov, beta, alpha, gamma, td_lam):
    """MPPIQ._calc_val, mjmpc/control/mppiq.py:138-165."""
    delta = actions - mean[None]
    q0 = mppiq_returns(costs + beta * mppiq_control_costs(mean, cov, delta, alpha), qvals, gamma, td_lam)[:, 0]
    return -beta * logsumexp_b((-1.0 / beta) * q0, 1.0 / q0.shape[0])


# ---------------------------------------------------------------------------- a12 (CEM)
def cem_update(costs, actions, mean, cov, gseq, elite_frac, step_size, cov_type):
    """CEM._update_distribution, mjmpc/control/cem.py:63-86."""
    P, H, A = actions.shape
    k = int(P * elite_frac)
    q0 = cost_to_go(costs.copy(), gseq)[:, 0]
    ids = np.argsort(q0, axis=-1)[:k]
    elite_actions = actions[ids]
    d = (actions - mean[None])[ids].reshape(H * k, A)
    if cov_type == "diagonal":
        cov_upd = np.diag(np.var(d, axis=0))
    elif cov_type == "full":
        cov_upd = np.cov(d, rowvar=False)
    else:
        raise ValueError(cov_type)
    new_cov = (1.0 - step_size) * cov + step_size * cov_upd
    new_mean = (1.0 - step_size) * mean + step_size * np.mean(elite_actions, axis=0)
    return new_mean, new_cov


def cem_shift_cov(cov, beta, init_cov_vec):
    """CEM._shift, mjmpc/control/cem.py:89-95."""
    return cov + beta * np.diag(init_cov_vec)


def mean_value(costs, gseq):
    """CEM / RandomShooting ._calc_val, mjmpc/control/cem.py:107-112, random_shooting.py:65-69."""
    return np.average(cost_to_go(costs.copy(), gseq)[:, 0])


# ---------------------------------------------------------------------------- a13 (DMD-MPC)
def dmd_update(costs, actions, mean, cov, gseq, lam, step_size, update_cov, cov_type):
    """DMDMPC._update_distribution / _exp_util, mjmpc/control/gaussian_dmd.py:65-104."""
    P, H, A = actions.shape
    delta = actions - mean[None]
    w = softmax0((-1.0 / lam) * cost_to_go(costs.copy(), gseq)[:, 0])
    new_cov = cov
    if update_cov:
        if cov_type == "diagonal":
            wd = w * (delta ** 2).T                       # (A, H, P)
            cov_upd = np.diag(np.mean(np.sum(wd.T, axis=0), axis=0))
        elif cov_type == "full":
            wd = (np.sqrt(w) * delta.T).T.reshape((H * P, A))
            cov_upd = np.dot(wd.T, wd) / H
        else:
            raise ValueError("Unidentified covariance type in update_distribution")
        new_cov = (1.0 - step_size) * cov + step_size * cov_upd
    new_mean = (1.0 - step_size) * mean + step_size * np.sum((w * actions.T).T, axis=0)
    return new_mean, new_cov


def dmd_shift_cov(cov, beta, update_cov):
    """DMDMPC._shift, mjmpc/control/gaussian_dmd.py:106-113."""
    return cov + beta * np.eye(cov.shape[0]) if update_cov else cov


def dmd_value(costs, gseq, lam):
    """DMDMPC._calc_val, mjmpc/control/gaussian_dmd.py:126-139."""
    tc = cost_to_go(costs.copy(), gseq)[:, 0]
    return -lam * logsumexp_b((-1.0 / lam) * tc, 1.0 / tc.shape[0])


# ---------------------------------------------------------------------------- a14 (random shooting)
def rs_update(costs, actions, mean, gseq, step_size):
    """RandomShooting._update_distribution, mjmpc/control/random_shooting.py:52-62."""
    q = cost_to_go(costs.copy(), gseq)
    best = np.argmin(q, axis=0)[0]
    return (1.0 - step_size) * mean + step_size * actions[best]


# ---------------------------------------------------------------------------- a15 (PFMPC)
def pf_weights(costs, gseq, lam):
    """PFMPC._exp_util, mjmpc/control/particle_filter_controller.py:104-113."""
    return softmax0((-1.0 / lam) * cost_to_go(costs.copy(), gseq)[:, 0])


def pf_resample(samples, weights, seed):
    """PFMPC._update_distribution + _resampling (low variance / systematic),
    mjmpc/control/particle_filter_controller.py:92-102,159-174."""
    random.seed(seed)
    np.random.seed(seed)
    M = samples.shape[0]
    out = np.zeros_like(samples)
    r = random.uniform(0.0, 1.0 / M * 1.0)
    c, i = 0.0, 0
    for m in range(M):
        u = r + m * 1.0 / M * 1.0
        while c < u and i < M:
            c += weights[i]
            i += 1
        out[m] = samples[i - 1]
    return out, np.mean(out, axis=0)


def pf_shift(samples, cov_shift, filter_coeffs, seed, base_action, cov_resample=None):
    """PFMPC._shift, mjmpc/control/particle_filter_controller.py:127-150."""
    samples = samples.copy()
    samples[:, :-1] = samples[:, 1:]
    samples = samples + generate_noise(cov_shift, filter_coeffs, samples.shape[:2], seed)
    if base_action == "random":
        samples[:, -1] = np.random.normal(0, cov_resample, samples.shape[2])
    elif base_action == "null":
        samples[:, -1] = 0.0
    elif base_action == "repeat":
        samples[:, -1] = samples[:, -2]
    else:
        raise NotImplementedError("invalid option for base action during shift")
    return samples


# ---------------------------------------------------------------------------- multi-GPU records
# numpy statement of the per-GPU records the HIP update kernels exchange (include/mjmpc_amd.h):
# used by the CPU gloo tests as the stand-in for the kernels, and by the GPU tests to check them.
def softmax_record(costs, actions, mean, gseq, lam, want_cov=False):
    """[xmax | S | W[H*A] | C[A*A]] of one shard (time_based_weights = False, alpha = 1)."""
    P, H, A = actions.shape
    x = (-1.0 / lam) * cost_to_go(costs.copy(), gseq)[:, 0]
    xmax = np.max(x)
    e = np.exp(x - xmax)
    W = np.tensordot(e, actions, axes=(0, 0)).reshape(-1)
    C = np.zeros((A, A))
    if want_cov:
        d = actions - mean[None]
        C = np.einsum("p,pti,ptj->ij", e, d, d)
    return np.concatenate([[xmax, e.sum()], W, C.reshape(-1)])


def softmax_combine(records, mean, cov, H, A, lam, step_size, cov_mode, P_total):
    """Combine G shard records exactly as softmax_combine_kernel does -> (mean, cov, value)."""
    records = np.asarray(records).reshape(len(records), -1)
    M = records[:, 0].max()
    sc = np.exp(records[:, 0] - M)
    S = np.sum(sc * records[:, 1])
    W = (sc[:, None] * records[:, 2:2 + H * A]).sum(0).reshape(H, A)
    new_mean = (1.0 - step_size) * mean + step_size * (W / S)
    new_cov = cov
    if cov_mode:
        C = (sc[:, None] * records[:, 2 + H * A:]).sum(0).reshape(A, A) / S / H
        if cov_mode == 1:
            C = np.diag(np.diag(C))
        new_cov = (1.0 - step_size) * cov + step_size * C
    value = -lam * (np.log(S / P_total) + M)
    return new_mean, new_cov, value


def elite_flags(q_local, q_all, offset, k):
    """rank_select_kernel: elite iff fewer than k particles precede in (q0, global index) order."""
    idx_all = np.arange(q_all.shape[0])
    out = np.zeros(q_local.shape[0], dtype=bool)
    for i, qi in enumerate(q_local):
        gi = offset + i
        out[i] = np.sum((q_all < qi) | ((q_all == qi) & (idx_all < gi))) < k
    return out

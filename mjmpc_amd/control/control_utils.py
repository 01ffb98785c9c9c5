"""Host-side helpers of the controllers.

``generate_noise`` is the PARITY-MODE sampler: it has to reproduce the reference's legacy global
numpy stream bit for bit (mjmpc/utils/control_utils.py:24-34), which only the host MT19937 can do;
the performance-mode sampler is the Philox kernel behind ``mjmpc_sample_noise``.
"""
import numpy as np


def generate_noise(cov, filter_coeffs, shape, base_seed):
    """N(0, cov) samples of shape ``shape + (A,)`` from ``np.random.seed(base_seed)``, then the
    in-place recursive filter eps[t] = b0 eps[t] + b1 eps[t-1] + b2 eps[t-2] for t >= 2.

    Like the reference this reseeds the GLOBAL numpy generator."""
    np.random.seed(base_seed)
    b0, b1, b2 = filter_coeffs
    cov = np.asarray(cov, np.float64)
    dim = cov.shape[0]
    eps = np.random.multivariate_normal(mean=np.zeros((dim,)), cov=cov, size=shape)
    horizon = eps.shape[1]
    for t in range(2, horizon):
        eps[:, t, :] = b0 * eps[:, t, :] + b1 * eps[:, t - 1, :] + b2 * eps[:, t - 2, :]
    return eps


def scale_ctrl(ctrl, action_low_limit, action_up_limit, squash_fn="clip"):
    """mjmpc/utils/control_utils.py:3-12 (unused by the open-loop controllers; kept for API parity)."""
    ctrl = np.asarray(ctrl)
    if ctrl.ndim == 1:
        ctrl = ctrl[np.newaxis, :, np.newaxis]
    half = (action_up_limit - action_low_limit) / 2.0
    mid = (action_up_limit + action_low_limit) / 2.0
    if squash_fn == "clip":
        ctrl = np.clip(ctrl, -1.0, 1.0)
    elif squash_fn == "tanh":
        ctrl = np.tanh(ctrl)
    return mid[np.newaxis, :] + ctrl * half[np.newaxis, :]

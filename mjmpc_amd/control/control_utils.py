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

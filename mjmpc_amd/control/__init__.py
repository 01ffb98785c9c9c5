from .cem import CEM
from .clgaussian_mpc import CLGaussianMPC
from .controller import Controller, OLGaussianMPC
from .gaussian_dmd import DMDMPC
from .mppi import MPPI
from .mppiq import MPPIQ
from .particle_filter_controller import PFMPC
from .random_shooting import RandomShooting

__all__ = ["Controller", "OLGaussianMPC", "CLGaussianMPC", "MPPI", "MPPIQ", "CEM", "DMDMPC", "RandomShooting", "PFMPC"]

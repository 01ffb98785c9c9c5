from .cem import CEM
from .controller import Controller, OLGaussianMPC
from .gaussian_dmd import DMDMPC
from .mppi import MPPI
from .particle_filter_controller import PFMPC
from .random_shooting import RandomShooting

__all__ = ["Controller", "OLGaussianMPC", "MPPI", "CEM", "DMDMPC", "RandomShooting", "PFMPC"]

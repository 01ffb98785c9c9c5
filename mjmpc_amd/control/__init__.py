"""Controllers of the sampling-MPC hot path, under the reference's class names (mjmpc.control)."""
import importlib

# class name -> module of this package that defines it
_WHERE = {
    "Controller": "controller", "OLGaussianMPC": "controller", "CLGaussianMPC": "clgaussian_mpc",
    "MPPI": "mppi", "MPPIQ": "mppiq", "CEM": "cem", "DMDMPC": "gaussian_dmd",
    "RandomShooting": "random_shooting", "PFMPC": "particle_filter_controller",
}
__all__ = sorted(_WHERE)

for _name, _mod in _WHERE.items():
    globals()[_name] = getattr(importlib.import_module("." + _mod, __name__), _name)
del _name, _mod

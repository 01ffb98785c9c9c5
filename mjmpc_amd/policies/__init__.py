from .mpc_policy import MPCPolicy

"""String -> controller dispatcher (reference mjmpc/policies/mpc_policy.py:7-40)."""
from .. import control

_TYPES = {"cem": "CEM", "dmd": "DMDMPC", "mppi": "MPPI", "mppiq": "MPPIQ", "pfmpc": "PFMPC",
          "random_shooting": "RandomShooting"}


class MPCPolicy:
    def __init__(self, controller_type, param_dict, batch_size=1):
        self.batch_size = batch_size
        if controller_type not in _TYPES:
            raise NotImplementedError("Controller type does not exist")
        self.controller = getattr(control, _TYPES[controller_type])(**param_dict)

    def get_action(self, state, calc_val=False, hotstart=True):
        return self.controller.optimize(state, calc_val, hotstart)

    def reset(self):
        self.controller.reset()

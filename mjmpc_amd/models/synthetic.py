"""Synthetic MJCF models written from scratch for round 4 (``mjmpc_amd/models/assets/*.xml``): the KINDS of model the
reference's experiment files name beyond its three vendored XMLs - whose own assets (mj_envs, Adroit, Sawyer / Panda
scenes) are absent from the reference tree:

* ``cartpole``  examples/configs/classic_control/cartpole*.yml (cartpole_dyn_randomize.yml:23 randomizes dof_frictionloss):
                slide + hinge with friction loss;
* ``tray``      examples/configs/panda/tray_glass-v0.yml: a free-jointed object (sphere feet, explicit inertial) on a tray
                of box geoms carried by a four-joint arm with position servos;
* ``door``      examples/configs/sawyer/door-v0.yml, hand/door-v0.yml: a door leaf on an off-origin hinge, a handle, a latch
                bolt coupled to the handle by a joint equality, a static strike box, angles in degrees;
* ``fourbar``   a closed loop (connect equality), a ball-jointed pendulum with an off-origin anchor, a limited fixed tendon;
* ``gripper``   (round 5) examples/configs/hand/pen-v0.yml, sawyer/peg_insertion-v0.yml: a free capsule (a pen) lying across
                the two BOX fingers of a small gripper (capsule / box contacts), a free cylinder standing on the plane,
                joint ref / margin, geom gap.

All of them run the tree engine's GENERAL kernel instantiation (tree_rollout.hip, GEN = true).
"""
import os

import numpy as np

from .mjcf import load_mjcf
from .raw import TASK_REACH

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
FRAME_SKIP = dict(cartpole=2, tray=2, door=2, fourbar=2, gripper=2)


def synthetic_raw(name, **kw):
    kw.setdefault("frame_skip", FRAME_SKIP[name])
    kw.setdefault("task", TASK_REACH)
    return load_mjcf(os.path.join(ASSETS, name + ".xml"), **kw)


def start_state(name, raw=None):
    """A start state for MPC episodes / benchmarks (MuJoCo's qpos layout)."""
    raw = raw or synthetic_raw(name)
    qp, qv = raw.qpos0.copy(), np.zeros(raw.nv)
    if name == "cartpole":
        qp[1] = np.pi                       # the pole hangs down: swing it up
    elif name == "tray":
        qp[2] -= 0.0072                     # the glass settled on the tray under the arm's sag
    elif name == "gripper":
        qp[2], qp[9], qp[14] = 0.3034, 0.0519, -0.0099      # the pen settled on the fingers, the can on the floor, the lift's sag
    return dict(qp=qp, qv=qv, target_pos=np.asarray(raw.target_pos, float))

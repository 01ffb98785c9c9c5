"""A synthetic pen-in-hand model (not a reference asset): the work of ``pen-v0`` - "In-hand manipulation of a 6-DOF pen
with a 24-DOF Shadow Hand" (reference examples/configs/hand/pen-v0.yml:8; the Adroit assets of mj_envs are not in the
reference tree) - on the kind of model the tree engine can hold:

* the OBJECT first: a pen (capsule) on three slide and three hinge joints - six dofs about the world, which is also how
  the Adroit model carries its pen (joints OBJTx/y/z, OBJRx/y/z [EXT]) -, gravity, its two ends against the table;
* then the 24-dof hand-on-an-arm tree of ``hand24`` with POSITION SERVOS (MJCF ``<position kp>``, as Adroit's actuators
  [EXT]): the control is a joint-angle target inside the joint's range;
* GEOM-GEOM contacts with friction cones between the pen and the finger / palm capsules (``RawModel.pairs``);
* the task of pen-v0's reward: bring the pen to a target position and its axis to a target direction (``TASK_ORIENT``).

30 dofs, 15 contact points.  Listing the object first lets the sparse factorisation eliminate its dofs last (the
manipulator's root hangs under the object in the elimination tree, compile_tree.py).  It makes no claim to Adroit's
numbers: no tendons, no meshes, one contact point per capsule pair."""
import dataclasses

import numpy as np

from .hand24 import hand24_raw
from .raw import GEOM_CAPSULE, JOINT_SLIDE, TASK_ORIENT, RawActuator, RawBody, RawGeom, RawJoint, RawModel

PEN_RADIUS, PEN_HALF = 0.008, 0.07


def pen_hand_raw(gravity=(0.0, 0.0, -9.81), mu=1.0) -> RawModel:
    hand = hand24_raw(gravity=gravity)
    X, Y, Z = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
    # the pen rests (at qpos0) across the fingers of the open hand; its six joints sit at its centre
    centre = (0.74, -0.4, 0.075)
    free = dict(limited=False, damping=0.0005, armature=0.0)
    slide = dict(limited=True, damping=0.01, armature=0.0, type=JOINT_SLIDE)
    pen = [
        RawBody("pen_tx", -1, centre, joint=RawJoint(axis=X, range=(-0.3, 0.3), name="OBJTx", **slide)),
        RawBody("pen_ty", 0, (0, 0, 0), joint=RawJoint(axis=Y, range=(-0.3, 0.3), name="OBJTy", **slide)),
        RawBody("pen_tz", 1, (0, 0, 0), joint=RawJoint(axis=Z, range=(-0.3, 0.3), name="OBJTz", **slide)),
        RawBody("pen_rx", 2, (0, 0, 0), joint=RawJoint(axis=X, range=(-3.2, 3.2), name="OBJRx", **free)),
        RawBody("pen_ry", 3, (0, 0, 0), joint=RawJoint(axis=Y, range=(-3.2, 3.2), name="OBJRy", **free)),
        RawBody("pen", 4, (0, 0, 0), joint=RawJoint(axis=Z, range=(-3.2, 3.2), name="OBJRz", **free),
                geoms=[RawGeom(GEOM_CAPSULE, PEN_RADIUS, (0, -PEN_HALF, 0), (0, PEN_HALF, 0), density=400.0, collide=True,
                               margin=0.002, name="pen", friction=mu, condim=3)]),
    ]
    n_obj = len(pen)
    bodies = pen + [dataclasses.replace(b, parent=b.parent + n_obj if b.parent >= 0 else -1,
                                        geoms=[dataclasses.replace(g, collide=False, friction=mu, condim=3) for g in b.geoms])
                    for b in hand.bodies]
    # servos: the control is a joint-angle target (the action bounds are the joint ranges); stiffer on the arm
    kps = [800.0, 800.0, 400.0, 60.0] + [0.6, 1.0, 0.6, 0.4] * 5
    joints = [b.joint for b in hand.bodies if b.joint is not None]
    acts = [RawActuator(j.name, 1.0, tuple(j.range), kp=kp) for j, kp in zip(joints, kps)]
    pairs = [("g_f%d_%s" % (f, seg), "pen") for f in range(5) for seg in ("p", "m")] + \
            [("g_palm_%s" % c, "pen") for c in "abc"]
    plane = dataclasses.replace(hand.plane, friction=mu, condim=3)
    return RawModel(bodies=bodies, actuators=acts, site_body=n_obj - 1, site_pos=(0.0, 0.0, 0.0),
                    target_pos=(0.74, -0.4, 0.10), plane=plane, timestep=0.002, frame_skip=5, gravity=gravity,
                    task=TASK_ORIENT, pairs=pairs, site_axis=Y, target_dir=(0.0, 0.7071067811865476, 0.7071067811865476))


def holding_state():
    """A start state with the palm level (it faces up), the fingers stretched, their tips raised a little, and the pen
    lying across the first phalanges, 3 mm above them."""
    qp = np.zeros(30)
    qp[6 + 3] = -0.2                                    # wrist: level the palm
    for f in range(1, 5):
        qp[6 + 4 + 4 * f + 1] = -0.15                   # first flexion joints: tips up, a shallow cradle
    return dict(qp=qp, qv=np.zeros(30))

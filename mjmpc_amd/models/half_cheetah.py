"""``HalfCheetah-v0``'s model as the reference vendors it (mjmpc/envs/assets/xml/half_cheetah.xml), restated as a
RawModel.

A planar runner: floating root (slide x, slide z, hinge y on the torso), two three-joint legs with springs, dampers,
armature and limits (their own ``solreflimit`` / ``solimplimit``), gravity, eight capsules that all collide with the
floor with friction (condim 3, mu 0.4 - MuJoCo's default pyramidal cone), ``settotalmass = 14``.
Task (mjmpc/envs/basic/half_cheetah.py:7-25): frame_skip 5, reward = forward progress of qpos[0] / dt - 0.1 |a|^2,
observation = [qpos[1:], qvel].
"""
import numpy as np

from .compile import _geom_inertial
from .raw import (GEOM_CAPSULE, JOINT_HINGE, JOINT_SLIDE, MJ20_CAPSULE_CAP, TASK_FORWARD, RawActuator, RawBody, RawGeom,
                  RawJoint, RawModel, RawPlane)

_R = 0.046
_MU = 0.4                                           # half_cheetah.xml:38: friction=".4 .1 .1", condim 3


def _cap(name, pos, half, angle):
    """size="0.046 half" pos axisangle="0 1 0 angle": the capsule's z axis turned about y."""
    u = np.array([np.sin(angle), 0.0, np.cos(angle)])
    p = np.asarray(pos, float)
    return RawGeom(GEOM_CAPSULE, _R, tuple(p - half * u), tuple(p + half * u), collide=True, margin=0.0, friction=_MU,
                   condim=3, name=name)


def _leg_joint(name, rng, damping, stiffness):
    return RawJoint((0, 1, 0), range=rng, limited=True, damping=damping, armature=0.1, name=name, type=JOINT_HINGE,
                    stiffness=stiffness)


def half_cheetah_raw(frame_skip=5) -> RawModel:
    free = dict(range=(0.0, 0.0), limited=False, damping=0.0, armature=0.0, stiffness=0.0)     # half_cheetah.xml:56-58
    bodies = [
        RawBody("rootx", -1, (0.0, 0.0, 0.7), joint=RawJoint((1, 0, 0), name="rootx", type=JOINT_SLIDE, **free)),
        RawBody("rootz", 0, (0.0, 0.0, 0.0), joint=RawJoint((0, 0, 1), name="rootz", type=JOINT_SLIDE, **free)),
        RawBody("torso", 1, (0.0, 0.0, 0.0), joint=RawJoint((0, 1, 0), name="rooty", type=JOINT_HINGE, **free),
                geoms=[RawGeom(GEOM_CAPSULE, _R, (-0.5, 0.0, 0.0), (0.5, 0.0, 0.0), collide=True, friction=_MU, condim=3,
                               name="torso"),
                       _cap("head", (0.6, 0.0, 0.1), 0.15, 0.87)]),
        RawBody("bthigh", 2, (-0.5, 0.0, 0.0), joint=_leg_joint("bthigh", (-0.52, 1.05), 6.0, 240.0),
                geoms=[_cap("bthigh", (0.1, 0.0, -0.13), 0.145, -3.8)]),
        RawBody("bshin", 3, (0.16, 0.0, -0.25), joint=_leg_joint("bshin", (-0.785, 0.785), 4.5, 180.0),
                geoms=[_cap("bshin", (-0.14, 0.0, -0.07), 0.15, -2.03)]),
        RawBody("bfoot", 4, (-0.28, 0.0, -0.14), joint=_leg_joint("bfoot", (-0.4, 0.785), 3.0, 120.0),
                geoms=[_cap("bfoot", (0.03, 0.0, -0.097), 0.094, -0.27)]),
        RawBody("fthigh", 2, (0.5, 0.0, 0.0), joint=_leg_joint("fthigh", (-1.0, 0.7), 4.5, 180.0),
                geoms=[_cap("fthigh", (-0.07, 0.0, -0.12), 0.133, 0.52)]),
        RawBody("fshin", 6, (-0.14, 0.0, -0.24), joint=_leg_joint("fshin", (-1.2, 0.87), 3.0, 120.0),
                geoms=[_cap("fshin", (0.065, 0.0, -0.09), 0.106, -0.6)]),
        RawBody("ffoot", 7, (0.13, 0.0, -0.18), joint=_leg_joint("ffoot", (-0.5, 0.5), 1.5, 60.0),
                geoms=[_cap("ffoot", (0.045, 0.0, -0.07), 0.07, -0.6)]),
    ]
    # settotalmass="14" (half_cheetah.xml:33): MuJoCo scales every mass and inertia by 14 / (mass from density 1000)
    total = sum(_geom_inertial(g, MJ20_CAPSULE_CAP)[0] for b in bodies for g in b.geoms)
    for b in bodies:
        for g in b.geoms:
            g.density *= 14.0 / total
    gears = dict(bthigh=120.0, bshin=90.0, bfoot=60.0, fthigh=120.0, fshin=60.0, ffoot=30.0)     # half_cheetah.xml:89-94
    actuators = [RawActuator(n, g, (-1.0, 1.0)) for n, g in gears.items()]
    return RawModel(bodies=bodies, actuators=actuators, site_body=len(bodies) - 1, site_pos=(0.0, 0.0, 0.0), target_pos=(0.0, 0.0, 0.0),
                    plane=RawPlane(pos=(0.0, 0.0, 0.0), normal=(0.0, 0.0, 1.0), margin=0.0, friction=_MU, condim=3),
                    timestep=0.01, frame_skip=frame_skip, gravity=(0.0, 0.0, -9.81),
                    solref=(0.02, 1.0), solimp=(0.0, 0.8, 0.01, 0.5, 2.0),
                    solref_limit=(0.02, 1.0), solimp_limit=(0.0, 0.8, 0.03, 0.5, 2.0),
                    task=TASK_FORWARD, ctrl_cost=0.1, obs_skip=1)

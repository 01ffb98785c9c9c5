"""``reacher_7dof-v0`` arm: the numbers of the reference asset, in this framework's format.

Every constant below restates ``mjmpc/envs/assets/xml/sawyer.xml`` (reference, line
numbers in the comments); ``mjmpc/envs/basic/reacher_env.py:21`` gives frame_skip = 2.
The XML itself is not shipped - ``mjmpc_amd.models.mjcf.load_mjcf`` can read such a
file directly and is tested to reproduce this table when the reference is present.
"""
from .raw import (GEOM_CAPSULE, GEOM_SPHERE, RawActuator, RawBody, RawGeom, RawJoint,
                  RawModel, RawPlane)

# <default> joint armature="0.004" damping="0.8" limited="true"          sawyer.xml:5
_ARM = 0.004
_DAMP = 0.8
# <default> geom margin="0.002" contype="0" conaffinity="0"              sawyer.xml:6
_MARGIN = 0.002


def _cap(name, r, a, b):
    return RawGeom(GEOM_CAPSULE, r, a, b, margin=_MARGIN, friction=0.5, name=name)


def _sph(name, r, p, collide=False):
    return RawGeom(GEOM_SPHERE, r, p, collide=collide, margin=_MARGIN, friction=0.5, name=name)


def _hinge(name, axis, lo, hi, damping=_DAMP):
    return RawJoint(axis=axis, range=(lo, hi), limited=True, damping=damping,
                    armature=_ARM, name=name)


def reacher7dof_raw() -> RawModel:
    X, Y, Z = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
    bodies = [
        # sawyer.xml:15-21
        RawBody("r_shoulder_pan_link", -1, (0.0, -0.6, 0.0),
                joint=_hinge("r_shoulder_pan_joint", Z, -2.2854, 1.714602, damping=2.0),
                geoms=[_sph("e1", 0.05, (-0.06, 0.05, 0.2)),
                       _sph("e2", 0.05, (0.06, 0.05, 0.2)),
                       _sph("e1p", 0.03, (-0.06, 0.09, 0.2)),
                       _sph("e2p", 0.03, (0.06, 0.09, 0.2)),
                       _cap("sp", 0.1, (0, 0, -0.4), (0, 0, 0.2))]),
        # sawyer.xml:23-25
        RawBody("r_shoulder_lift_link", 0, (0.1, 0.0, 0.0),
                joint=_hinge("r_shoulder_lift_joint", Y, -0.5236, 1.3963, damping=2.0),
                geoms=[_cap("sl", 0.1, (0, -0.1, 0), (0, 0.1, 0))]),
        # sawyer.xml:27-29
        RawBody("r_upper_arm_roll_link", 1, (0.0, 0.0, 0.0),
                joint=_hinge("r_upper_arm_roll_joint", X, -1.5, 1.7),
                geoms=[_cap("uar", 0.02, (-0.1, 0, 0), (0.1, 0, 0))]),
        # sawyer.xml:31-32 (no joint: welded to its parent)
        RawBody("r_upper_arm_link", 2, (0.0, 0.0, 0.0),
                geoms=[_cap("ua", 0.06, (0, 0, 0), (0.4, 0, 0))]),
        # sawyer.xml:34-36
        RawBody("r_elbow_flex_link", 3, (0.4, 0.0, 0.0),
                joint=_hinge("r_elbow_flex_joint", Y, -2.3213, 0.0),
                geoms=[_cap("ef", 0.06, (0, -0.02, 0), (0, 0.02, 0))]),
        # sawyer.xml:38-40
        RawBody("r_forearm_roll_link", 4, (0.0, 0.0, 0.0),
                joint=_hinge("r_forearm_roll_joint", X, -1.5, 1.5),
                geoms=[_cap("fr", 0.02, (-0.1, 0, 0), (0.1, 0, 0))]),
        # sawyer.xml:42-43 (welded)
        RawBody("r_forearm_link", 5, (0.0, 0.0, 0.0),
                geoms=[_cap("fa", 0.05, (0, 0, 0), (0.291, 0, 0))]),
        # sawyer.xml:45-47
        RawBody("r_wrist_flex_link", 6, (0.321, 0.0, 0.0),
                joint=_hinge("r_wrist_flex_joint", Y, -1.094, 0.0),
                geoms=[_cap("wf", 0.01, (0, -0.02, 0), (0, 0.02, 0))]),
        # sawyer.xml:49-59 (lines 51-57 are an XML comment); the sphere is the only
        # body geom with contype = conaffinity = 1
        RawBody("r_wrist_roll_link", 7, (0.0, 0.0, 0.0),
                joint=_hinge("r_wrist_roll_joint", X, -1.5, 1.5),
                geoms=[_sph("wrist_ball", 0.08, (0.03, 0, 0), collide=True)]),
    ]
    # sawyer.xml:101-109
    names = [b.joint.name for b in bodies if b.joint is not None]
    gears = [20.0, 10.0, 10.0, 10.0, 10.0, 10.0, 10.0]
    actuators = [RawActuator(n, g, (-1.0, 1.0)) for n, g in zip(names, gears)]
    return RawModel(
        bodies=bodies,
        actuators=actuators,
        site_body=8, site_pos=(0.0, 0.0, 0.0),              # "finger", sawyer.xml:59
        target_pos=(0.1, 0.1, 0.1),                         # "target", sawyer.xml:13
        plane=RawPlane(pos=(0.0, 0.5, -0.425), normal=Z, margin=_MARGIN, friction=0.5),   # sawyer.xml:11
        timestep=0.01,                                      # sawyer.xml:3
        frame_skip=2,                                       # reacher_env.py:21
        gravity=(0.0, 0.0, 0.0),                            # sawyer.xml:3
    )

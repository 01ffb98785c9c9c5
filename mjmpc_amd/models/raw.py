"""Raw (un-compiled) articulated-body description.

This is the framework's own, minimal model format: the subset of MJCF that the reference's vendored models need
(``mjmpc/envs/assets/xml/sawyer.xml``, ``swimmer.xml``, ``half_cheetah.xml``): a kinematic tree of bodies with at
most one hinge or slide joint each (a body with several joints is a chain of massless bodies, which is what
MuJoCo's kinematics does with it), joint springs, sphere / capsule geoms (``inertiafromgeom``; colliding ones
against one world plane, frictionless or with a pyramidal friction cone), joint-torque motors on some or all
joints, the inertia-box fluid model, and a task: reach a target with a tracked site (reacher) or move forward
(swimmer / half-cheetah).

``RawModel.to_flat()`` serialises it to a flat float64 vector.  The SAME flat vector
feeds two independent compilers:

* ``mjmpc_amd.models.compile.compile_arm`` (host side of the product, numpy) and
* ``oracle/reacher_ref.c::or_model_compile`` (the test oracle, plain C),

so that masses / inertias / invweight0 are cross-checked rather than shared.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

GEOM_SPHERE = 1
GEOM_CAPSULE = 2

HEADER_LEN = 56
JOINT_HINGE = 1
JOINT_SLIDE = 2
# MuJoCo 2.0 - the version the reference pins (mujoco-py >=2.0,<2.1) - computes a capsule's volume as pi r^2 (h + r): the
# two end caps count pi r^3 instead of 4/3 pi r^3.  That is what the body masses gym users have printed for years say
# to nine digits (HalfCheetah under mujoco-py 2.0: 6.36031332, 1.53524804, 1.58093995, 1.0691906, 1.42558747,
# 1.17885117, 0.84986945 with settotalmass 14; Hopper: 3.53429174, 3.92699082, 2.71433605, 5.0893801), and
# tests/test_locomotion_cpu.py holds the model compilers to those numbers.  MuJoCo >= 2.1.2 uses 4/3 (Hopper-v4:
# 3.6651914 ...): set ``RawModel.capsule_cap_factor = 4/3`` for that.  The inertia of a capsule of given mass follows
# MuJoCo's published formula in both cases ([EXT], unpinned).
MJ20_CAPSULE_CAP = 1.0
TASK_REACH = 0              # reward -(|h-g|_1 + 5 |h-g|_2), obs [qpos, qvel, h, h-g]      (reacher_env.py:29-47)
TASK_FORWARD = 1            # reward (x' - x)/dt - c |a|^2, obs [qpos[skip:], qvel]         (swimmer.py, half_cheetah.py)
TASK_ORIENT = 2             # in-hand reorientation, the shape of pen-v0's reward (examples/configs/hand/pen-v0.yml:8): the
                            # tracked site rides on the object; reward -|h-g|_2 + d.d*, d = the object's axis (site_axis
                            # carried by the site's body), d* = target_dir; obs as TASK_REACH
BODY_STRIDE = 20
GEOM_STRIDE = 16
ACT_STRIDE = 5
PAIR_STRIDE = 2


@dataclass
class RawJoint:
    axis: Sequence[float]
    range: Sequence[float]
    limited: bool = True
    damping: float = 0.0
    armature: float = 0.0
    name: str = ""
    type: int = JOINT_HINGE
    stiffness: float = 0.0                  # joint spring towards springref (MuJoCo qfrc_passive)
    springref: float = 0.0


@dataclass
class RawGeom:
    type: int
    radius: float
    a: Sequence[float]                      # sphere: centre; capsule: "from"
    b: Sequence[float] = (0.0, 0.0, 0.0)    # capsule: "to"
    density: float = 1000.0
    collide: bool = False                   # contype & conaffinity match the plane
    margin: float = 0.0
    name: str = ""
    friction: float = 1.0                   # sliding friction (MuJoCo default "1 0.005 0.0001", first entry)
    condim: int = 1


@dataclass
class RawBody:
    name: str
    parent: int                             # index into bodies list, -1 = world
    pos: Sequence[float]
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    joint: Optional[RawJoint] = None
    geoms: List[RawGeom] = field(default_factory=list)


@dataclass
class RawActuator:
    joint: str
    gear: float
    ctrlrange: Sequence[float]
    kp: float = 0.0                         # 0: motor, force = gear * clip(ctrl).  > 0: MJCF <position kp=...>, a servo -
                                            # force = kp * (clip(ctrl) - gear * q), applied through the gear


@dataclass
class RawPlane:
    pos: Sequence[float]
    normal: Sequence[float]
    margin: float
    friction: float = 1.0
    condim: int = 1


@dataclass
class RawModel:
    bodies: List[RawBody]
    actuators: List[RawActuator]
    site_body: int                          # body index carrying the tracked ("finger") site
    site_pos: Sequence[float]
    target_pos: Sequence[float]             # default world position of the "target" site
    plane: Optional[RawPlane]
    timestep: float
    frame_skip: int
    gravity: Sequence[float] = (0.0, 0.0, 0.0)
    solref: Sequence[float] = (0.02, 1.0)           # MuJoCo defaults
    solimp: Sequence[float] = (0.9, 0.95, 0.001, 0.5, 2.0)
    solref_limit: Optional[Sequence[float]] = None  # joint-limit rows (MJCF solreflimit / solimplimit); None: as above
    solimp_limit: Optional[Sequence[float]] = None
    density: float = 0.0                            # medium (MuJoCo <option density viscosity>): inertia-box fluid model
    viscosity: float = 0.0
    task: int = TASK_REACH
    ctrl_cost: float = 0.0                          # TASK_FORWARD: weight of |a|^2
    obs_skip: int = 0                               # TASK_FORWARD: leading qpos entries left out of the observation
    capsule_cap_factor: float = MJ20_CAPSULE_CAP    # capsule volume = pi r^2 h + factor * pi r^3 (see MJ20_CAPSULE_CAP)
    # geom-geom collision candidates, as names (geom on the manipulator, geom on the object): sphere / capsule pairs, one
    # contact point each (closest points of the two segments); friction / condim / margin = the larger of the two geoms'
    pairs: List[Sequence[str]] = field(default_factory=list)
    site_axis: Sequence[float] = (0.0, 0.0, 0.0)    # TASK_ORIENT: the object's axis in the frame of the site's body
    target_dir: Sequence[float] = (0.0, 0.0, 1.0)   # TASK_ORIENT: the direction that axis should point in (world)

    # ------------------------------------------------------------------
    @property
    def joint_names(self):
        return [b.joint.name for b in self.bodies if b.joint is not None]

    @property
    def nv(self):
        return len(self.joint_names)

    def dof_of_joint(self, name):
        return self.joint_names.index(name)

    def to_flat(self) -> np.ndarray:
        """Flat float64 serialisation (layout documented in include/mjmpc_amd.h)."""
        geoms = [(bi, g) for bi, b in enumerate(self.bodies) for g in b.geoms]
        h = np.zeros(HEADER_LEN)
        h[0] = len(self.bodies)
        h[1] = len(geoms)
        h[2] = len(self.actuators)
        h[3] = self.timestep
        h[4:7] = self.gravity
        h[7] = self.frame_skip
        h[8:10] = self.solref
        h[10:15] = self.solimp
        h[15] = self.site_body
        h[16:19] = self.site_pos
        h[19:22] = self.target_pos
        if self.plane is not None:
            h[22] = 1.0
            h[23:26] = self.plane.pos
            h[26:29] = self.plane.normal
            h[29] = self.plane.margin
            h[35] = self.plane.friction
            h[36] = self.plane.condim
        h[30], h[31] = self.density, self.viscosity
        h[32], h[33], h[34] = self.task, self.ctrl_cost, self.obs_skip
        h[37] = self.capsule_cap_factor
        h[38] = len(self.pairs)
        h[47:50] = self.site_axis
        h[50:53] = self.target_dir
        h[40:42] = self.solref if self.solref_limit is None else self.solref_limit
        h[42:47] = self.solimp if self.solimp_limit is None else self.solimp_limit
        out = [h]
        for b in self.bodies:
            r = np.zeros(BODY_STRIDE)
            r[0] = b.parent
            r[1:4] = b.pos
            r[4:8] = b.quat
            if b.joint is not None:
                r[8] = float(b.joint.type)
                r[17] = b.joint.stiffness
                r[18] = b.joint.springref
                r[9:12] = b.joint.axis
                r[12:14] = b.joint.range
                r[14] = 1.0 if b.joint.limited else 0.0
                r[15] = b.joint.damping
                r[16] = b.joint.armature
            out.append(r)
        for bi, g in geoms:
            r = np.zeros(GEOM_STRIDE)
            r[0] = bi
            r[1] = g.type
            r[2] = g.radius
            r[3:6] = g.a
            r[6:9] = g.b
            r[9] = g.density
            r[10] = 1.0 if g.collide else 0.0
            r[11] = g.margin
            r[12] = g.friction
            r[13] = g.condim
            out.append(r)
        for a in self.actuators:
            r = np.zeros(ACT_STRIDE)
            r[0] = self.dof_of_joint(a.joint)
            r[1] = a.gear
            r[2:4] = a.ctrlrange
            r[4] = a.kp
            out.append(r)
        names = [g.name for _, g in geoms]
        for ga, gb in self.pairs:
            if names.count(ga) != 1 or names.count(gb) != 1:
                raise ValueError("collision pair (%r, %r) must name one geom each" % (ga, gb))
            out.append(np.array([names.index(ga), names.index(gb)], float))
        return np.concatenate(out).astype(np.float64)

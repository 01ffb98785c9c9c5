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
GEOM_BOX = 3                # a = centre, b = half sizes, quat = orientation, all in the body frame
GEOM_CYLINDER = 4           # a = "from", b = "to" (the centres of the two flat ends), radius; collides with the plane only

HEADER_LEN = 80
JOINT_HINGE = 1
JOINT_SLIDE = 2
JOINT_BALL = 3              # 3 dofs (angular velocity in the body frame), qpos = unit quaternion (w, x, y, z)
JOINT_FREE = 4              # 6 dofs (world-frame linear velocity, body-frame angular velocity), qpos = position + quaternion
EQ_CONNECT = 1              # MJCF <equality><connect body1 body2 anchor>: a point shared by two bodies (3 rows)
EQ_WELD = 2                 # <weld body1 body2>: relative pose of two bodies fixed at its qpos0 value (6 rows)
EQ_JOINT = 3                # <joint joint1 joint2 polycoef>: q1 = poly(q2) (1 row)
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
BODY_STRIDE = 56
GEOM_STRIDE = 33
ACT_STRIDE = 14
PAIR_STRIDE = 12
EQ_STRIDE = 28
TENDON_MAX_JOINTS = 4
TENDON_STRIDE = 8 + 2 * TENDON_MAX_JOINTS + 7


def _solimp5(si):
    """MJCF solimp with fewer than five entries: the rest from MuJoCo's default (0.9 0.95 0.001 0.5 2)."""
    si = tuple(float(x) for x in si)
    return si + (0.9, 0.95, 0.001, 0.5, 2.0)[len(si):]


def mix_contact_solver(a, b):
    """(solref, solimp) of a contact between two geoms / a geom and the plane, each given as (solref, solimp, solmix,
    priority): MuJoCo mj_contactParam [EXT] - the higher priority wins; equal priorities: weights solmix_a : solmix_b
    (both below mjMINVAL: 1 : 1; one below: the other wins)."""
    (ra, ia, ma, pa), (rb, ib, mb, pb) = a, b
    if pa != pb:
        r, i = (ra, ia) if pa > pb else (rb, ib)
        return tuple(float(x) for x in r), _solimp5(i)
    MINVAL = 1e-15
    if ma >= MINVAL and mb >= MINVAL:
        w = ma / (ma + mb)
    elif ma < MINVAL and mb < MINVAL:
        w = 0.5
    else:
        w = 0.0 if ma < MINVAL else 1.0
    ia, ib = _solimp5(ia), _solimp5(ib)
    # solref: blended when both are in the standard format (timeconst, dampratio > 0), else - one of them gives stiffness and
    # damping directly, as negative numbers - the element-wise minimum: the stiffer of the two
    if ra[0] > 0 and rb[0] > 0:
        ref = tuple(w * x + (1 - w) * y for x, y in zip(ra, rb))
    else:
        ref = tuple(float(min(x, y)) for x, y in zip(ra, rb))
    return (ref, tuple(w * x + (1 - w) * y for x, y in zip(ia, ib)))


JOINT_NDOF = {JOINT_HINGE: 1, JOINT_SLIDE: 1, JOINT_BALL: 3, JOINT_FREE: 6}
JOINT_NQ = {JOINT_HINGE: 1, JOINT_SLIDE: 1, JOINT_BALL: 4, JOINT_FREE: 7}


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
    pos: Sequence[float] = (0.0, 0.0, 0.0)  # anchor in the body frame (hinge / ball; MJCF joint pos)
    frictionloss: float = 0.0               # dry friction: one friction-loss constraint row per dof (MuJoCo dof_frictionloss)
    # this joint's own solver parameters (MJCF solreflimit / solimplimit, solreffriction / solimpfriction); None: the model's
    solref_limit: Optional[Sequence[float]] = None
    solimp_limit: Optional[Sequence[float]] = None
    solref_friction: Optional[Sequence[float]] = None
    solimp_friction: Optional[Sequence[float]] = None
    margin: float = 0.0                     # MJCF joint margin: the limit row exists while dist < margin (hinge / slide)
    ref: float = 0.0                        # MJCF joint ref: qpos0 of a hinge / slide joint (the pose the model is drawn in)

    @property
    def ndof(self):
        return JOINT_NDOF[self.type]

    @property
    def nq(self):
        return JOINT_NQ[self.type]


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
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)    # box: orientation in the body frame
    # contact solver parameters of this geom (None: the model's solref / solimp) and how two geoms' are combined
    # (MuJoCo mj_contactParam [EXT]: the higher priority wins; equal priorities: solref / solimp averaged with weights
    # solmix_1 : solmix_2, friction / condim / margin the larger of the two)
    solref: Optional[Sequence[float]] = None
    solimp: Optional[Sequence[float]] = None
    solmix: float = 1.0
    priority: int = 0
    gap: float = 0.0                        # MJCF geom gap: a contact enters the solver while dist < margin - gap (MuJoCo's
                                            # includemargin; margin and gap of a pair: the larger of the two geoms' each)


@dataclass
class RawInertial:
    """MJCF <inertial>: replaces inertiafromgeom for its body (tensor about ``pos`` in the BODY frame's axes)."""
    mass: float
    pos: Sequence[float]
    inertia: Sequence[Sequence[float]]      # 3 x 3, body-frame axes


@dataclass
class RawBody:
    name: str
    parent: int                             # index into bodies list, -1 = world
    pos: Sequence[float]
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    joint: Optional[RawJoint] = None
    geoms: List[RawGeom] = field(default_factory=list)
    inertial: Optional[RawInertial] = None


@dataclass
class RawEquality:
    type: int                               # EQ_CONNECT / EQ_WELD / EQ_JOINT
    obj1: str                               # body (connect, weld) or joint (joint) name
    obj2: str = ""                          # "" = the world body / no second joint
    anchor: Sequence[float] = (0.0, 0.0, 0.0)       # connect: the shared point in body1's frame
    polycoef: Sequence[float] = (0.0, 1.0, 0.0, 0.0, 0.0)
    solref: Sequence[float] = (0.02, 1.0)
    solimp: Sequence[float] = (0.9, 0.95, 0.001, 0.5, 2.0)


@dataclass
class RawTendon:
    """MJCF <tendon><fixed>: length = sum coef_i q_i over hinge / slide joints; ``limited`` adds limit rows."""
    name: str
    joints: Sequence[Sequence]              # [(joint name, coef), ...], at most TENDON_MAX_JOINTS
    limited: bool = False
    range: Sequence[float] = (0.0, 0.0)
    margin: float = 0.0
    solref_limit: Optional[Sequence[float]] = None     # None: the model's joint-limit set
    solimp_limit: Optional[Sequence[float]] = None


@dataclass
class RawActuator:
    joint: str
    gear: float
    ctrlrange: Sequence[float]
    kp: float = 0.0                         # 0: motor, force = gear * clip(ctrl).  > 0: MJCF <position kp=...>, a servo -
                                            # force = kp * (clip(ctrl) - gear * q), applied through the gear
    tendon: str = ""                        # not "": the actuator pulls on this fixed tendon instead of a joint
    # MJCF <general> with gaintype fixed / biastype affine (and <velocity kv>): scalar force = gainprm * clip(ctrl) +
    # biasprm[0] + biasprm[1] * length + biasprm[2] * velocity, length = gear * q; None = what `kp` says
    gainprm: Optional[float] = None
    biasprm: Optional[Sequence[float]] = None
    ctrllimited: bool = True                # False: the control is not clamped (ctrlrange still bounds the action space)
    forcerange: Optional[Sequence[float]] = None    # MJCF forcelimited / forcerange: the scalar force is clamped to it

    @property
    def gain(self):
        return float(self.gainprm) if self.gainprm is not None else (self.kp if self.kp > 0 else 1.0)

    @property
    def bias(self):
        if self.biasprm is not None:
            return tuple(float(x) for x in self.biasprm)
        return (0.0, -self.kp, 0.0) if self.kp > 0 else (0.0, 0.0, 0.0)


@dataclass
class RawPlane:
    pos: Sequence[float]
    normal: Sequence[float]
    margin: float
    friction: float = 1.0
    condim: int = 1
    solref: Optional[Sequence[float]] = None        # as RawGeom
    solimp: Optional[Sequence[float]] = None
    solmix: float = 1.0
    priority: int = 0
    gap: float = 0.0


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
    # friction cones of condim-3 contacts (MJCF <option cone impratio>): "pyramidal" - MuJoCo's default, four rows
    # Jn +- mu Jt per contact - or "elliptic" - three rows (normal, two tangents) under the cone's own cost, the friction
    # rows' regulariser R = R_normal / impratio (round 5; impratio has no effect on pyramidal cones)
    cone: str = "pyramidal"
    impratio: float = 1.0
    # MJCF <sensor> elements by name -> their ``noise`` attribute: nothing on the path reads a sensor (the envs' observations
    # are qpos / qvel / site positions) and MuJoCo itself does not apply the value; kept so that ``randomize_dynamics``'
    # ``sensor_noise`` entries (gym_env_wrapper.py:396-398) find their sensor and consume their draw
    sensors: dict = field(default_factory=dict)
    # geom-geom collision candidates, as names (geom on the manipulator, geom on the object): sphere / capsule pairs, one
    # contact point each (closest points of the two segments); friction / condim / margin = the larger of the two geoms'
    pairs: List[Sequence[str]] = field(default_factory=list)
    # MJCF <pair> attributes that replace what the two geoms would give: {(geom1, geom2): {"condim", "friction", "margin",
    # "solref", "solimp"}} (any subset)
    pair_params: dict = field(default_factory=dict)
    site_axis: Sequence[float] = (0.0, 0.0, 0.0)    # TASK_ORIENT: the object's axis in the frame of the site's body
    target_dir: Sequence[float] = (0.0, 0.0, 1.0)   # TASK_ORIENT: the direction that axis should point in (world)
    world_geoms: List[RawGeom] = field(default_factory=list)   # static sphere / capsule / box geoms of the world body
                                                                # (collide only through ``pairs``)
    equalities: List[RawEquality] = field(default_factory=list)
    tendons: List[RawTendon] = field(default_factory=list)
    solref_friction: Sequence[float] = (0.02, 1.0)  # friction-loss rows (MJCF solreffriction / solimpfriction)
    solimp_friction: Sequence[float] = (0.9, 0.95, 0.001, 0.5, 2.0)

    # ------------------------------------------------------------------
    @property
    def joint_names(self):
        return [b.joint.name for b in self.bodies if b.joint is not None]

    @property
    def joints(self):
        return [b.joint for b in self.bodies if b.joint is not None]

    @property
    def nv(self):
        return sum(j.ndof for j in self.joints)

    @property
    def nq(self):
        return sum(j.nq for j in self.joints)

    def dof_of_joint(self, name):
        """Address of the joint's first dof (MuJoCo jnt_dofadr)."""
        adr = 0
        for j in self.joints:
            if j.name == name:
                return adr
            adr += j.ndof
        raise ValueError("unknown joint %r" % (name,))

    def qpos_of_joint(self, name):
        adr = 0
        for j in self.joints:
            if j.name == name:
                return adr
            adr += j.nq
        raise ValueError("unknown joint %r" % (name,))

    @property
    def qpos0(self):
        """MuJoCo's qpos0: ``ref`` (default zero) for hinge / slide joints, the identity quaternion for a ball joint, the
        body's own position and orientation for a free joint."""
        out = []
        for b in self.bodies:
            if b.joint is None:
                continue
            if b.joint.type == JOINT_BALL:
                out += [1.0, 0.0, 0.0, 0.0]
            elif b.joint.type == JOINT_FREE:
                out += list(b.pos) + list(np.asarray(b.quat, float) / np.linalg.norm(b.quat))
            else:
                out += [float(b.joint.ref)]
        return np.array(out, float)

    def pair_contact(self, ga, gb, key=None):
        """(condim, friction, margin, solref, solimp) of the contacts of a geom pair: MuJoCo mj_contactParam [EXT] - the
        larger condim / friction / margin, solref / solimp mixed (mix_contact_solver) - unless the model's <pair> says
        otherwise (``pair_params``)."""
        def cset(g):
            return (self.solref if g.solref is None else g.solref, self.solimp if g.solimp is None else g.solimp, g.solmix, g.priority)
        solref, solimp = mix_contact_solver(cset(ga), cset(gb))
        if ga.priority != gb.priority:
            top = ga if ga.priority > gb.priority else gb
            condim, mu = int(top.condim), float(top.friction)
        else:
            condim, mu = max(int(ga.condim), int(gb.condim)), max(float(ga.friction), float(gb.friction))
        margin = max(float(ga.margin), float(gb.margin)) - max(float(ga.gap), float(gb.gap))       # (includemargin)
        over = self.pair_params.get(tuple(key), self.pair_params.get(tuple(key)[::-1], {})) if key is not None else {}
        condim, mu, margin = int(over.get("condim", condim)), float(over.get("friction", mu)), float(over.get("margin", margin))
        solref = tuple(float(x) for x in over.get("solref", solref))
        solimp = _solimp5(over.get("solimp", solimp))
        return condim, mu, margin, solref, solimp

    def to_flat(self) -> np.ndarray:
        """Flat float64 serialisation (layout documented in include/mjmpc_amd.h)."""
        geoms = [(bi, g) for bi, b in enumerate(self.bodies) for g in b.geoms] + [(-1, g) for g in self.world_geoms]
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
            h[63:65] = self.solref if self.plane.solref is None else self.plane.solref
            h[65:70] = _solimp5(self.solimp if self.plane.solimp is None else self.plane.solimp)
            h[70], h[71] = self.plane.solmix, self.plane.priority
            h[72] = self.plane.gap
        h[30], h[31] = self.density, self.viscosity
        h[32], h[33], h[34] = self.task, self.ctrl_cost, self.obs_skip
        h[37] = self.capsule_cap_factor
        if self.cone not in ("pyramidal", "elliptic"):
            raise ValueError("cone must be pyramidal or elliptic")
        h[73], h[74] = (1.0 if self.cone == "elliptic" else 0.0), float(self.impratio)
        h[38] = len(self.pairs)
        h[47:50] = self.site_axis
        h[50:53] = self.target_dir
        h[40:42] = self.solref if self.solref_limit is None else self.solref_limit
        h[42:47] = self.solimp if self.solimp_limit is None else self.solimp_limit
        h[53], h[54] = len(self.equalities), len(self.tendons)
        h[56:58] = self.solref_friction
        h[58:63] = self.solimp_friction
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
                r[19:22] = b.joint.pos
                r[22] = b.joint.frictionloss
                lim_ref = self.solref if self.solref_limit is None else self.solref_limit
                lim_imp = self.solimp if self.solimp_limit is None else self.solimp_limit
                r[40:42] = lim_ref if b.joint.solref_limit is None else b.joint.solref_limit
                r[42:47] = _solimp5(lim_imp if b.joint.solimp_limit is None else b.joint.solimp_limit)
                r[47:49] = self.solref_friction if b.joint.solref_friction is None else b.joint.solref_friction
                r[49:54] = _solimp5(self.solimp_friction if b.joint.solimp_friction is None else b.joint.solimp_friction)
                r[54], r[55] = b.joint.margin, b.joint.ref
            if b.inertial is not None:
                r[23] = 1.0
                r[24] = b.inertial.mass
                r[25:28] = b.inertial.pos
                r[28:37] = np.asarray(b.inertial.inertia, float).reshape(9)
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
            r[14:18] = g.quat
            r[18] = g.gap
            r[24:26] = self.solref if g.solref is None else g.solref
            r[26:31] = _solimp5(self.solimp if g.solimp is None else g.solimp)
            r[31], r[32] = g.solmix, g.priority
            out.append(r)
        tnames = [t.name for t in self.tendons]
        for a in self.actuators:
            r = np.zeros(ACT_STRIDE)
            if a.tendon:
                r[0] = tnames.index(a.tendon)
                r[5] = 1.0
            else:
                r[0] = self.dof_of_joint(a.joint)
            r[1] = a.gear
            r[2:4] = a.ctrlrange
            r[4] = a.kp
            r[6] = a.gain
            r[7:10] = a.bias
            r[10] = 1.0 if a.ctrllimited else 0.0
            if a.forcerange is not None:
                r[11], r[12:14] = 1.0, a.forcerange
            out.append(r)
        names = [g.name for _, g in geoms]
        for ga, gb in self.pairs:
            if names.count(ga) != 1 or names.count(gb) != 1:
                raise ValueError("collision pair (%r, %r) must name one geom each" % (ga, gb))
            gd = {g.name: g for _, g in geoms}
            condim, mu, margin, solref, solimp = self.pair_contact(gd[ga], gd[gb], (ga, gb))
            out.append(np.array([names.index(ga), names.index(gb), condim, mu, margin, *solref, *solimp], float))
        bnames = [b.name for b in self.bodies]
        for e in self.equalities:
            r = np.zeros(EQ_STRIDE)
            r[0] = e.type
            if e.type == EQ_JOINT:
                r[1] = self.dof_of_joint(e.obj1)
                r[2] = self.dof_of_joint(e.obj2) if e.obj2 else -1
            else:
                r[1] = bnames.index(e.obj1)
                r[2] = bnames.index(e.obj2) if e.obj2 else -1
            r[3:6] = e.anchor
            r[6:11] = e.polycoef
            r[11:13] = e.solref
            r[13:18] = e.solimp
            out.append(r)
        for t in self.tendons:
            if not 1 <= len(t.joints) <= TENDON_MAX_JOINTS:
                raise ValueError("a fixed tendon takes 1..%d joints" % TENDON_MAX_JOINTS)
            r = np.zeros(TENDON_STRIDE)
            r[0] = len(t.joints)
            r[1] = 1.0 if t.limited else 0.0
            r[2:4] = t.range
            r[4] = t.margin
            for k, (jn, coef) in enumerate(t.joints):
                r[8 + 2 * k] = self.dof_of_joint(jn)
                r[9 + 2 * k] = coef
            lim_ref = self.solref if self.solref_limit is None else self.solref_limit
            lim_imp = self.solimp if self.solimp_limit is None else self.solimp_limit
            r[8 + 2 * TENDON_MAX_JOINTS:10 + 2 * TENDON_MAX_JOINTS] = lim_ref if t.solref_limit is None else t.solref_limit
            r[10 + 2 * TENDON_MAX_JOINTS:15 + 2 * TENDON_MAX_JOINTS] = _solimp5(lim_imp if t.solimp_limit is None else t.solimp_limit)
            out.append(r)
        return np.concatenate(out).astype(np.float64)

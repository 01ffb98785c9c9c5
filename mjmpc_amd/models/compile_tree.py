"""Host-side model compiler for kinematic TREES of hinge / slide links: RawModel -> the constant block of the tree
kernel (``mjmpc_amd/csrc/tree_rollout.hip``, layout mirrored by ``csrc/tree_model.h``).

Same conventions as ``compile.compile_arm`` (one link per joint, welded bodies merged into the link that carries
them, link frames world-aligned at qpos0, MuJoCo's inertiafromgeom and ``mj_setConst`` constants), without the
serial-chain restriction: a link may carry several child links.  The kernel wants the links numbered depth-first, so
that a subtree is a contiguous index range; an MJCF file lists its bodies that way, and a model that does not is
rejected.  Up to 32 dofs (hinge or slide; a body with several joints arrives as a chain of massless bodies), joint
springs, motors on any subset of the joints, up to 16 contact points against one plane (a colliding capsule is its two
end spheres; frictionless rows or pyramidal friction cones), the inertia-box fluid model, and the task (reach /
forward progress) the kernel's cost and observation follow.
"""
from dataclasses import dataclass

import numpy as np

from .compile import _geom_inertial, _quat2mat, _shift_inertia, principal_inertia
from .raw import (EQ_CONNECT, EQ_JOINT, EQ_WELD, GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, JOINT_BALL, JOINT_FREE, JOINT_HINGE, JOINT_SLIDE,
                  TASK_FORWARD, TASK_ORIENT, TASK_REACH, RawModel, mix_contact_solver)

TL = 32                     # lanes per particle
TREE_MAX_SPHERES = 16
SPH_STRIDE = 24             # contact record: [0] link A, [1:4] centre / segment start on A, [4] radius, [5] margin,
                            # [6] invweight (both bodies), [7] mu, [8:11] capsule axis (plane contacts: frame hint) /
                            # segment vector on A (geom-geom), [11] depth of link A in the elimination tree, [12] kind
                            # (0 sphere-plane, 1 geom-geom), [13] link B, [14:17] segment start on B, [17] radius B,
                            # [18:21] segment vector on B
MJ_MINIMP, MJ_MAXIMP = 1e-4, 0.9999     # MuJoCo's clamp on solimp (getsolparam)


def _mat2quat(R):
    """Unit quaternion (w, x, y, z) of a rotation matrix."""
    R = np.asarray(R, float)
    tr = np.trace(R)
    if tr > 0:
        sq = np.sqrt(tr + 1.0) * 2
        q = [0.25 * sq, (R[2, 1] - R[1, 2]) / sq, (R[0, 2] - R[2, 0]) / sq, (R[1, 0] - R[0, 1]) / sq]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        sq = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / sq, 0.25 * sq, (R[0, 1] + R[1, 0]) / sq, (R[0, 2] + R[2, 0]) / sq]
    elif R[1, 1] > R[2, 2]:
        sq = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / sq, (R[0, 1] + R[1, 0]) / sq, 0.25 * sq, (R[1, 2] + R[2, 1]) / sq]
    else:
        sq = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / sq, (R[0, 2] + R[2, 0]) / sq, (R[1, 2] + R[2, 1]) / sq, 0.25 * sq]
    return np.asarray(q, float)
# link kinds (T_JTYPE): one link = one dof.  A ball joint is three links (the first holds the quaternion and turns the
# frame, the other two ride along: their axes are the body's own y and z), a free joint three slides along the WORLD axes
# followed by a ball
LINK_HINGE, LINK_SLIDE, LINK_BALL_X, LINK_BALL_Y, LINK_BALL_Z = 1, 2, 3, 4, 5
# contact-record kinds ([12]): sphere/plane, segment/segment, sphere(A)/box(B), box(A)/sphere(B), connect equality,
# dof row (joint equality or fixed-tendon limit)
PT_PLANE, PT_SEGSEG, PT_SPHERE_BOX, PT_BOX_SPHERE, PT_CONNECT, PT_DOFROW, PT_WELD = 0, 1, 2, 3, 4, 5, 6
# round 5: a cylinder's candidate point k (record [22]) on the plane; a capsule (A) against a box (B) and the other way
# round - candidate k of three: where the capsule's axis comes nearest to the box, its two ends.  A box's corners on the
# plane stay PT_PLANE records with [23] = 8 (the group's size), [22] = the corner's index and [14:17] = the box centre
PT_PLANE_CYL, PT_CAPSULE_BOX, PT_BOX_CAPSULE, PT_BOX_BOX = 7, 8, 9, 10     # (box-box: contact k of four; see tree_model.h)
PT_SEG_CYL, PT_CYL_SEG = 11, 12     # a sphere / capsule (geom A) against a cylinder (geom B), and the other way round
PEXT_STRIDE = 24            # per contact record, general instantiation: [0:3] box half sizes, [3:12] box orientation in its
                            # link's frame (row-major) | dof row: [0] 0 joint equality / 1 tendon limit, [1] coef A, [2] coef
                            # B, [3:5] range, [5] margin, [6:11] polycoef; [12:19] the row's solver set {K, B, dmin, dmax,
                            # width, mid, power} (equalities), [19] bilateral flag
TREE_NQ_MAX = 40

SOL_CLASSES = 8                 # distinct solver-parameter sets a model may use (contacts, joint limits, friction loss, tendon limits)
TREE_LAYOUT = [
    # ---- staged in LDS by the kernel: per-link constants, scalars, contact records
    ("off", 3 * TL), ("axis", 3 * TL), ("mass", TL), ("com", 3 * TL), ("inertia", 6 * TL),
    ("armature", TL), ("damping", TL), ("range_lo", TL), ("range_hi", TL), ("limited", TL), ("gear", TL),
    ("ctrl_lo", TL), ("ctrl_hi", TL), ("dof_invweight0", TL),
    ("stiffness", TL), ("springref", TL), ("fbox", 3 * TL), ("frot", 9 * TL),
    ("kpg", TL),                    # actuator bias on the length, as a joint stiffness about 0: -gear^2 biasprm[1] (servo: gear^2 kp)
    ("kvg", TL),                    # ... on the velocity: -gear^2 biasprm[2]  (explicit: MuJoCo's Euler is implicit in joint damping only)
    ("tau0", TL),                   # ... constant: gear biasprm[0]
    ("tau_lo", TL), ("tau_hi", TL), # the actuator's forcerange at the joint (gear * forcerange, ordered); +-inf: none
    ("tcoef", TL), ("tpartner", TL), ("tpcoef", TL),   # actuators on fixed tendons over one or two joints (tcoef 1, tpartner -1: a joint actuator)
    ("nv", 1), ("timestep", 1), ("frame_skip", 1), ("jumps", 1), ("site_link", 1), ("site_pos", 3),
    ("n_sphere", 1), ("plane_n", 3), ("plane_d", 1),
    ("gravity", 3),
    ("nu", 1), ("task", 1), ("ctrl_cost", 1), ("obs_skip", 1), ("density", 1), ("viscosity", 1),
    ("any_friction", 1),
    ("site_axis", 3), ("target_dir", 3),    # TASK_ORIENT: object axis in the site link's frame, its target direction
    # the model's solver-parameter sets {K, B, dmin, dmax, width, mid, power}: contact records name theirs in [21], dofs in dofcls
    ("soltab", SOL_CLASSES * 7), ("dofcls", TL),
    ("spheres", TREE_MAX_SPHERES * SPH_STRIDE),
    # ---- read once per launch from global memory: topology, joint kinds, action map
    ("parent", TL), ("subsize", TL), ("anc", 5 * TL), ("ancmask", 2 * TL), ("jtype", TL), ("act", TL),
    ("eparent", TL),                # parent in the ELIMINATION tree (= parent, but the manipulator's root hangs under the
                                    # object's last link when geom-geom contacts couple the two trees)
    # tree-sparse L'DL (MuJoCo's factorisation order: leaves first, no fill-in): links of equal HEIGHT above their
    # deepest leaf are mutually unrelated and are eliminated together, one round per height
    ("depth", TL),                  # strict ancestors of the link IN THE ELIMINATION TREE
    ("n_rounds", 1),                # max height + 1
    ("elim", (TL - 1) * TL),        # [entry][lane]: my descendants sorted by height, packed k | dist << 8 | height << 16
                                    # (-1 terminates): the rows that update mine, round by round
    # ---- the GENERAL instantiation's constants (round 4): ball / free joints, friction loss, boxes, equalities, tendons
    ("gen", 1),                     # 1: the model needs the general instantiation; 2: with round 5's record kinds (their own instantiations)
    ("nq", 1), ("has_ball", 1),
    ("frictionloss", TL),           # per dof: dry friction (one friction-loss row each)
    ("qadr", TL),                   # the link's entry in MuJoCo's qpos (ball: the quaternion's w, on the BALL_X link; -1: none)
    ("qoff", TL),                   # added to the link's coordinate in qpos (a free joint's translations: the body position);
                                    # on a quaternion's three links: x, y, z of the joint's qpos0 quaternion (see qw0)
    ("pext", TREE_MAX_SPHERES * PEXT_STRIDE),
    ("qw0", TL),                    # BALL_X links: w of qpos0's quaternion q0 (ball joint: 1, 0, 0, 0; free joint: the body's
                                    # orientation).  The kernel's quaternion is RELATIVE to the qpos0 pose: qpos = q0 * q_link
    ("jmargin", TL),                # round 5: MJCF joint margin - the limit row of a hinge / slide dof exists while dist < margin
]
TREE_BLOB_LEN = sum(n for _, n in TREE_LAYOUT)
TREE_STATE_LEN = 3 * TL + 6           # device: qpos[32] | qvel[32] | target_pos[3] | fresh site[3] | quaternion w[32], per LINK
TREE_PUBLIC_STATE_LEN = TREE_NQ_MAX + TL + 6     # C ABI (per-shard states): qpos[40] (MuJoCo layout) | qvel[32] | target[3] | -


def _offsets():
    o, out = 0, {}
    for name, n in TREE_LAYOUT:
        out[name] = (o, n)
        o += n
    return out


TREE_OFFSETS = _offsets()


@dataclass
class TreeModel:
    blob: np.ndarray
    nv: int
    nu: int
    d_obs: int
    timestep: float
    frame_skip: int
    target_default: np.ndarray
    ctrl_lo: np.ndarray
    ctrl_hi: np.ndarray
    task: int
    obs_skip: int
    parent: np.ndarray             # parent link of every link (-1: root)
    max_path: int                  # links on the longest root-to-leaf path
    body_mass: np.ndarray
    body_inertia: np.ndarray       # per raw body, tensor about its centre of mass in the body frame
    body_invweight0: np.ndarray
    dof_invweight0: np.ndarray
    link_of_body: list
    nq: int = 0
    qpos0: np.ndarray = None

    def field(self, name):
        o, n = TREE_OFFSETS[name]
        return self.blob[o:o + n]


def _principal_frame(I):
    """Principal moments and axes (columns) of a body's inertia tensor in its own frame: the body axes when the
    tensor is diagonal there, else its eigenvectors - which are only defined when the moments are distinct."""
    I = np.asarray(I, float)
    off = abs(I[0, 1]) + abs(I[0, 2]) + abs(I[1, 2])
    if off <= 1e-12 * np.trace(I):
        return np.diag(I).copy(), np.eye(3)
    w, V = np.linalg.eigh(I)
    if min(w[1] - w[0], w[2] - w[1]) < 1e-9 * w[2]:
        raise NotImplementedError("fluid model: a body whose inertia has equal principal moments must be aligned with "
                                  "its body frame (the inertial frame is otherwise ambiguous)")
    return w, V


def body_inertials(raw: RawModel, overrides=None):
    """Per raw body: pose at qpos0 (R0, p0), mass, centre of mass and inertia tensor about it in the body frame
    (inertiafromgeom, or the body's explicit <inertial>), with the run-time edits of ``overrides`` applied."""
    overrides = overrides or {}
    nb = len(raw.bodies)
    R0, p0 = [None] * nb, [None] * nb
    mass, ipos, inert = np.zeros(nb), np.zeros((nb, 3)), np.zeros((nb, 3, 3))
    for i, b in enumerate(raw.bodies):
        if b.parent >= i:
            raise ValueError("bodies must be listed parents-first")
        Rp, pp = (np.eye(3), np.zeros(3)) if b.parent < 0 else (R0[b.parent], p0[b.parent])
        R0[i] = Rp @ _quat2mat(b.quat)
        p0[i] = pp + Rp @ np.asarray(b.pos, float)
        if b.inertial is not None:
            mass[i] = float(b.inertial.mass)
            ipos[i] = np.asarray(b.inertial.pos, float)
            inert[i] = np.asarray(b.inertial.inertia, float).reshape(3, 3)
        else:
            parts = [_geom_inertial(g, raw.capsule_cap_factor) for g in b.geoms]
            mass[i] = sum(m for m, _, _ in parts)
            if mass[i] > 0:
                ipos[i] = sum(m * c for m, c, _ in parts) / mass[i]
                inert[i] = sum(_shift_inertia(I, m, c - ipos[i]) for m, c, I in parts)
        if b.name in overrides.get("body_mass", {}):
            mass[i] = float(overrides["body_mass"][b.name])          # inertia / COM untouched, as in MuJoCo
        if b.name in overrides.get("body_inertia", {}):
            _, V = principal_inertia(inert[i])
            inert[i] = V @ np.diag(np.asarray(overrides["body_inertia"][b.name], float)) @ V.T
    return R0, p0, mass, ipos, inert


def compile_tree(raw: RawModel, overrides=None, base: "TreeModel" = None) -> TreeModel:
    """``overrides`` mimics run-time edits of the compiled MuJoCo model (dynamics randomization,
    mjmpc/envs/gym_env_wrapper.py:367-416), as ``compile.compile_arm`` does: ``{"body_mass": {body: m}, "body_inertia":
    {body: [I1, I2, I3]}, "dof_damping": {joint: d}, "dof_frictionloss": {joint: f}, "geom_size": {geom: [r, half, ..]},
    "geom_friction": {geom: [mu, ..]}}``.
    Like MuJoCo, such edits do NOT recompute the qpos0 constants: pass the unperturbed model as ``base`` and its
    dof / body invweight0 are kept."""
    overrides = overrides or {}
    nb = len(raw.bodies)
    R0, p0, mass, ipos, inert = body_inertials(raw, overrides)

    # ---- links: one per DOF.  Welded bodies merge into the link that carries them; a ball joint is three links, a free
    # joint three world-axis slides and a ball; the body itself (mass, geoms, children) rides on the LAST link of its joint
    link_body, link_kind, link_axis, link_first = [], [], [], []
    first_link, link_of_body, parent = [-1] * nb, [-1] * nb, []
    gen = False
    for i, b in enumerate(raw.bodies):
        pl = -1 if b.parent < 0 else link_of_body[b.parent]
        if b.joint is None:
            # a static body (welded to the world before any joint) carries geoms that never move: nothing to simulate
            link_of_body[i] = pl
            continue
        jt = b.joint
        if jt.type in (JOINT_HINGE, JOINT_SLIDE):
            ax = np.asarray(jt.axis, float)
            kinds = [(LINK_HINGE if jt.type == JOINT_HINGE else LINK_SLIDE, R0[i] @ (ax / np.linalg.norm(ax)))]
        elif jt.type == JOINT_BALL:
            kinds = [(LINK_BALL_X + c, R0[i][:, c].copy()) for c in range(3)]
        elif jt.type == JOINT_FREE:
            if b.parent >= 0:
                raise ValueError("a free joint belongs to a child of the world body")
            kinds = [(LINK_SLIDE, np.eye(3)[c]) for c in range(3)] + [(LINK_BALL_X + c, R0[i][:, c].copy()) for c in range(3)]
        else:
            raise ValueError("unknown joint type %r" % (jt.type,))
        gen = gen or jt.type in (JOINT_BALL, JOINT_FREE)
        first_link[i] = len(parent)
        for c, (kd, ax) in enumerate(kinds):
            parent.append(pl if c == 0 else len(parent) - 1)
            link_body.append(i)
            link_kind.append(kd)
            link_axis.append(ax)
            link_first.append(c == 0)
        link_of_body[i] = len(parent) - 1
    nv = len(parent)
    if nv != raw.nv:
        raise AssertionError("dof count mismatch")
    if not 1 <= nv <= TL:
        raise ValueError("tree kernel supports 1..%d dofs, got %d" % (TL, nv))
    for li in range(nv):        # a ball's three links must share a 16-lane row (their velocities meet through row shifts)
        if link_kind[li] == LINK_BALL_X and li // 16 != (li + 2) // 16:
            raise NotImplementedError("a ball joint's three dofs must not straddle lanes 15 / 16: reorder the bodies")
    parent = np.array(parent, int)
    jtype = list(link_kind)
    subsize = np.ones(nv, int)
    for i in range(nv - 1, -1, -1):
        if parent[i] >= 0:
            subsize[parent[i]] += subsize[i]
    for i in range(nv):         # depth-first numbering: every link's subtree is the index range [i, i + subsize)
        for j in range(nv):
            inside = i <= j < i + subsize[i]
            k, desc = j, False
            while k >= 0:
                if k == i:
                    desc = True
                k = parent[k]
            if inside != desc:
                raise ValueError("links must be numbered depth-first (as an MJCF file lists its bodies)")
    height = np.zeros(nv, int)
    for i in range(nv - 1, -1, -1):
        if parent[i] >= 0:
            height[parent[i]] = max(height[parent[i]], height[i] + 1)
    depth = np.zeros(nv, int)
    anc = -np.ones((5, nv), int)
    ancmask = np.zeros(nv, np.int64)
    for i in range(nv):
        k, d = i, 0
        while k >= 0:
            ancmask[i] |= 1 << k
            for e in range(5):
                if d == 1 << e:
                    anc[e, i] = k
            k, d = parent[k], d + 1
        depth[i] = d                            # links on my path to the root, myself included
    jumps = int(np.ceil(np.log2(depth.max()))) if depth.max() > 1 else 0

    # ---- elimination tree of the sparse factorisation -------------------------------------------------------
    # H = M + J' D J.  M couples a link with its kinematic ancestors; a constraint row couples every dof it touches with
    # every other: the links on the paths of BOTH bodies of a geom-geom pair or a connect equality, the two joints of a
    # joint equality or a tendon.  The factorisation eliminates links from the highest index down (leaves first), and the
    # tree it walks is the ELIMINATION TREE of that pattern (symbolic Cholesky: the parent of a link is the highest link
    # below it that its elimination touches): every row's dofs then lie on one path of it, no fill-in leaves the
    # path-indexed rows, and for a plain kinematic tree it IS the kinematic tree.  An object listed BEFORE the manipulator
    # that touches it ends up above the manipulator's root (its dofs are eliminated last: short paths); couplings between
    # siblings or across branches chain the branches up instead of being refused.
    geom_names = {g.name: (i, g) for i, b in enumerate(raw.bodies) for g in b.geoms if g.name}
    geom_names.update({g.name: (-1, g) for g in raw.world_geoms if g.name})
    bnames = [b.name for b in raw.bodies]

    def blink(i):
        return -1 if i < 0 else link_of_body[i]

    def kpath(link):
        out = []
        while link >= 0:
            out.append(link)
            link = parent[link]
        return out

    pair_geoms = []
    for ga, gb in raw.pairs:
        if ga not in geom_names or gb not in geom_names:
            raise ValueError("collision pair names an unknown geom: %r / %r" % (ga, gb))
        A, B = geom_names[ga], geom_names[gb]
        if blink(A[0]) < 0 and blink(B[0]) < 0:
            raise NotImplementedError("pair %r / %r: both geoms are static" % (ga, gb))
        if blink(B[0]) > blink(A[0]):               # the record is anchored at the higher link, whose elimination path
            A, B = B, A                             # holds the other's
        pair_geoms.append((A, B))
    coupled = [set(kpath(blink(A[0])) + kpath(blink(B[0]))) for A, B in pair_geoms]
    for e in raw.equalities:
        if e.type in (EQ_CONNECT, EQ_WELD):
            coupled.append(set(kpath(blink(bnames.index(e.obj1))) + kpath(blink(bnames.index(e.obj2)) if e.obj2 else -1)))
        elif e.type == EQ_JOINT:
            coupled.append({raw.dof_of_joint(e.obj1)} | ({raw.dof_of_joint(e.obj2)} if e.obj2 else set()))
    for t in raw.tendons:
        if t.limited:
            coupled.append({raw.dof_of_joint(jn) for jn, _ in t.joints})
    struct = [set(kpath(i)[1:]) for i in range(nv)]             # lower-index links my row touches: M's pattern ...
    for S in coupled:                                           # ... and every constraint row's clique
        for i in S:
            struct[i] |= {j for j in S if j < i}
    eparent = -np.ones(nv, int)
    for j in range(nv - 1, -1, -1):
        if struct[j]:
            pj = max(struct[j])
            eparent[j] = pj
            struct[pj] |= struct[j] - {pj}
    edepth = np.zeros(nv, int)
    for i in range(nv):
        k, d = i, 0
        while k >= 0:
            k, d = eparent[k], d + 1
        edepth[i] = d
    if edepth.max() > TL:
        raise ValueError("elimination path too long")
    eheight = np.zeros(nv, int)
    for i in range(nv - 1, -1, -1):            # (links are numbered parents-first in both trees)
        if eparent[i] >= 0:
            eheight[eparent[i]] = max(eheight[eparent[i]], eheight[i] + 1)

    def e_is_ancestor(a, k):
        while k >= 0:
            if k == a:
                return True
            k = eparent[k]
        return False

    def e_descendants(i):
        out = []
        for k in range(nv):
            j = eparent[k]
            while j >= 0 and j != i:
                j = eparent[j]
            if j == i:
                out.append(k)
        return out

    f = {name: np.zeros(n) for name, n in TREE_LAYOUT}
    origin = []
    for li in range(nv):
        bj = link_body[li]
        jt = raw.bodies[bj].joint
        anchor = p0[bj] + (R0[bj] @ np.asarray(jt.pos, float) if jt.type in (JOINT_HINGE, JOINT_BALL) else 0.0)
        origin.append(np.asarray(anchor, float))
    axis_w = list(link_axis)
    slide_like = [k == LINK_SLIDE for k in link_kind]
    qadr, nq_run = np.full(nv, -1.0), 0
    for li in range(nv):
        bj = link_body[li]
        jt = raw.bodies[bj].joint
        last = link_of_body[bj] == li
        prev = origin[parent[li]] if parent[li] >= 0 else np.zeros(3)
        axis = axis_w[li]
        members = [i for i in range(nb) if link_of_body[i] == li] if last else []
        m = mass[members].sum() if members else 0.0
        if m > 0:
            com_w = sum(mass[i] * (p0[i] + R0[i] @ ipos[i]) for i in members) / m
            I = sum(_shift_inertia(R0[i] @ inert[i] @ R0[i].T, mass[i], p0[i] + R0[i] @ ipos[i] - com_w)
                    for i in members)
        else:
            com_w, I = origin[li], np.zeros((3, 3))
        for c in range(3):
            f["off"][c * TL + li] = (origin[li] - prev)[c]
            f["axis"][c * TL + li] = axis[c]
            f["com"][c * TL + li] = (com_w - origin[li])[c]
        f["mass"][li] = m
        for k, (r, c) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            f["inertia"][k * TL + li] = I[r, c]
        f["armature"][li] = jt.armature
        f["damping"][li] = float(overrides.get("dof_damping", {}).get(jt.name, jt.damping))
        f["frictionloss"][li] = float(overrides.get("dof_frictionloss", {}).get(jt.name, jt.frictionloss))
        f["jtype"][li] = link_kind[li]
        single = jt.type in (JOINT_HINGE, JOINT_SLIDE)
        if single:
            # (MJCF joint ref = qpos0: the kernel's coordinate is qpos - ref, so everything stated on qpos - range, spring
            # reference, actuator and tendon lengths - moves by it; qoff brings qpos back at the boundary)
            f["stiffness"][li], f["springref"][li] = jt.stiffness, jt.springref - jt.ref
            f["qoff"][li] = jt.ref
            f["jmargin"][li] = jt.margin
            gen = gen or jt.margin != 0.0
        elif jt.stiffness != 0 or (jt.limited and jt.type != JOINT_BALL):
            raise NotImplementedError("ball / free joints: no springs; limits on ball joints only")
        # MuJoCo's qpos: one entry per hinge / slide, (w, x, y, z) per ball, position + quaternion per free joint
        if link_first[li]:
            nq_here = nq_run
            nq_run += jt.nq
        k_in = li - first_link[bj]
        if single:
            qadr[li] = nq_here
        elif jt.type == JOINT_BALL:
            qadr[li] = nq_here if k_in == 0 else -1
        else:
            qadr[li] = nq_here + k_in if k_in < 3 else (nq_here + 3 if k_in == 3 else -1)
            if k_in < 3:
                f["qoff"][li] = p0[bj][k_in]
            else:                                   # a free joint's quaternion is the body's ABSOLUTE orientation
                bq = np.asarray(raw.bodies[bj].quat, float) / np.linalg.norm(raw.bodies[bj].quat)
                f["qoff"][li] = bq[1 + k_in - 3]
                if k_in == 3:
                    f["qw0"][li] = bq[0]
        if jt.type == JOINT_BALL and k_in == 0:
            f["qw0"][li] = 1.0
        if raw.density > 0 or raw.viscosity > 0:
            # MuJoCo's fluid model acts body by body on the box of equal inertia, in the body's inertial frame
            massive = [i for i in members if mass[i] > 0]
            if len(massive) > 1:
                raise NotImplementedError("the fluid model needs one body per link (no welded bodies with mass)")
            if massive:
                i = massive[0]
                w, V = _principal_frame(inert[i])
                for c in range(3):
                    x = max(w[(c + 1) % 3] + w[(c + 2) % 3] - w[c], 1e-15)
                    f["fbox"][c * TL + li] = np.sqrt(x / mass[i] * 6.0)
                f["frot"][li::TL] = (R0[i] @ V).reshape(-1)
            else:
                f["frot"][li::TL] = np.eye(3).reshape(-1)
        if single:
            f["range_lo"][li], f["range_hi"][li] = jt.range[0] - jt.ref, jt.range[1] - jt.ref
            f["limited"][li] = 1.0 if jt.limited else 0.0
    nq = nq_run
    if nq > TREE_NQ_MAX:
        raise ValueError("too many qpos entries")
    f["qadr"][:] = -1.0
    f["qadr"][:nv] = qadr
    f["axis"][2 * TL + nv:3 * TL] = 1.0         # spare lanes: a unit axis keeps their (unused) rotation orthonormal
    f["jtype"][nv:] = JOINT_HINGE
    f["parent"][:] = -1.0
    f["parent"][:nv] = parent
    f["subsize"][:nv] = subsize
    f["anc"][:] = -1.0
    for e in range(5):
        f["anc"][e * TL:e * TL + nv] = anc[e]
    f["ancmask"][:nv] = ancmask & 0xFFFF
    f["ancmask"][TL:TL + nv] = ancmask >> 16
    f["eparent"][:] = -1.0
    f["eparent"][:nv] = eparent
    f["depth"][:nv] = edepth - 1
    f["n_rounds"][0] = eheight.max() + 1
    f["elim"][:] = -1.0
    for i in range(nv):
        desc = sorted(e_descendants(i), key=lambda k: (eheight[k], k))
        for e, k in enumerate(desc):
            f["elim"][e * TL + i] = k | ((edepth[k] - edepth[i]) << 8) | (eheight[k] << 16)

    nu = len(raw.actuators)
    if not 1 <= nu <= nv:
        raise ValueError("tree kernel expects between one motor and one per joint")
    ctrl_lo, ctrl_hi = np.zeros(nu), np.zeros(nu)
    f["act"][:] = -1.0
    f["tau_lo"][:], f["tau_hi"][:] = -np.inf, np.inf
    f["tcoef"][:], f["tpartner"][:] = 1.0, -1.0
    for a, act in enumerate(raw.actuators):         # action a drives the dof of its joint (any subset, any order)
        if act.tendon:
            # the tendon's one or two dofs share the action; each carries its coefficient and the other's
            tn = [t for t in raw.tendons if t.name == act.tendon]
            if len(tn) != 1 or not 1 <= len(tn[0].joints) <= 2:
                raise NotImplementedError("actuators on fixed tendons over one or two joints are compiled for the kernel")
            tdofs = [(raw.dof_of_joint(jn), float(c)) for jn, c in tn[0].joints]
        else:
            tdofs = [(raw.dof_of_joint(act.joint), 1.0)]
        for k, (d, c) in enumerate(tdofs):
            if link_kind[d] not in (LINK_HINGE, LINK_SLIDE) or raw.bodies[link_body[d]].joint.type not in (JOINT_HINGE, JOINT_SLIDE):
                raise NotImplementedError("actuators drive hinge / slide joints")
            if f["act"][d] >= 0:
                raise ValueError("two actuators on the joint of dof %d" % d)
            f["act"][d] = a
            f["tcoef"][d] = c
            if len(tdofs) == 2:
                f["tpartner"][d], f["tpcoef"][d] = tdofs[1 - k]
        d = tdofs[0][0]
        # joint torque = gear * (gain * clip(ctrl) + b0 + b1 * gear q + b2 * gear v)  (mj_fwdActuation, gaintype fixed,
        # biastype affine; motor: gain 1; <position kp>: gain kp, b1 = -kp; <velocity kv>: gain kv, b2 = -kv): the ctrl part
        # through an effective gear, the rest as a stiffness about 0, a damping-like term and a constant at the joint
        b0, b1, b2 = act.bias
        f["gear"][d] = act.gear * act.gain
        f["kpg"][d] = -act.gear * act.gear * b1
        f["kvg"][d] = -act.gear * act.gear * b2
        # (length = gear * sum coef qpos, qpos = the kernel's coordinate + ref)
        f["tau0"][d] = act.gear * b0 + act.gear * act.gear * b1 * sum(c * raw.bodies[link_body[dd]].joint.ref for dd, c in tdofs)
        f["ctrl_lo"][d], f["ctrl_hi"][d] = act.ctrlrange if act.ctrllimited else (-np.inf, np.inf)
        if act.forcerange is not None:
            ends = sorted((act.gear * act.forcerange[0], act.gear * act.forcerange[1]))
            f["tau_lo"][d], f["tau_hi"][d] = ends
        for d2, _ in tdofs[1:]:
            for name in ("gear", "kpg", "kvg", "tau0", "ctrl_lo", "ctrl_hi", "tau_lo", "tau_hi"):
                f[name][d2] = f[name][d]
        ctrl_lo[a], ctrl_hi[a] = act.ctrlrange

    # ---- constants at qpos0 (MuJoCo mj_setConst): M0, dof / body invweight0 ------------------------------
    def on_path(k, link):
        return link >= 0 and bool((ancmask[link] >> k) & 1)

    def jac_point(pt, link):
        J = np.zeros((3, nv))
        for k in range(nv):
            if on_path(k, link):
                J[:, k] = axis_w[k] if slide_like[k] else np.cross(axis_w[k], pt - origin[k])
        return J

    def jac_rot(link):
        J = np.zeros((3, nv))
        for k in range(nv):
            if on_path(k, link) and not slide_like[k]:
                J[:, k] = axis_w[k]
        return J

    M0 = np.diag(f["armature"][:nv]).astype(float)
    for i in range(nb):
        if mass[i] <= 0 or link_of_body[i] < 0:
            continue
        li = link_of_body[i]
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], li)
        Jr = jac_rot(li)
        Iw = R0[i] @ inert[i] @ R0[i].T
        M0 += mass[i] * Jp.T @ Jp + Jr.T @ Iw @ Jr
    M0inv = np.linalg.inv(M0)
    dof_iw = np.diag(M0inv).copy()
    for i, b in enumerate(raw.bodies):      # MuJoCo averages a ball's three entries, a free joint's translations and rotations
        if b.joint is not None and b.joint.type in (JOINT_BALL, JOINT_FREE):
            j = first_link[i]
            for g in range(2 if b.joint.type == JOINT_FREE else 1):
                dof_iw[j + 3 * g:j + 3 * g + 3] = dof_iw[j + 3 * g:j + 3 * g + 3].mean()
    f["dof_invweight0"][:nv] = dof_iw
    body_iw, body_iw_rot = np.zeros(nb), np.zeros(nb)
    for i in range(nb):
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], link_of_body[i])
        body_iw[i] = np.trace(Jp @ M0inv @ Jp.T) / 3.0
        Jr = jac_rot(link_of_body[i])
        body_iw_rot[i] = np.trace(Jr @ M0inv @ Jr.T) / 3.0

    # ---- scalars, site, contacts --------------------------------------------------------------------------
    f["nv"][0], f["timestep"][0], f["frame_skip"][0], f["jumps"][0] = nv, raw.timestep, raw.frame_skip, jumps
    sb = raw.site_body
    if link_of_body[sb] < 0:
        raise ValueError("the tracked site must ride on a moving body")
    f["site_link"][0] = link_of_body[sb]
    f["site_pos"][:] = p0[sb] + R0[sb] @ np.asarray(raw.site_pos, float) - origin[link_of_body[sb]]
    # contact points: a colliding sphere, the two end spheres of a colliding capsule - the "to" end first, as
    # MuJoCo's mjc_PlaneCapsule tests them - which also hand their axis to the contact frame, or the eight corners of a
    # colliding box (mjc_PlaneBox; radius 0).  Friction and condim of a contact are the larger of the two geoms'
    # (mj_contactParam with equal priorities).
    points = []
    for i, b in enumerate(raw.bodies):
        if link_of_body[i] < 0:
            continue
        for g in b.geoms:
            if not g.collide:
                continue
            size = overrides.get("geom_size", {}).get(g.name)       # run-time edit: [radius, half-length, -]
            radius = float(np.ravel(size)[0]) if size is not None else g.radius
            if g.type == GEOM_SPHERE:
                points.append((i, g, np.asarray(g.a, float), np.zeros(3), radius))
            elif g.type == GEOM_BOX:
                half = np.asarray(np.ravel(size)[:3] if size is not None else g.b, float)
                Rg = _quat2mat(g.quat)
                for e in range(8):
                    c = np.array([half[0] if e & 1 else -half[0], half[1] if e & 2 else -half[1], half[2] if e & 4 else -half[2]])
                    points.append((i, g, np.asarray(g.a, float) + Rg @ c, np.zeros(3), 0.0, ("box", e, np.asarray(g.a, float))))
            elif g.type == GEOM_CYLINDER:
                a, e = np.asarray(g.a, float), np.asarray(g.b, float)
                u, c, hh = (e - a) / np.linalg.norm(e - a), 0.5 * (a + e), 0.5 * np.linalg.norm(e - a)
                if size is not None:
                    hh = float(np.ravel(size)[1])
                for k in range(4):
                    points.append((i, g, c, u, radius, ("cyl", k, hh)))
            else:
                a, e = np.asarray(g.a, float), np.asarray(g.b, float)
                u = (e - a) / np.linalg.norm(e - a)
                if size is not None:                                # the ends move with the half-length, the centre stays
                    c, half = 0.5 * (a + e), float(np.ravel(size)[1])
                    a, e = c - half * u, c + half * u
                points += [(i, g, e, u, radius), (i, g, a, u, radius)]
    if raw.plane is None:
        points = []
    n_eq_pts = len(raw.equalities) + sum(1 for e in raw.equalities if e.type == EQ_WELD)     # (a weld takes two records)
    n_tn_pts = sum(1 for t in raw.tendons if t.limited) + sum(1 for b in raw.bodies if b.joint is not None and b.joint.type == JOINT_BALL and b.joint.limited)
    n_pair_pts = sum(3 if sorted((ga.type, gb.type)) in ([GEOM_CAPSULE, GEOM_BOX], [GEOM_CAPSULE, GEOM_CYLINDER])
                     else (4 if (ga.type, gb.type) == (GEOM_BOX, GEOM_BOX) else 1) for (_, ga), (_, gb) in pair_geoms)
    if len(points) + n_pair_pts + n_eq_pts + n_tn_pts > TREE_MAX_SPHERES:
        raise ValueError("tree kernel supports %d contact records (a capsule on the plane counts two, a box eight, a cylinder four, "
                         "a capsule-box / capsule-cylinder pair three, a box-box pair four, any other geom-geom pair, an equality and a tendon limit one each)" % TREE_MAX_SPHERES)
    if raw.plane is not None:
        n = np.asarray(raw.plane.normal, float)
        n = n / np.linalg.norm(n)
        f["plane_n"][:] = n
        f["plane_d"][0] = n @ np.asarray(raw.plane.pos, float)

    def geom_mu(g):
        mu_g = overrides.get("geom_friction", {}).get(g.name)
        return float(np.ravel(mu_g)[0]) if mu_g is not None else g.friction

    def body_w(i):
        if i < 0:
            return 0.0                                              # the world body weighs 0
        return base.body_invweight0[i] if base is not None else body_iw[i]

    def in_link(i, v):
        """A point / vector given in body i's frame, in the world-aligned frame of the body's link (world: as it is)."""
        return np.asarray(v, float) if i < 0 else R0[i] @ np.asarray(v, float)

    def link_point(i, pos):
        if i < 0 or link_of_body[i] < 0:        # static: world coordinates (a static body's pose is baked in)
            return (p0[i] + R0[i] @ np.asarray(pos, float)) if i >= 0 else np.asarray(pos, float)
        return p0[i] + R0[i] @ np.asarray(pos, float) - origin[link_of_body[i]]

    def sol_values(solref, solimp):
        tc, dr = solref
        if (tc > 0) != (dr > 0):
            raise NotImplementedError("solref must be (timeconst, dampratio), both positive, or (-stiffness, -damping), both negative")
        direct = tc <= 0                                            # MuJoCo's direct format (round 5): no refsafe clamp
        if not direct:
            tc = max(tc, 2 * raw.timestep)                          # refsafe
        dmin, dmax, width, mid, power = full_solimp(solimp)
        dmin, dmax = np.clip(dmin, MJ_MINIMP, MJ_MAXIMP), np.clip(dmax, MJ_MINIMP, MJ_MAXIMP)
        mid, width, power = np.clip(mid, MJ_MINIMP, MJ_MAXIMP), max(width, 0.0), max(power, 1.0)
        if power != int(power) or power > 64:
            raise NotImplementedError("solimp power must be an integer in [1, 64] (MuJoCo's default is 2), got %r" % (power,))
        K, B = (-tc / (dmax * dmax), -dr / dmax) if direct else (1.0 / (dmax * dmax * tc * tc * dr * dr), 2.0 / (dmax * tc))
        if width <= 1e-15:                                          # MuJoCo getimpedance: a flat impedance
            dmin = dmax = 0.5 * (dmin + dmax)
            width = 1.0
        return [K, B, dmin, dmax, width, mid, power]

    def full_solimp(si):
        return tuple(si) + (0.9, 0.95, 0.001, 0.5, 2.0)[len(si):]

    # the model's distinct solver-parameter sets (contacts after mj_contactParam's mixing, joint limits, friction loss,
    # tendon limits): a table in the block, an index per record / dof
    sol_classes = []

    def sol_class(solref, solimp):
        vals = tuple(float(x) for x in sol_values(solref, solimp))
        if vals not in sol_classes:
            if len(sol_classes) == SOL_CLASSES:
                raise NotImplementedError("more than %d distinct solref / solimp sets in one model" % SOL_CLASSES)
            sol_classes.append(vals)
            f["soltab"][7 * (len(sol_classes) - 1):7 * len(sol_classes)] = vals
        return float(sol_classes.index(vals))

    def contact_set(g):
        """(solref, solimp, solmix, priority) of a geom or of the plane, the model's set where it has none of its own."""
        return (raw.solref if g.solref is None else g.solref, raw.solimp if g.solimp is None else g.solimp, g.solmix, g.priority)

    lim_default = (raw.solref if raw.solref_limit is None else raw.solref_limit, raw.solimp if raw.solimp_limit is None else raw.solimp_limit)
    sol_class(raw.solref, raw.solimp)           # (class 0: the model's contact set - what records of models built by hand get)
    for b in raw.bodies:
        if b.joint is None:
            continue
        jt = b.joint
        lc = sol_class(lim_default[0] if jt.solref_limit is None else jt.solref_limit,
                       lim_default[1] if jt.solimp_limit is None else jt.solimp_limit) if jt.limited else 0.0
        fc = sol_class(raw.solref_friction if jt.solref_friction is None else jt.solref_friction,
                       raw.solimp_friction if jt.solimp_friction is None else jt.solimp_friction) if jt.frictionloss > 0 else 0.0
        d0 = raw.dof_of_joint(jt.name)
        f["dofcls"][d0:d0 + jt.ndof] = lc + 8.0 * fc

    for s, pt in enumerate(points):
        i, g, pos, u, radius = pt[:5]
        li = link_of_body[i]
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        rec[0] = li
        rec[1:4] = p0[i] + R0[i] @ pos - origin[li]
        rec[4] = radius
        rec[5] = max(raw.plane.margin, g.margin) - max(raw.plane.gap, g.gap)    # MuJoCo: max of the two geom margins, less the larger gap (includemargin)
        rec[6] = 0.0 + body_w(i)                            # the world body weighs 0
        condim = max(int(g.condim), int(raw.plane.condim))
        if condim not in (1, 3):
            raise NotImplementedError("contacts are condim 1 (frictionless) or 3 (pyramidal cone), got %d" % condim)
        rec[7] = max(geom_mu(g), raw.plane.friction) if condim == 3 else 0.0
        rec[8:11] = R0[i] @ u
        rec[11] = edepth[li] - 1                            # strict ancestors of the point's link (elimination tree)
        rec[13] = -1.0
        rec[21] = sol_class(*mix_contact_solver(contact_set(g), contact_set(raw.plane)))
        if len(pt) > 5 and pt[5][0] == "box":       # mjc_PlaneBox: corners above the centre are skipped, four contacts at most
            rec[22], rec[23] = pt[5][1], 8.0
            rec[14:17] = p0[i] + R0[i] @ pt[5][2] - origin[li]
            gen = True
        elif len(pt) > 5:                           # mjc_PlaneCylinder: candidate point k of four
            rec[12], rec[22], rec[23] = PT_PLANE_CYL, pt[5][1], 4.0
            rec[14] = pt[5][2]
            gen = True
    # geom-geom pairs: spheres / capsules as segments (start, vector; a sphere has a zero vector) in their links' frames;
    # ONE geom of a pair may be a box (against a sphere).  The record is anchored at link A, whose elimination path
    # contains link B (an object's links above a manipulator, an ancestor in the same tree, or the world: -1)
    s = len(points)
    for (ia, ga), (ib, gb) in pair_geoms:
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        ext = f["pext"][s * PEXT_STRIDE:(s + 1) * PEXT_STRIDE]
        la, lb = blink(ia), blink(ib)
        assert lb < 0 or e_is_ancestor(lb, la)

        def seg(i, g):
            size = overrides.get("geom_size", {}).get(g.name)
            r = float(np.ravel(size)[0]) if size is not None else g.radius
            a = np.asarray(g.a, float)
            if g.type == GEOM_SPHERE:
                return link_point(i, a), np.zeros(3), r
            if g.type == GEOM_BOX:
                return link_point(i, a), np.zeros(3), 0.0
            e = np.asarray(g.b, float)
            if size is not None:
                u, c, half = (e - a) / np.linalg.norm(e - a), 0.5 * (a + e), float(np.ravel(size)[1])
                a, e = c - half * u, c + half * u
            return link_point(i, a), in_link(i, e - a), r

        a0, da, ra = seg(ia, ga)
        b0, db, rb = seg(ib, gb)
        # the pair's contact parameters: mj_contactParam's mixing of the two geoms', or the model's <pair> element
        condim, mu_p, margin_p, psolref, psolimp = raw.pair_contact(ga, gb, (ga.name, gb.name))
        over = raw.pair_params.get((ga.name, gb.name), raw.pair_params.get((gb.name, ga.name), {}))
        if "friction" not in over:                          # (run-time geom_friction edits act on the geoms' values)
            mu_p = max(geom_mu(ga), geom_mu(gb)) if ga.priority == gb.priority else geom_mu(ga if ga.priority > gb.priority else gb)
        if condim not in (1, 3):
            raise NotImplementedError("contacts are condim 1 (frictionless) or 3 (pyramidal cone), got %d" % condim)
        rec[0], rec[1:4], rec[4] = la, a0, ra
        rec[5] = margin_p
        rec[6] = body_w(ia) + body_w(ib)
        rec[7] = mu_p if condim == 3 else 0.0
        rec[8:11] = da
        rec[11] = edepth[la] - 1
        rec[12] = PT_SEGSEG
        rec[13], rec[14:17], rec[17], rec[18:21] = lb, b0, rb, db
        rec[21] = sol_class(psolref, psolimp)
        boxes = [k for k, g in enumerate((ga, gb)) if g.type == GEOM_BOX]
        copies = 1
        if GEOM_CYLINDER in (ga.type, gb.type):
            # a sphere / capsule against a cylinder (round 5): the cylinder is its axis segment + radius, as a capsule's record
            # - only the kind tells the kernel to close it with flat caps; a capsule brings three candidate contacts
            other = gb if ga.type == GEOM_CYLINDER else ga
            if other.type not in (GEOM_SPHERE, GEOM_CAPSULE):
                raise NotImplementedError("a cylinder collides with the plane, spheres and capsules")
            rec[12] = PT_CYL_SEG if ga.type == GEOM_CYLINDER else PT_SEG_CYL
            if other.type == GEOM_CAPSULE:
                rec[23], copies = 3.0, 3
            gen = True
        if len(boxes) == 2:
            # two boxes: box A as a sphere-box record's box, box B's orientation in ITS link's frame as a quaternion + half sizes
            for (ibx, gbx), at in (((ia, ga), 0), ((ib, gb), 12)):
                size = overrides.get("geom_size", {}).get(gbx.name)
                half = np.ravel(size)[:3] if size is not None else gbx.b
                Rl = _quat2mat(gbx.quat) if ibx < 0 else R0[ibx] @ _quat2mat(gbx.quat)
                if at == 0:
                    ext[0:3], ext[3:12] = half, Rl.reshape(-1)
                else:
                    ext[12:16], ext[16:19] = _mat2quat(Rl), half
            rec[12], rec[23], copies = PT_BOX_BOX, 4.0, 4
            gen = True
        elif boxes:
            other = gb if boxes[0] == 0 else ga
            ibx, gbx = (ia, ga) if boxes[0] == 0 else (ib, gb)
            size = overrides.get("geom_size", {}).get(gbx.name)
            ext[0:3] = np.ravel(size)[:3] if size is not None else gbx.b
            Rl = _quat2mat(gbx.quat) if ibx < 0 else R0[ibx] @ _quat2mat(gbx.quat)
            ext[3:12] = Rl.reshape(-1)
            if other.type == GEOM_CAPSULE:          # three candidate contacts: the axis' nearest point, the two ends
                rec[12] = PT_BOX_CAPSULE if boxes[0] == 0 else PT_CAPSULE_BOX
                rec[23] = 3.0
                copies = 3
            else:
                rec[12] = PT_BOX_SPHERE if boxes[0] == 0 else PT_SPHERE_BOX
            gen = True
        if lb < 0:
            gen = True              # (a static second geom: the general instantiation knows the world as "link -1")
        for k in range(1, copies):
            f["spheres"][(s + k) * SPH_STRIDE:(s + k + 1) * SPH_STRIDE] = rec
            f["spheres"][(s + k) * SPH_STRIDE + 22] = k
            f["pext"][(s + k) * PEXT_STRIDE:(s + k + 1) * PEXT_STRIDE] = ext
        s += copies

    # equality constraints and tendon limits ride in the contact records too (general instantiation)
    for e in raw.equalities:
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        ext = f["pext"][s * PEXT_STRIDE:(s + 1) * PEXT_STRIDE]
        ext[12:19] = sol_values(e.solref, full_solimp(e.solimp))
        ext[19] = 1.0
        gen = True
        if e.type in (EQ_CONNECT, EQ_WELD):
            i1, i2 = bnames.index(e.obj1), (bnames.index(e.obj2) if e.obj2 else -1)
            # the shared point at qpos0, world: a connect's anchor (given in body 1); a weld's is body 2's origin
            anchor_w = (p0[i1] + R0[i1] @ np.asarray(e.anchor, float)) if e.type == EQ_CONNECT else (p0[i2] if i2 >= 0 else np.zeros(3))
            j1, j2 = i1, i2
            l1, l2 = blink(i1), blink(i2)
            if l1 < 0 and l2 < 0:
                raise NotImplementedError("connect %r / %r: both bodies are static" % (e.obj1, e.obj2))
            if l2 > l1:
                i1, i2, l1, l2 = i2, i1, l2, l1         # anchored at the higher link (a bilateral row's sign is immaterial)
            assert l2 < 0 or e_is_ancestor(l2, l1) or all(e_is_ancestor(k, l1) for k in kpath(l2))
            rec[0], rec[1:4] = l1, anchor_w - origin[l1]
            rec[13], rec[14:17] = l2, (anchor_w - origin[l2]) if l2 >= 0 else anchor_w
            rec[6] = body_w(i1) + body_w(i2)
            rec[11] = edepth[l1] - 1
            rec[12] = PT_CONNECT
            if e.type == EQ_WELD:
                # second record: the rotation rows.  ext[0:9] = body 2's orientation at qpos0 (world: identity), [20] which
                # of the record's links carries body 1
                s += 1
                rec2 = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
                ext2 = f["pext"][s * PEXT_STRIDE:(s + 1) * PEXT_STRIDE]
                rec2[0], rec2[13], rec2[11], rec2[12] = l1, l2, edepth[l1] - 1, PT_WELD
                rw = (lambda i: 0.0 if i < 0 else (base.body_invweight0_rot[i] if base is not None else body_iw_rot[i]))
                rec2[6] = rw(j1) + rw(j2)
                ext2[0:9] = (R0[j2] if j2 >= 0 else np.eye(3)).reshape(-1)
                ext2[12:19] = ext[12:19]
                ext2[19] = 1.0
                ext2[20] = 1.0 if i1 == j1 else -1.0
        elif e.type == EQ_JOINT:
            d1, d2 = raw.dof_of_joint(e.obj1), (raw.dof_of_joint(e.obj2) if e.obj2 else -1)
            for d in (d1, d2):
                if d >= 0 and link_kind[d] not in (LINK_HINGE, LINK_SLIDE):
                    raise NotImplementedError("joint equalities couple hinge / slide joints")
            w = (base.dof_invweight0 if base is not None else dof_iw)
            rec[6] = w[d1] + (w[d2] if d2 >= 0 else 0.0)
            # anchored at the higher dof; [1] which of the two carries q1 (coefficient +1)
            if d2 > d1:
                rec[0], rec[13], ext[1] = d2, d1, 1.0       # anchor = joint 2, the other (joint 1) above it
            else:
                rec[0], rec[13], ext[1] = d1, d2, 0.0
            assert rec[13] < 0 or e_is_ancestor(int(rec[13]), int(rec[0]))
            ext[0] = 0.0
            ext[6:11] = e.polycoef
            rec[11] = edepth[int(rec[0])] - 1
            rec[12] = PT_DOFROW
        else:
            raise NotImplementedError("unknown equality type %r" % (e.type,))
        s += 1
    for t in raw.tendons:
        if not t.limited:
            continue
        if len(t.joints) > 2:
            raise NotImplementedError("tendon limits over more than two joints are not compiled for the kernel")
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        ext = f["pext"][s * PEXT_STRIDE:(s + 1) * PEXT_STRIDE]
        dofs = [(raw.dof_of_joint(jn), float(c)) for jn, c in t.joints]
        if any(link_kind[d] not in (LINK_HINGE, LINK_SLIDE) for d, _ in dofs):
            raise NotImplementedError("fixed tendons run over hinge / slide joints")
        dofs.sort(key=lambda dc: -dc[0])                    # the higher dof first: the other lies on its elimination path
        assert len(dofs) == 1 or e_is_ancestor(dofs[1][0], dofs[0][0])
        Jt = np.zeros(nv)
        for d, c in dofs:
            Jt[d] += c
        rec[0], rec[13] = dofs[0][0], (dofs[1][0] if len(dofs) == 2 else -1)
        rec[5] = t.margin
        rec[6] = float(Jt @ M0inv @ Jt) if base is None else base.tendon_invweight0[t.name]
        rec[11] = edepth[dofs[0][0]] - 1
        rec[12] = PT_DOFROW
        ext[0] = 1.0
        ext[1], ext[2] = dofs[0][1], (dofs[1][1] if len(dofs) == 2 else 0.0)
        shift = sum(c * raw.bodies[link_body[d]].joint.ref for d, c in dofs)      # (length on qpos = on the kernel's coordinates + this)
        ext[3:5] = t.range[0] - shift, t.range[1] - shift
        ext[5] = t.margin
        rec[21] = sol_class(lim_default[0] if t.solref_limit is None else t.solref_limit,
                            lim_default[1] if t.solimp_limit is None else t.solimp_limit)
        gen = True
        s += 1
    # ball-joint limits (mj_instantiateLimit, mjJNT_BALL): one row over the joint's three dofs, J = -axis of the joint
    # quaternion's rotation; a dof-row record anchored at the joint's LAST link (its first two lie on its elimination path)
    for b in raw.bodies:
        jt = b.joint
        if jt is None or jt.type != JOINT_BALL or not jt.limited:
            continue
        d0 = raw.dof_of_joint(jt.name)
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        ext = f["pext"][s * PEXT_STRIDE:(s + 1) * PEXT_STRIDE]
        assert link_kind[d0] == LINK_BALL_X and e_is_ancestor(d0 + 1, d0 + 2) and e_is_ancestor(d0, d0 + 2)
        rec[0], rec[13] = d0 + 2, d0 + 1
        rec[6] = dof_iw[d0] if base is None else base.dof_invweight0[d0]
        rec[11] = edepth[d0 + 2] - 1
        rec[12] = PT_DOFROW
        ext[0] = 2.0
        ext[1] = d0                                         # the link that holds the quaternion
        ext[3:5] = 0.0, max(jt.range)
        rec[21] = sol_class(lim_default[0] if jt.solref_limit is None else jt.solref_limit,
                            lim_default[1] if jt.solimp_limit is None else jt.solimp_limit)
        gen = True
        s += 1
    nsp = s
    f["n_sphere"][0] = nsp
    gen = gen or bool(np.any(f["frictionloss"] > 0))
    # (joint ref and joint margin are read by the general instantiations only: a model that has nothing else of theirs still needs
    # them - found by the round-5 soak, seed 8666: a three-hinge chain with two refs, simulated in the other kernels with ref = 0)
    gen = gen or bool(np.any(f["qoff"] != 0)) or bool(np.any(f["jmargin"] != 0))
    kinds_used = f["spheres"].reshape(TREE_MAX_SPHERES, SPH_STRIDE)[:nsp]
    gen2 = bool(np.any(kinds_used[:, 12] >= PT_PLANE_CYL) or np.any(kinds_used[:, 23] == 8.0))
    # elliptic friction cones (round 5, MJCF <option cone="elliptic" impratio>): the contact records with friction carry
    # impratio in their extension [21] and the model takes the GEN = 3 instantiations
    gen3 = False
    if getattr(raw, "cone", "pyramidal") == "elliptic":
        for k in range(nsp):
            if int(kinds_used[k, 12]) not in (PT_CONNECT, PT_DOFROW, PT_WELD) and kinds_used[k, 7] > 0:
                f["pext"][k * PEXT_STRIDE + 21] = max(float(raw.impratio), 1e-15)
                gen3 = True
    f["gen"][0] = 3.0 if gen3 else (2.0 if gen2 else (1.0 if gen else 0.0))
    f["nq"][0] = nq
    f["has_ball"][0] = 1.0 if any(k == LINK_BALL_X for k in link_kind) else 0.0
    f["any_friction"][0] = 1.0 if (any(f["spheres"][k * SPH_STRIDE + 7] > 0 for k in range(nsp)) or pair_geoms
                                   or any(f["kpg"] != 0) or any(f["kvg"] != 0) or any(f["tau0"] != 0) or bool(np.any(np.isfinite(f["tau_lo"]))) or bool(np.any(f["tpartner"] >= 0)) or bool(np.any(f["tcoef"] != 1.0)) or bool(np.any(np.isfinite(f["tau_hi"]))) or raw.task == TASK_ORIENT or gen) else 0.0

    f["gravity"][:] = raw.gravity
    f["nu"][0], f["task"][0], f["ctrl_cost"][0], f["obs_skip"][0] = nu, raw.task, raw.ctrl_cost, raw.obs_skip
    f["density"][0], f["viscosity"][0] = raw.density, raw.viscosity
    if raw.task not in (TASK_REACH, TASK_FORWARD, TASK_ORIENT) or not 0 <= raw.obs_skip < nv:
        raise ValueError("unknown task / observation layout")
    f["site_axis"][:] = R0[sb] @ np.asarray(raw.site_axis, float)
    f["target_dir"][:] = raw.target_dir
    d_obs = nq + nv - raw.obs_skip if raw.task == TASK_FORWARD else nq + nv + 6
    tendon_iw = {}
    for t in raw.tendons:
        Jt = np.zeros(nv)
        for jn, c in t.joints:
            Jt[raw.dof_of_joint(jn)] += float(c)
        tendon_iw[t.name] = float(Jt @ M0inv @ Jt)
    if base is not None:            # run-time edit: MuJoCo keeps the constants mj_setConst computed at load time
        f["dof_invweight0"][:] = base.field("dof_invweight0")
        dof_iw, body_iw = base.dof_invweight0.copy(), base.body_invweight0.copy()
        tendon_iw = dict(base.tendon_invweight0)
    blob = np.concatenate([f[name] for name, _ in TREE_LAYOUT]).astype(np.float64)
    assert blob.size == TREE_BLOB_LEN
    tm = TreeModel(blob=blob, nv=nv, nu=nu, d_obs=d_obs, timestep=raw.timestep, frame_skip=raw.frame_skip,
                   target_default=np.asarray(raw.target_pos, float), ctrl_lo=ctrl_lo, ctrl_hi=ctrl_hi, parent=parent,
                   task=int(raw.task), obs_skip=int(raw.obs_skip), max_path=int(edepth.max()),
                   body_mass=mass, body_inertia=inert, body_invweight0=body_iw, dof_invweight0=dof_iw, link_of_body=link_of_body,
                   nq=nq, qpos0=raw.qpos0)
    tm.tendon_invweight0 = tendon_iw
    tm.body_invweight0_rot = base.body_invweight0_rot.copy() if base is not None else body_iw_rot
    tm.general = bool(gen)
    return tm

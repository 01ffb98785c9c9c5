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
from .raw import GEOM_SPHERE, JOINT_HINGE, JOINT_SLIDE, TASK_FORWARD, TASK_ORIENT, TASK_REACH, RawModel

TL = 32                     # lanes per particle
TREE_MAX_SPHERES = 16
SPH_STRIDE = 24             # contact record: [0] link A, [1:4] centre / segment start on A, [4] radius, [5] margin,
                            # [6] invweight (both bodies), [7] mu, [8:11] capsule axis (plane contacts: frame hint) /
                            # segment vector on A (geom-geom), [11] depth of link A in the elimination tree, [12] kind
                            # (0 sphere-plane, 1 geom-geom), [13] link B, [14:17] segment start on B, [17] radius B,
                            # [18:21] segment vector on B
MJ_MINIMP, MJ_MAXIMP = 1e-4, 0.9999     # MuJoCo's clamp on solimp (getsolparam)

TREE_LAYOUT = [
    # ---- staged in LDS by the kernel: per-link constants, scalars, contact records
    ("off", 3 * TL), ("axis", 3 * TL), ("mass", TL), ("com", 3 * TL), ("inertia", 6 * TL),
    ("armature", TL), ("damping", TL), ("range_lo", TL), ("range_hi", TL), ("limited", TL), ("gear", TL),
    ("ctrl_lo", TL), ("ctrl_hi", TL), ("dof_invweight0", TL),
    ("stiffness", TL), ("springref", TL), ("fbox", 3 * TL), ("frot", 9 * TL),
    ("kpg", TL),                    # position servos: gear^2 kp, the stiffness of their bias -kp * (gear q) at the joint
    ("nv", 1), ("timestep", 1), ("frame_skip", 1), ("jumps", 1), ("site_link", 1), ("site_pos", 3),
    ("n_sphere", 1), ("plane_n", 3), ("plane_d", 1),
    ("sol_K", 1), ("sol_B", 1), ("sol_dmin", 1), ("sol_dmax", 1), ("sol_width", 1), ("sol_mid", 1), ("sol_power", 1),
    ("gravity", 3),
    ("nu", 1), ("task", 1), ("ctrl_cost", 1), ("obs_skip", 1), ("density", 1), ("viscosity", 1),
    ("lsol_K", 1), ("lsol_B", 1), ("lsol_dmin", 1), ("lsol_dmax", 1), ("lsol_width", 1), ("lsol_mid", 1), ("lsol_power", 1),
    ("any_friction", 1),
    ("site_axis", 3), ("target_dir", 3),    # TASK_ORIENT: object axis in the site link's frame, its target direction
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
]
TREE_BLOB_LEN = sum(n for _, n in TREE_LAYOUT)
TREE_STATE_LEN = 2 * TL + 6           # qpos[32] | qvel[32] | target_pos[3] | fresh site[3]


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


def compile_tree(raw: RawModel, overrides=None, base: "TreeModel" = None) -> TreeModel:
    """``overrides`` mimics run-time edits of the compiled MuJoCo model (dynamics randomization,
    mjmpc/envs/gym_env_wrapper.py:367-416), as ``compile.compile_arm`` does: ``{"body_mass": {body: m}, "body_inertia":
    {body: [I1, I2, I3]}, "dof_damping": {joint: d}, "geom_size": {geom: [r, half, ..]}, "geom_friction": {geom: [mu, ..]}}``.
    Like MuJoCo, such edits do NOT recompute the qpos0 constants: pass the unperturbed model as ``base`` and its
    dof / body invweight0 are kept."""
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

    # ---- links: one per joint, welded bodies merged into the link that carries them ----------------------
    jointed = [i for i, b in enumerate(raw.bodies) if b.joint is not None]
    nv = len(jointed)
    if not 1 <= nv <= TL:
        raise ValueError("tree kernel supports 1..%d dofs, got %d" % (TL, nv))
    jtype = [raw.bodies[i].joint.type for i in jointed]
    if any(t not in (JOINT_HINGE, JOINT_SLIDE) for t in jtype):
        raise ValueError("joints must be hinges or slides")
    link_of_body, parent = [-1] * nb, []
    for i, b in enumerate(raw.bodies):
        pl = -1 if b.parent < 0 else link_of_body[b.parent]
        if b.joint is not None:
            link_of_body[i] = jointed.index(i)
            parent.append(pl)
        else:
            if pl < 0:
                raise ValueError("static body %s before the first joint is not supported" % b.name)
            link_of_body[i] = pl
    parent = np.array(parent, int)
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
    # H = M + J' D J has M's pattern (entries only between a link and its ancestors) as long as every constraint row
    # touches the dofs of ONE root path.  A contact between a manipulator link and an object link touches two trees.
    # Hanging the manipulator's root under the object's last link - in the tree the FACTORISATION walks, not in the
    # kinematic one - puts both on one path again: the object's dofs are eliminated last, fill-in stays inside the
    # path-indexed rows.  (The object must be a serial chain listed before the manipulator.)
    geom_names = {g.name: (i, g) for i, b in enumerate(raw.bodies) for g in b.geoms if g.name}
    pair_geoms = []
    for ga, gb in raw.pairs:
        if ga not in geom_names or gb not in geom_names:
            raise ValueError("collision pair names an unknown geom: %r / %r" % (ga, gb))
        pair_geoms.append((geom_names[ga], geom_names[gb]))

    def root_of(k):
        while parent[k] >= 0:
            k = parent[k]
        return k

    def is_ancestor(a, k):          # link a is k or one of k's (kinematic) ancestors
        while k >= 0:
            if k == a:
                return True
            k = parent[k]
        return False

    eparent = parent.copy()
    # pairs inside ONE kinematic tree (self-collision: swimmer.xml's segments) need no help as long as one link is an
    # ancestor of the other - the row then lives on the deeper link's path; listed deeper geom first
    cross = []
    for (ia, ga), (ib, gb) in pair_geoms:
        la, lb = link_of_body[ia], link_of_body[ib]
        if la < 0 or lb < 0:
            raise NotImplementedError("geom-geom pairs must name geoms on moving bodies")
        if root_of(la) != root_of(lb):
            cross.append(((ia, ga), (ib, gb)))
        elif not is_ancestor(lb, la):
            raise NotImplementedError("a self-collision pair must list the deeper geom first, and its links must lie on one "
                                      "root path (a pair across two branches would fill the sparse factorisation in): "
                                      "%r / %r" % (ga.name, gb.name))
    if cross:
        obj_roots = {root_of(link_of_body[ib]) for (_, _), (ib, _) in cross}
        man_roots = {root_of(link_of_body[ia]) for (ia, _), (_, _) in cross}
        if len(obj_roots) != 1 or obj_roots & man_roots:
            raise NotImplementedError("geom-geom pairs must pair geoms of the manipulator(s) with geoms of ONE object tree")
        ro = obj_roots.pop()
        chain = list(range(ro, ro + subsize[ro]))
        if any(parent[k] != k - 1 for k in chain[1:]):
            raise NotImplementedError("the object of geom-geom pairs must be a serial chain of joints")
        for rm in sorted(man_roots):
            if rm < ro:
                raise NotImplementedError("list the object before the manipulator (its dofs are eliminated last)")
            eparent[rm] = chain[-1]
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
    origin = [p0[j] for j in jointed]
    axis_w = []
    for li, bj in enumerate(jointed):
        jt = raw.bodies[bj].joint
        prev = origin[parent[li]] if parent[li] >= 0 else np.zeros(3)
        axis = R0[bj] @ (np.asarray(jt.axis, float) / np.linalg.norm(jt.axis))
        axis_w.append(axis)
        members = [i for i in range(nb) if link_of_body[i] == li]
        m = mass[members].sum()
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
        f["jtype"][li] = jt.type
        f["stiffness"][li], f["springref"][li] = jt.stiffness, jt.springref
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
        f["range_lo"][li], f["range_hi"][li] = jt.range
        f["limited"][li] = 1.0 if jt.limited else 0.0
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
    for a, act in enumerate(raw.actuators):         # action a drives the dof of its joint (any subset, any order)
        d = raw.dof_of_joint(act.joint)
        if f["act"][d] >= 0:
            raise ValueError("two motors on joint %r" % act.joint)
        f["act"][d] = a
        # motor: gear * clip(ctrl).  Position servo (MJCF <position kp>): gear * kp * (clip(ctrl) - gear * q) - the ctrl
        # part through an effective gear, the bias as a joint stiffness gear^2 kp about 0
        f["gear"][d] = act.gear * (act.kp if act.kp > 0 else 1.0)
        f["kpg"][d] = act.gear * act.gear * act.kp
        f["ctrl_lo"][d], f["ctrl_hi"][d] = act.ctrlrange
        ctrl_lo[a], ctrl_hi[a] = act.ctrlrange

    # ---- constants at qpos0 (MuJoCo mj_setConst): M0, dof / body invweight0 ------------------------------
    def on_path(k, link):
        return bool((ancmask[link] >> k) & 1)

    def jac_point(pt, link):
        J = np.zeros((3, nv))
        for k in range(nv):
            if on_path(k, link):
                J[:, k] = axis_w[k] if jtype[k] == JOINT_SLIDE else np.cross(axis_w[k], pt - origin[k])
        return J

    M0 = np.diag(f["armature"][:nv]).astype(float)
    for i in range(nb):
        if mass[i] <= 0:
            continue
        li = link_of_body[i]
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], li)
        Jr = np.zeros((3, nv))
        for k in range(nv):
            if on_path(k, li) and jtype[k] == JOINT_HINGE:
                Jr[:, k] = axis_w[k]
        Iw = R0[i] @ inert[i] @ R0[i].T
        M0 += mass[i] * Jp.T @ Jp + Jr.T @ Iw @ Jr
    M0inv = np.linalg.inv(M0)
    dof_iw = np.diag(M0inv).copy()
    f["dof_invweight0"][:nv] = dof_iw
    body_iw = np.zeros(nb)
    for i in range(nb):
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], link_of_body[i])
        body_iw[i] = np.trace(Jp @ M0inv @ Jp.T) / 3.0

    # ---- scalars, site, contacts --------------------------------------------------------------------------
    f["nv"][0], f["timestep"][0], f["frame_skip"][0], f["jumps"][0] = nv, raw.timestep, raw.frame_skip, jumps
    sb = raw.site_body
    f["site_link"][0] = link_of_body[sb]
    f["site_pos"][:] = p0[sb] + R0[sb] @ np.asarray(raw.site_pos, float) - origin[link_of_body[sb]]
    # contact points: a colliding sphere, or the two end spheres of a colliding capsule - the "to" end first, as
    # MuJoCo's mjc_PlaneCapsule tests them - which also hand their axis to the contact frame.  Friction and condim of
    # a contact are the larger of the two geoms' (mj_contactParam with equal priorities).
    points = []
    for i, b in enumerate(raw.bodies):
        for g in b.geoms:
            if not g.collide:
                continue
            size = overrides.get("geom_size", {}).get(g.name)       # run-time edit: [radius, half-length, -]
            radius = float(np.ravel(size)[0]) if size is not None else g.radius
            if g.type == GEOM_SPHERE:
                points.append((i, g, np.asarray(g.a, float), np.zeros(3), radius))
            else:
                a, e = np.asarray(g.a, float), np.asarray(g.b, float)
                u = (e - a) / np.linalg.norm(e - a)
                if size is not None:                                # the ends move with the half-length, the centre stays
                    c, half = 0.5 * (a + e), float(np.ravel(size)[1])
                    a, e = c - half * u, c + half * u
                points += [(i, g, e, u, radius), (i, g, a, u, radius)]
    if raw.plane is None:
        points = []
    if len(points) + len(pair_geoms) > TREE_MAX_SPHERES:
        raise ValueError("tree kernel supports %d contact points (a capsule on the plane counts two, a geom-geom pair one)"
                         % TREE_MAX_SPHERES)
    if raw.plane is not None:
        n = np.asarray(raw.plane.normal, float)
        n = n / np.linalg.norm(n)
        f["plane_n"][:] = n
        f["plane_d"][0] = n @ np.asarray(raw.plane.pos, float)
    f["n_sphere"][0] = len(points) + len(pair_geoms)

    def geom_mu(g):
        mu_g = overrides.get("geom_friction", {}).get(g.name)
        return float(np.ravel(mu_g)[0]) if mu_g is not None else g.friction

    def body_w(i):
        return base.body_invweight0[i] if base is not None else body_iw[i]

    for s, (i, g, pos, u, radius) in enumerate(points):
        li = link_of_body[i]
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        rec[0] = li
        rec[1:4] = p0[i] + R0[i] @ pos - origin[li]
        rec[4] = radius
        rec[5] = max(raw.plane.margin, g.margin)            # MuJoCo: max of the two geom margins
        rec[6] = 0.0 + body_w(i)                            # the world body weighs 0
        condim = max(int(g.condim), int(raw.plane.condim))
        if condim not in (1, 3):
            raise NotImplementedError("contacts are condim 1 (frictionless) or 3 (pyramidal cone), got %d" % condim)
        rec[7] = max(geom_mu(g), raw.plane.friction) if condim == 3 else 0.0
        rec[8:11] = R0[i] @ u
        rec[11] = edepth[li] - 1                            # strict ancestors of the point's link (elimination tree)
        rec[13] = -1.0
    # geom-geom pairs: both geoms as segments (start, vector; a sphere has a zero vector) in their links' frames; the
    # record is anchored at the manipulator's link, whose elimination path contains the object's links
    for s, ((ia, ga), (ib, gb)) in enumerate(pair_geoms, start=len(points)):
        rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
        la, lb = link_of_body[ia], link_of_body[ib]

        def seg(i, g, li):
            size = overrides.get("geom_size", {}).get(g.name)
            r = float(np.ravel(size)[0]) if size is not None else g.radius
            a = np.asarray(g.a, float)
            if g.type == GEOM_SPHERE:
                return p0[i] + R0[i] @ a - origin[li], np.zeros(3), r
            e = np.asarray(g.b, float)
            if size is not None:
                u, c, half = (e - a) / np.linalg.norm(e - a), 0.5 * (a + e), float(np.ravel(size)[1])
                a, e = c - half * u, c + half * u
            return p0[i] + R0[i] @ a - origin[li], R0[i] @ (e - a), r

        a0, da, ra = seg(ia, ga, la)
        b0, db, rb = seg(ib, gb, lb)
        condim = max(int(ga.condim), int(gb.condim))
        if condim not in (1, 3):
            raise NotImplementedError("contacts are condim 1 (frictionless) or 3 (pyramidal cone), got %d" % condim)
        rec[0], rec[1:4], rec[4] = la, a0, ra
        rec[5] = max(ga.margin, gb.margin)
        rec[6] = body_w(ia) + body_w(ib)
        rec[7] = max(geom_mu(ga), geom_mu(gb)) if condim == 3 else 0.0
        rec[8:11] = da
        rec[11] = edepth[la] - 1
        rec[12] = 1.0
        rec[13], rec[14:17], rec[17], rec[18:21] = lb, b0, rb, db
    nsp = len(points) + len(pair_geoms)
    f["any_friction"][0] = 1.0 if (any(f["spheres"][s * SPH_STRIDE + 7] > 0 for s in range(nsp)) or pair_geoms
                                   or any(f["kpg"] != 0) or raw.task == TASK_ORIENT) else 0.0

    def sol_set(prefix, solref, solimp):
        tc, dr = solref
        if tc <= 0 or dr <= 0:
            raise NotImplementedError("solref must be the standard (timeconst, dampratio) pair")
        tc = max(tc, 2 * raw.timestep)                              # refsafe
        dmin, dmax, width, mid, power = solimp
        dmin, dmax = np.clip(dmin, MJ_MINIMP, MJ_MAXIMP), np.clip(dmax, MJ_MINIMP, MJ_MAXIMP)
        mid, width, power = np.clip(mid, MJ_MINIMP, MJ_MAXIMP), max(width, 0.0), max(power, 1.0)
        if power != int(power) or power > 64:
            raise NotImplementedError("solimp power must be an integer in [1, 64] (MuJoCo's default is 2), got %r" % (power,))
        f[prefix + "_K"][0] = 1.0 / (dmax * dmax * tc * tc * dr * dr)
        f[prefix + "_B"][0] = 2.0 / (dmax * tc)
        if width <= 1e-15:                                          # MuJoCo getimpedance: a flat impedance
            dmin = dmax = 0.5 * (dmin + dmax)
            width = 1.0
        f[prefix + "_dmin"][0], f[prefix + "_dmax"][0] = dmin, dmax
        f[prefix + "_width"][0], f[prefix + "_mid"][0], f[prefix + "_power"][0] = width, mid, power

    sol_set("sol", raw.solref, raw.solimp)
    sol_set("lsol", raw.solref if raw.solref_limit is None else raw.solref_limit,
            raw.solimp if raw.solimp_limit is None else raw.solimp_limit)
    f["gravity"][:] = raw.gravity
    f["nu"][0], f["task"][0], f["ctrl_cost"][0], f["obs_skip"][0] = nu, raw.task, raw.ctrl_cost, raw.obs_skip
    f["density"][0], f["viscosity"][0] = raw.density, raw.viscosity
    if raw.task not in (TASK_REACH, TASK_FORWARD, TASK_ORIENT) or not 0 <= raw.obs_skip < nv:
        raise ValueError("unknown task / observation layout")
    f["site_axis"][:] = R0[sb] @ np.asarray(raw.site_axis, float)
    f["target_dir"][:] = raw.target_dir
    d_obs = 2 * nv - raw.obs_skip if raw.task == TASK_FORWARD else 2 * nv + 6
    if base is not None:            # run-time edit: MuJoCo keeps the constants mj_setConst computed at load time
        f["dof_invweight0"][:] = base.field("dof_invweight0")
        dof_iw, body_iw = base.dof_invweight0.copy(), base.body_invweight0.copy()
    blob = np.concatenate([f[name] for name, _ in TREE_LAYOUT]).astype(np.float64)
    assert blob.size == TREE_BLOB_LEN
    return TreeModel(blob=blob, nv=nv, nu=nu, d_obs=d_obs, timestep=raw.timestep, frame_skip=raw.frame_skip,
                     target_default=np.asarray(raw.target_pos, float), ctrl_lo=ctrl_lo, ctrl_hi=ctrl_hi, parent=parent,
                     task=int(raw.task), obs_skip=int(raw.obs_skip), max_path=int(edepth.max()),
                     body_mass=mass, body_inertia=inert, body_invweight0=body_iw, dof_invweight0=dof_iw, link_of_body=link_of_body)

"""Host-side model compiler for kinematic TREES of hinge links: RawModel -> the constant block of the tree kernel
(``mjmpc_amd/csrc/tree_rollout.hip``, layout mirrored by ``csrc/tree_model.h``).

Same conventions as ``compile.compile_arm`` (one link per hinge, welded bodies merged into the link that carries
them, link frames world-aligned at qpos0, MuJoCo's inertiafromgeom and ``mj_setConst`` constants), without the
serial-chain restriction: a link may carry several child links.  The kernel wants the links numbered depth-first, so
that a subtree is a contiguous index range; an MJCF file lists its bodies that way, and a model that does not is
rejected.  Up to 32 dofs, one motor per hinge in joint order, up to 8 collision spheres against one plane.
"""
from dataclasses import dataclass

import numpy as np

from .compile import _geom_inertial, _quat2mat, _shift_inertia
from .raw import GEOM_SPHERE, RawModel

TL = 32                     # lanes per particle
TREE_MAX_SPHERES = 8
SPH_STRIDE = 8

TREE_LAYOUT = [
    ("off", 3 * TL), ("axis", 3 * TL), ("mass", TL), ("com", 3 * TL), ("inertia", 6 * TL),
    ("armature", TL), ("damping", TL), ("range_lo", TL), ("range_hi", TL), ("limited", TL), ("gear", TL),
    ("ctrl_lo", TL), ("ctrl_hi", TL), ("dof_invweight0", TL),
    ("parent", TL), ("subsize", TL), ("anc", 5 * TL), ("ancmask", 2 * TL),
    ("nv", 1), ("timestep", 1), ("frame_skip", 1), ("jumps", 1), ("site_link", 1), ("site_pos", 3),
    ("n_sphere", 1), ("plane_n", 3), ("plane_d", 1),
    ("sol_K", 1), ("sol_B", 1), ("sol_dmin", 1), ("sol_dmax", 1), ("sol_width", 1), ("sol_mid", 1), ("sol_power", 1),
    ("gravity", 3), ("spheres", TREE_MAX_SPHERES * SPH_STRIDE),
    # tree-sparse L'DL (MuJoCo's factorisation order: leaves first, no fill-in): links of equal HEIGHT above their
    # deepest leaf are mutually unrelated and are eliminated together, one round per height
    ("depth", TL),                  # strict ancestors of the link
    ("n_rounds", 1),                # max height + 1
    ("elim", (TL - 1) * TL),        # [entry][lane]: my descendants sorted by height, packed k | dist << 8 | height << 16
                                    # (-1 terminates): the rows that update mine, round by round
]
TREE_BLOB_LEN = sum(n for _, n in TREE_LAYOUT)
TREE_STATE_LEN = 2 * TL + 3           # qpos[32] | qvel[32] | target_pos[3]


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
    parent: np.ndarray             # parent link of every link (-1: root)
    max_path: int                  # links on the longest root-to-leaf path
    body_mass: np.ndarray
    body_invweight0: np.ndarray
    dof_invweight0: np.ndarray
    link_of_body: list

    def field(self, name):
        o, n = TREE_OFFSETS[name]
        return self.blob[o:o + n]


def compile_tree(raw: RawModel) -> TreeModel:
    nb = len(raw.bodies)
    R0, p0 = [None] * nb, [None] * nb
    mass, ipos, inert = np.zeros(nb), np.zeros((nb, 3)), np.zeros((nb, 3, 3))
    for i, b in enumerate(raw.bodies):
        if b.parent >= i:
            raise ValueError("bodies must be listed parents-first")
        Rp, pp = (np.eye(3), np.zeros(3)) if b.parent < 0 else (R0[b.parent], p0[b.parent])
        R0[i] = Rp @ _quat2mat(b.quat)
        p0[i] = pp + Rp @ np.asarray(b.pos, float)
        parts = [_geom_inertial(g) for g in b.geoms]
        mass[i] = sum(m for m, _, _ in parts)
        if mass[i] > 0:
            ipos[i] = sum(m * c for m, c, _ in parts) / mass[i]
            inert[i] = sum(_shift_inertia(I, m, c - ipos[i]) for m, c, I in parts)

    # ---- links: one per hinge, welded bodies merged into the link that carries them ----------------------
    jointed = [i for i, b in enumerate(raw.bodies) if b.joint is not None]
    nv = len(jointed)
    if not 1 <= nv <= TL:
        raise ValueError("tree kernel supports 1..%d hinge dofs, got %d" % (TL, nv))
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
        f["damping"][li] = jt.damping
        f["range_lo"][li], f["range_hi"][li] = jt.range
        f["limited"][li] = 1.0 if jt.limited else 0.0
    f["axis"][2 * TL + nv:3 * TL] = 1.0         # spare lanes: a unit axis keeps their (unused) rotation orthonormal
    f["parent"][:] = -1.0
    f["parent"][:nv] = parent
    f["subsize"][:nv] = subsize
    f["anc"][:] = -1.0
    for e in range(5):
        f["anc"][e * TL:e * TL + nv] = anc[e]
    f["ancmask"][:nv] = ancmask & 0xFFFF
    f["ancmask"][TL:TL + nv] = ancmask >> 16
    f["depth"][:nv] = depth - 1
    f["n_rounds"][0] = height.max() + 1
    f["elim"][:] = -1.0
    for i in range(nv):
        desc = sorted(range(i + 1, i + subsize[i]), key=lambda k: (height[k], k))
        for e, k in enumerate(desc):
            f["elim"][e * TL + i] = k | ((depth[k] - depth[i]) << 8) | (height[k] << 16)

    if len(raw.actuators) != nv:
        raise ValueError("tree kernel expects one motor per hinge")
    ctrl_lo, ctrl_hi = np.zeros(nv), np.zeros(nv)
    for a, act in enumerate(raw.actuators):
        if raw.dof_of_joint(act.joint) != a:
            raise ValueError("motors must be listed in joint order")
        f["gear"][a] = act.gear
        f["ctrl_lo"][a], f["ctrl_hi"][a] = act.ctrlrange
        ctrl_lo[a], ctrl_hi[a] = act.ctrlrange

    # ---- constants at qpos0 (MuJoCo mj_setConst): M0, dof / body invweight0 ------------------------------
    def on_path(k, link):
        return bool((ancmask[link] >> k) & 1)

    def jac_point(pt, link):
        J = np.zeros((3, nv))
        for k in range(nv):
            if on_path(k, link):
                J[:, k] = np.cross(axis_w[k], pt - origin[k])
        return J

    M0 = np.diag(f["armature"][:nv]).astype(float)
    for i in range(nb):
        if mass[i] <= 0:
            continue
        li = link_of_body[i]
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], li)
        Jr = np.zeros((3, nv))
        for k in range(nv):
            if on_path(k, li):
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
    spheres = [(i, g) for i, b in enumerate(raw.bodies) for g in b.geoms if g.collide]
    if raw.plane is not None and spheres:
        if len(spheres) > TREE_MAX_SPHERES:
            raise ValueError("tree kernel supports %d collision spheres" % TREE_MAX_SPHERES)
        n = np.asarray(raw.plane.normal, float)
        n = n / np.linalg.norm(n)
        f["plane_n"][:] = n
        f["plane_d"][0] = n @ np.asarray(raw.plane.pos, float)
        f["n_sphere"][0] = len(spheres)
        for s, (i, g) in enumerate(spheres):
            if g.type != GEOM_SPHERE:
                raise ValueError("only sphere-plane contacts are supported")
            li = link_of_body[i]
            rec = f["spheres"][s * SPH_STRIDE:(s + 1) * SPH_STRIDE]
            rec[0] = li
            rec[1:4] = p0[i] + R0[i] @ np.asarray(g.a, float) - origin[li]
            rec[4] = g.radius
            rec[5] = max(raw.plane.margin, g.margin)            # MuJoCo: max of the two geom margins
            rec[6] = 0.0 + body_iw[i]                           # the world body weighs 0
    tc, dr = raw.solref
    tc = max(tc, 2 * raw.timestep)                              # refsafe
    dmin, dmax, width, mid, power = raw.solimp
    if power < 1 or power != int(power) or power > 64:
        raise NotImplementedError("solimp power must be an integer in [1, 64] (MuJoCo's default is 2), got %r" % (power,))
    f["sol_K"][0] = 1.0 / (dmax * dmax * tc * tc * dr * dr)
    f["sol_B"][0] = 2.0 / (dmax * tc)
    f["sol_dmin"][0], f["sol_dmax"][0] = dmin, dmax
    f["sol_width"][0], f["sol_mid"][0], f["sol_power"][0] = width, mid, power
    f["gravity"][:] = raw.gravity
    blob = np.concatenate([f[name] for name, _ in TREE_LAYOUT]).astype(np.float64)
    assert blob.size == TREE_BLOB_LEN
    return TreeModel(blob=blob, nv=nv, nu=nv, d_obs=2 * nv + 6, timestep=raw.timestep, frame_skip=raw.frame_skip,
                     target_default=np.asarray(raw.target_pos, float), ctrl_lo=ctrl_lo, ctrl_hi=ctrl_hi, parent=parent,
                     max_path=int(depth.max()),
                     body_mass=mass, body_invweight0=body_iw, dof_invweight0=dof_iw, link_of_body=link_of_body)

"""Host-side model compiler: RawModel -> the flat constant block the HIP arm kernel consumes.

What MuJoCo's XML compiler + ``mj_setConst`` do for ``sawyer.xml`` (inertiafromgeom, welded-body
bookkeeping, ``dof_invweight0`` / ``body_invweight0``) is done here once, in float64 numpy, and
laid out for the kernel's execution model: a serial chain of <= 7 hinge links, ONE LANE PER LINK
inside an 8-lane group (lane 7 carries the contact row).  Bodies without a joint are merged into
the link that carries them, and every link frame is chosen to be world-aligned at qpos0, so the
kernel never multiplies by a fixed body rotation.

The layout of the block (``ARM_LAYOUT``) is mirrored by ``struct ArmConst`` in
``mjmpc_amd/csrc/arm_model.h`` and documented in ``include/mjmpc_amd.h``.
"""
from dataclasses import dataclass

import numpy as np

from .raw import GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, RawModel

LANES = 8          # lanes per particle in the HIP kernel
MAX_LINKS = 7

# (name, length) in float64 slots; per-link fields are stored [component][lane]
ARM_LAYOUT = [
    ("off", 3 * LANES), ("axis", 3 * LANES), ("mass", LANES), ("com", 3 * LANES),
    ("inertia", 6 * LANES),                    # xx, yy, zz, xy, xz, yz about the link COM
    ("armature", LANES), ("damping", LANES), ("range_lo", LANES), ("range_hi", LANES),
    ("limited", LANES), ("gear", LANES), ("ctrl_lo", LANES), ("ctrl_hi", LANES),
    ("dof_invweight0", LANES),
    ("nv", 1), ("timestep", 1), ("frame_skip", 1),
    ("site_link", 1), ("site_pos", 3),
    ("n_sphere", 1), ("sph_link", 1), ("sph_pos", 3), ("sph_r", 1), ("sph_margin", 1),
    ("sph_invweight", 1), ("plane_n", 3), ("plane_d", 1),
    ("sol_K", 1), ("sol_B", 1), ("sol_dmin", 1), ("sol_dmax", 1), ("sol_width", 1),
    ("sol_mid", 1), ("sol_power", 1), ("gravity", 3),
    # round 6 (the EXTENDED-JOINT instantiation of the arm kernels, csrc/arm_rollout_xj.hip): slide joints and dry friction -
    # the reference's classic-control models (examples/configs/classic_control/cartpole*.yml) on the serial-chain kernel
    ("jtype", LANES),                           # 0 hinge, 1 slide
    ("frictionloss", LANES), ("floss_D", LANES),    # dof_frictionloss and its row's D = 1 / R (impedance at position 0)
    ("floss_B", 1),                             # b of the friction rows' reference acceleration -b v (solref_friction)
    ("nu", 1),                                  # motors: they drive dofs 0 .. nu - 1 (until round 6 nu = nv by construction)
]
ARM_BLOB_LEN = sum(n for _, n in ARM_LAYOUT)
MJ_MINVAL = 1e-15


def _offsets():
    o, out = 0, {}
    for name, n in ARM_LAYOUT:
        out[name] = (o, n)
        o += n
    return out


ARM_OFFSETS = _offsets()


def _quat2mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _geom_inertial(g, cap=1.0):
    """(mass, centre, inertia about centre) of one geom, MuJoCo inertiafromgeom conventions; ``cap``: the capsule end
    caps' volume in units of pi r^3 (MuJoCo 2.0: 1, later versions 4/3 - raw.MJ20_CAPSULE_CAP)."""
    r = float(g.radius)
    if g.type == GEOM_SPHERE:
        m = g.density * 4.0 / 3.0 * np.pi * r ** 3
        return m, np.asarray(g.a, float), np.eye(3) * (0.4 * m * r * r)
    if g.type == GEOM_CAPSULE:
        a, b = np.asarray(g.a, float), np.asarray(g.b, float)
        h = np.linalg.norm(b - a)
        u = (b - a) / h
        m = g.density * (np.pi * r * r * h + cap * np.pi * r ** 3)
        ms = m * 4 * r / (4 * r + 3 * h)
        mc = m - ms
        i_perp = mc * (3 * r * r + h * h) / 12 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8
        i_ax = mc * r * r / 2 + 0.4 * ms * r * r
        return m, 0.5 * (a + b), i_perp * np.eye(3) + (i_ax - i_perp) * np.outer(u, u)
    if g.type == GEOM_CYLINDER:             # between a and b (flat ends)
        a, b = np.asarray(g.a, float), np.asarray(g.b, float)
        h = np.linalg.norm(b - a)
        u = (b - a) / h
        m = g.density * np.pi * r * r * h
        i_ax, i_perp = 0.5 * m * r * r, m * (3 * r * r + h * h) / 12
        return m, 0.5 * (a + b), i_perp * np.eye(3) + (i_ax - i_perp) * np.outer(u, u)
    if g.type == GEOM_BOX:                  # a = centre, b = half sizes, quat = orientation in the body frame
        hx, hy, hz = (float(x) for x in g.b)
        m = g.density * 8.0 * hx * hy * hz
        R = _quat2mat(g.quat)
        Ib = np.diag([m / 3.0 * (hy * hy + hz * hz), m / 3.0 * (hx * hx + hz * hz), m / 3.0 * (hx * hx + hy * hy)])
        return m, np.asarray(g.a, float), R @ Ib @ R.T
    raise ValueError("unsupported geom type %r" % (g.type,))


def _shift_inertia(I, m, d):
    return I + m * (d @ d * np.eye(3) - np.outer(d, d))


@dataclass
class ArmModel:
    """Compiled arm.  ``blob`` is what crosses the C ABI; the rest is for host code and tests."""
    blob: np.ndarray
    nv: int
    nu: int
    d_obs: int
    timestep: float
    frame_skip: int
    target_default: np.ndarray
    ctrl_lo: np.ndarray
    ctrl_hi: np.ndarray
    body_mass: np.ndarray          # per raw body (world excluded)
    body_ipos: np.ndarray
    body_inertia: np.ndarray
    body_invweight0: np.ndarray
    dof_invweight0: np.ndarray
    link_of_body: list

    def field(self, name):
        o, n = ARM_OFFSETS[name]
        return self.blob[o:o + n]


def principal_inertia(I):
    """Principal moments of a body inertia tensor, descending (MuJoCo's ``body_inertia`` order [EXT])
    and the rotation whose columns are the matching axes."""
    w, V = np.linalg.eigh(np.asarray(I, float))
    return w[::-1].copy(), V[:, ::-1].copy()


def compile_arm(raw: RawModel, overrides=None, base: "ArmModel" = None) -> ArmModel:
    """``overrides`` mimics run-time edits of the compiled MuJoCo model (dynamics randomization,
    mjmpc/envs/gym_env_wrapper.py:367-416): ``{"body_mass": {body: m}, "body_inertia": {body: [I1,I2,I3]},
    "dof_damping": {joint: d}, "geom_size": {geom: [..]}}``.  Like MuJoCo, such edits do NOT recompute the
    qpos0 constants: pass the unperturbed model as ``base`` and its dof/body invweight0 are kept."""
    overrides = overrides or {}
    nb = len(raw.bodies)
    # what only the tree kernel executes (models/compile_tree.py)
    joints = [b.joint for b in raw.bodies if b.joint is not None]
    if any(j.type not in (1, 2) or j.stiffness != 0 or any(np.asarray(j.pos, float) != 0) or j.margin != 0 or j.ref != 0
           for j in joints):
        raise ValueError("arm kernel: hinge / slide joints at the body origin without springs only "
                         "(ball / free joints, springs, joint anchors, margin / ref: the tree engine)")
    if raw.equalities or raw.tendons or raw.world_geoms or any(b.inertial is not None for b in raw.bodies):
        raise ValueError("arm kernel: no equalities, tendons, static geoms or explicit inertials (the tree engine has them)")
    if raw.density > 0 or raw.viscosity > 0 or raw.task != 0:
        raise ValueError("arm kernel: no medium, reach task only (the tree engine runs the locomotion models)")
    if raw.plane is not None and any(g.collide and max(g.condim, raw.plane.condim) > 1 for b in raw.bodies for g in b.geoms):
        raise ValueError("arm kernel: frictionless (condim 1) contacts only (friction cones: the tree engine)")
    if (raw.solref_limit is not None and tuple(raw.solref_limit) != tuple(raw.solref)) or \
            (raw.solimp_limit is not None and tuple(raw.solimp_limit) != tuple(raw.solimp)):
        raise ValueError("arm kernel: joint limits share the contacts' solref / solimp")
    # ---- per-body inertial + pose at qpos0 ------------------------------------------------
    R0 = [None] * nb
    p0 = [None] * nb
    mass = np.zeros(nb)
    ipos = np.zeros((nb, 3))
    inert = np.zeros((nb, 3, 3))
    for i, b in enumerate(raw.bodies):
        Rp, pp = (np.eye(3), np.zeros(3)) if b.parent < 0 else (R0[b.parent], p0[b.parent])
        if b.parent >= i:
            raise ValueError("bodies must be listed parents-first")
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

    # ---- links: one per hinge, welded bodies merged in ------------------------------------
    jointed = [i for i, b in enumerate(raw.bodies) if b.joint is not None]
    nv = len(jointed)
    if not 1 <= nv <= MAX_LINKS:
        raise ValueError("arm kernel supports 1..%d hinge links, got %d" % (MAX_LINKS, nv))
    link_of_body = [-1] * nb
    for i, b in enumerate(raw.bodies):
        if b.joint is not None:
            li = jointed.index(i)
            pl = -1 if b.parent < 0 else link_of_body[b.parent]
            if pl != li - 1:
                raise ValueError("arm kernel needs a serial chain (body %s branches)" % b.name)
            link_of_body[i] = li
        else:
            if b.parent < 0 or link_of_body[b.parent] < 0:
                raise ValueError("static body %s before the first joint is not supported" % b.name)
            link_of_body[i] = link_of_body[b.parent]

    L = LANES
    f = {name: np.zeros(n) for name, n in ARM_LAYOUT}
    origin = [p0[j] for j in jointed]
    for li, bj in enumerate(jointed):
        jt = raw.bodies[bj].joint
        prev = origin[li - 1] if li > 0 else np.zeros(3)
        axis = R0[bj] @ (np.asarray(jt.axis, float) / np.linalg.norm(jt.axis))
        members = [i for i in range(nb) if link_of_body[i] == li]
        m = mass[members].sum()
        com_w = sum(mass[i] * (p0[i] + R0[i] @ ipos[i]) for i in members) / m
        I = sum(_shift_inertia(R0[i] @ inert[i] @ R0[i].T, mass[i], p0[i] + R0[i] @ ipos[i] - com_w)
                for i in members)
        for c in range(3):
            f["off"][c * L + li] = (origin[li] - prev)[c]
            f["axis"][c * L + li] = axis[c]
            f["com"][c * L + li] = (com_w - origin[li])[c]
        f["mass"][li] = m
        for k, (r, c) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            f["inertia"][k * L + li] = I[r, c]
        f["armature"][li] = jt.armature
        f["damping"][li] = float(overrides.get("dof_damping", {}).get(jt.name, jt.damping))
        f["range_lo"][li], f["range_hi"][li] = jt.range
        f["limited"][li] = 1.0 if jt.limited else 0.0
        f["jtype"][li] = 1.0 if jt.type == 2 else 0.0
        f["frictionloss"][li] = float(overrides.get("dof_frictionloss", {}).get(jt.name, jt.frictionloss))

    f["armature"][nv:] = 1.0        # spare lanes: unit diagonal keeps the in-register LDL^T regular

    if not 1 <= len(raw.actuators) <= nv:
        raise ValueError("arm kernel expects motors on the first dofs, in joint order (at most one per dof)")
    ctrl_lo, ctrl_hi = np.zeros(len(raw.actuators)), np.zeros(len(raw.actuators))
    for a, act in enumerate(raw.actuators):
        if act.tendon or act.gain != 1.0 or any(x != 0.0 for x in act.bias) or not act.ctrllimited or act.forcerange is not None:
            raise ValueError("arm kernel: ctrllimited motors only (servos and general actuators run on the tree engine)")
        if raw.dof_of_joint(act.joint) != a:
            raise ValueError("motors must be listed in joint order")
        f["gear"][a] = act.gear
        f["ctrl_lo"][a], f["ctrl_hi"][a] = act.ctrlrange
        ctrl_lo[a], ctrl_hi[a] = act.ctrlrange

    # ---- constants at qpos0 (MuJoCo mj_setConst): M0, dof / body invweight0 ----------------
    def jac_point(pt, link):
        J = np.zeros((3, nv))
        for k in range(link + 1):
            ax = np.array([f["axis"][c * L + k] for c in range(3)])
            J[:, k] = ax if f["jtype"][k] else np.cross(ax, pt - origin[k])
        return J

    M0 = np.diag([f["armature"][k] for k in range(nv)]).astype(float)
    for i in range(nb):
        if mass[i] <= 0:
            continue
        li = link_of_body[i]
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], li)
        Jr = np.zeros((3, nv))
        for k in range(li + 1):
            if not f["jtype"][k]:
                Jr[:, k] = [f["axis"][c * L + k] for c in range(3)]
        Iw = R0[i] @ inert[i] @ R0[i].T
        M0 += mass[i] * Jp.T @ Jp + Jr.T @ Iw @ Jr
    M0inv = np.linalg.inv(M0)
    dof_iw = np.diag(M0inv).copy()
    f["dof_invweight0"][:nv] = dof_iw
    body_iw = np.zeros(nb)
    for i in range(nb):
        Jp = jac_point(p0[i] + R0[i] @ ipos[i], link_of_body[i])
        body_iw[i] = np.trace(Jp @ M0inv @ Jp.T) / 3.0

    # ---- scalars -------------------------------------------------------------------------
    f["nv"][0] = nv
    f["timestep"][0] = raw.timestep
    f["frame_skip"][0] = raw.frame_skip
    sb = raw.site_body
    f["site_link"][0] = link_of_body[sb]
    f["site_pos"][:] = p0[sb] + R0[sb] @ np.asarray(raw.site_pos, float) - origin[link_of_body[sb]]
    spheres = [(i, g) for i, b in enumerate(raw.bodies) for g in b.geoms if g.collide]
    if raw.plane is not None and spheres:
        if len(spheres) > 1:
            raise ValueError("arm kernel supports one collision sphere")
        i, g = spheres[0]
        if g.type != GEOM_SPHERE:
            raise ValueError("only sphere-plane contacts are supported")
        li = link_of_body[i]
        n = np.asarray(raw.plane.normal, float)
        n = n / np.linalg.norm(n)
        f["n_sphere"][0] = 1
        f["sph_link"][0] = li
        f["sph_pos"][:] = p0[i] + R0[i] @ np.asarray(g.a, float) - origin[li]
        f["sph_r"][0] = float(np.ravel(overrides.get("geom_size", {}).get(g.name, [g.radius]))[0])
        f["sph_margin"][0] = max(raw.plane.margin, g.margin) - max(raw.plane.gap, g.gap)     # MuJoCo: max of geom margins, less the larger gap
        f["sph_invweight"][0] = 0.0 + body_iw[i]                 # world body weighs 0
        f["plane_n"][:] = n
        f["plane_d"][0] = n @ np.asarray(raw.plane.pos, float)
    tc, dr = raw.solref
    if tc <= 0 or dr <= 0:
        raise NotImplementedError("arm kernel: solref must be the standard (timeconst, dampratio) pair (direct stiffness / damping: the tree engine)")
    tc = max(tc, 2 * raw.timestep)                               # refsafe
    dmin, dmax, width, mid, power = raw.solimp
    if power < 1 or power != int(power) or power > 64:
        # the kernel evaluates the impedance power by repeated multiplication (no library pow in the hot loop)
        raise NotImplementedError("solimp power must be an integer in [1, 64] (MuJoCo's default is 2), got %r" % (power,))
    f["sol_K"][0] = 1.0 / (dmax * dmax * tc * tc * dr * dr)
    f["sol_B"][0] = 2.0 / (dmax * tc)
    f["sol_dmin"][0], f["sol_dmax"][0] = dmin, dmax
    f["sol_width"][0], f["sol_mid"][0], f["sol_power"][0] = width, mid, power
    f["gravity"][:] = raw.gravity
    f["nu"][0] = len(raw.actuators)
    # friction-loss rows (mj_instantiateFriction): J = e_j at position 0 -> D = 1 / R from the impedance at 0, aref = -b v
    lossy = [j for j in joints if f["frictionloss"][raw.dof_of_joint(j.name)] > 0]
    if lossy:
        sets = {(tuple(raw.solref_friction if j.solref_friction is None else j.solref_friction),
                 tuple(raw.solimp_friction if j.solimp_friction is None else j.solimp_friction)) for j in lossy}
        if len(sets) != 1:
            raise ValueError("arm kernel: the friction-loss rows share one solref / solimp (per-joint sets: the tree engine)")
        (ftc, fdr), fimp = sets.pop()
        if ftc <= 0 or fdr <= 0:
            raise NotImplementedError("arm kernel: solreffriction must be the standard (timeconst, dampratio) pair")
        ftc = max(ftc, 2 * raw.timestep)
        f["floss_B"][0] = 2.0 / (fimp[1] * ftc)
        imp0 = fimp[0]                                          # the impedance at position 0 (x = 0 -> y = 0 for every power)
        for j in lossy:
            d = raw.dof_of_joint(j.name)
            f["floss_D"][d] = 1.0 / max(MJ_MINVAL, (1.0 - imp0) / imp0 * dof_iw[d])

    if base is not None:            # run-time edit: MuJoCo keeps the constants mj_setConst computed at load time
        for j in lossy:             # (the rows' D follows the base model's dof_invweight0)
            d = raw.dof_of_joint(j.name)
            f["floss_D"][d] = 1.0 / max(MJ_MINVAL, (1.0 - imp0) / imp0 * base.dof_invweight0[d])
        f["dof_invweight0"][:] = base.field("dof_invweight0")
        f["sph_invweight"][:] = base.field("sph_invweight")
        dof_iw, body_iw = base.dof_invweight0.copy(), base.body_invweight0.copy()
    blob = np.concatenate([f[name] for name, _ in ARM_LAYOUT]).astype(np.float64)
    assert blob.size == ARM_BLOB_LEN
    return ArmModel(blob=blob, nv=nv, nu=len(raw.actuators), d_obs=2 * nv + 6,
                    timestep=raw.timestep, frame_skip=raw.frame_skip,
                    target_default=np.asarray(raw.target_pos, float),
                    ctrl_lo=ctrl_lo, ctrl_hi=ctrl_hi,
                    body_mass=mass, body_ipos=ipos, body_inertia=inert,
                    body_invweight0=body_iw, dof_invweight0=dof_iw,
                    link_of_body=link_of_body)

"""CPU: the physics added for in-hand manipulation (geom-geom contacts, position actuators), held to
mechanics - like the rest of the physics oracle it is PARITY UNPINNED (no MuJoCo here) - and the host-side tables of
the elimination tree the kernel's sparse factorisation walks."""
import numpy as np

from mjmpc_amd.models.compile_tree import TL, compile_tree
from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
from mjmpc_amd.models.raw import (GEOM_CAPSULE, GEOM_SPHERE, JOINT_SLIDE, RawActuator, RawBody, RawGeom, RawJoint,
                                  RawModel)
from oracle.physics_ref import RefArm

X, Y, Z = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)


def _free_ball(name, pos, radius, density, parent_offset):
    """A sphere on three slide joints (translations only)."""
    sl = dict(limited=False, damping=0.0, armature=0.0, type=JOINT_SLIDE, range=(-9, 9))
    b0 = parent_offset
    return [RawBody(name + "x", -1, pos, joint=RawJoint(axis=X, name=name + "x", **sl)),
            RawBody(name + "y", b0, (0, 0, 0), joint=RawJoint(axis=Y, name=name + "y", **sl)),
            RawBody(name + "z", b0 + 1, (0, 0, 0), joint=RawJoint(axis=Z, name=name + "z", **sl),
                    geoms=[RawGeom(GEOM_SPHERE, radius, (0, 0, 0), density=density, margin=0.001, name=name)])]


def _two_balls(mu=0.0, condim=1):
    bodies = _free_ball("a", (0.0, 0.0, 0.0), 0.05, 1000.0, 0) + _free_ball("b", (0.2, 0.01, 0.0), 0.03, 3000.0, 3)
    for b in bodies:
        for g in b.geoms:
            g.friction, g.condim = mu, condim
    acts = [RawActuator("ax", 1.0, (-1, 1))]
    return RawModel(bodies=bodies, actuators=acts, site_body=2, site_pos=(0, 0, 0), target_pos=(0, 0, 0), plane=None,
                    timestep=0.001, frame_skip=1, pairs=[("b", "a")])


def test_sphere_sphere_collision_conserves_momentum_and_separates():
    """Two free spheres, no gravity, slightly off-centre impact: the contact force acts on both bodies with opposite
    signs (total momentum constant to rounding), along the line of centres, and only pushes."""
    raw = _two_balls()
    ref = RefArm(raw.to_flat())
    mass, _, _ = ref.inertial()
    ma, mb = mass[3], mass[6]
    q, v = np.zeros(6), np.zeros(6)
    v[0], v[3] = 0.5, -0.3                               # approaching along x
    p0 = ma * v[:3] + mb * v[3:]
    touched = False
    for _ in range(400):
        q, v, _, diag = ref.step(q, v, np.zeros(1))
        touched = touched or diag[0] > 0
        np.testing.assert_allclose(ma * v[:3] + mb * v[3:], p0, rtol=0, atol=1e-12)
    assert touched and v[3] - v[0] > 0                  # they met and now separate
    assert abs(v[1]) > 1e-4 and abs(v[2]) < 1e-12       # the off-centre impact deflects in y, nothing in z
    assert ref.newton_stats()["fails"] == 0


def test_position_servo_is_a_spring_towards_the_control():
    """MJCF <position kp>: force = kp (ctrl - q).  One hinge of known inertia, no gravity: small oscillations about the
    control at omega = sqrt(kp / I), settling there with damping."""
    I = 0.4 * (4.0 / 3.0 * np.pi * 0.1 ** 3 * 1000.0) * 0.1 ** 2            # sphere about its centre
    body = RawBody("b", -1, (0, 0, 0), joint=RawJoint(axis=Z, range=(-3, 3), limited=False, damping=0.0, armature=0.0, name="j"),
                   geoms=[RawGeom(GEOM_SPHERE, 0.1, (0, 0, 0), name="g")])
    kp = 2.0
    raw = RawModel(bodies=[body], actuators=[RawActuator("j", 1.0, (-2.0, 2.0), kp=kp)], site_body=0, site_pos=(0, 0, 0),
                   target_pos=(0, 0, 0), plane=None, timestep=0.0005, frame_skip=1)
    ref = RefArm(raw.to_flat())
    q, v = np.array([0.0]), np.array([0.0])
    u = np.array([0.3])
    qs = []
    for _ in range(4000):
        q, v, _, _ = ref.step(q, v, u)
        qs.append(q[0])
    qs = np.array(qs)
    assert abs(qs.max() - 0.6) < 2e-3 and abs(qs.min()) < 2e-3                # undamped: swings between 0 and 2 ctrl
    crossings = np.where(np.diff(np.sign(qs - 0.3)) != 0)[0]
    period = 2 * np.mean(np.diff(crossings)) * raw.timestep
    assert abs(period - 2 * np.pi / np.sqrt(kp / I)) < 0.01 * period
    # the control is clamped to ctrlrange before it enters the servo
    q2, v2, _, _ = ref.step(np.array([0.0]), np.array([0.0]), np.array([50.0]))
    q3, v3, _, _ = ref.step(np.array([0.0]), np.array([0.0]), np.array([2.0]))
    assert v2[0] == v3[0]


def test_pen_rests_on_the_hand_and_friction_carries_it_along():
    """Capsule-capsule contacts with friction cones: the pen settles on the fingers (its weight carried by the contact
    forces: no vertical acceleration), and when the whole hand swings sideways the pen goes with it."""
    raw = pen_hand_raw()
    ref = RefArm(raw.to_flat())
    st = holding_state()
    q, v = st["qp"].copy(), st["qv"].copy()
    u = q[6:].copy()
    for _ in range(600):
        q, v, _, diag = ref.step(q, v, u)
    # resting on the first phalanges (it keeps rolling slowly down their slope: condim 3 has no rolling friction)
    assert diag[0] >= 8 and abs(v[2]) < 0.03 and q[2] > -0.02
    y0, n = q[1], 1500
    for k in range(n):                                                       # pan the arm slowly: the hand sweeps in +y
        u2 = u.copy()
        u2[0] += 0.15 * (k + 1) / n
        q, v, _, diag = ref.step(q, v, u2)
    assert abs(q[6] - 0.15) < 0.02
    assert q[1] - y0 > 0.08 and q[2] > -0.03, (q[:6], "the pen should have travelled ~0.74 sin(0.15) = 0.11 m with the hand")
    assert ref.newton_stats()["fails"] == 0


def test_elimination_tree_tables_factor_a_coupled_matrix():
    """compile_tree's elimination tree (eparent / depth / elim): a numpy walk of the kernel's round-by-round sparse L'DL
    over those tables factors H = M + J'DJ - M block diagonal (object | hand), J touching one finger path AND the
    object - exactly, where the kinematic tree's tables would lose the fill-in."""
    m = compile_tree(pen_hand_raw())
    nv = m.nv
    ep = m.field("eparent")[:nv].astype(int)
    par = m.parent
    rs = np.random.RandomState(0)

    def path(tab, i):
        out = []
        while i >= 0:
            out.append(i)
            i = tab[i]
        return out

    # a random SPD matrix with the KINEMATIC pattern, plus rank-one terms over (finger path + object chain)
    H = np.zeros((nv, nv))
    for i in range(nv):
        for j in path(par, i):
            H[i, j] = H[j, i] = 0.1 * rs.standard_normal()
    H = H @ H.T + np.eye(nv)
    mask = np.zeros((nv, nv), bool)
    for i in range(nv):
        for j in path(par, i):
            mask[i, j] = mask[j, i] = True
    H = np.where(mask, H, 0.0) + 3 * np.eye(nv)
    for leaf in (13, 21, 29):
        jrow = np.zeros(nv)
        for k in path(par, leaf) + path(par, 5):
            jrow[k] = rs.standard_normal()
        H += 5.0 * np.outer(jrow, jrow)
    for i in range(nv):             # everything must lie on elimination paths
        for j in range(nv):
            if H[i, j] != 0:
                assert j in path(ep, i) or i in path(ep, j)
    # the kernel's algorithm: rows path-indexed over the elimination tree, one round per height, pulls from the elim lists
    depth = m.field("depth")[:nv].astype(int)
    elim = m.field("elim").reshape(TL - 1, TL)[:, :nv].astype(int)
    rounds = int(m.field("n_rounds")[0])
    DP = depth.max() + 1
    AT = -np.ones((DP, nv), int)
    for i in range(nv):
        for c, a in enumerate(path(ep, i)):
            AT[c, i] = a
    r = np.zeros((nv, 2 * DP + 1))
    for i in range(nv):
        for c in range(DP):
            if AT[c, i] >= 0:
                r[i, c] = H[i, AT[c, i]]
    e = np.zeros(nv, int)
    for hgt in range(rounds - 1):
        pub = r.copy()
        pub[:, 2 * DP] = 1.0 / r[:, 0]
        for i in range(nv):
            while e[i] < TL - 1 and elim[e[i], i] >= 0 and (elim[e[i], i] >> 16) == hgt:
                ent = elim[e[i], i]
                k, a = ent & 255, (ent >> 8) & 255
                f = pub[k, a] * pub[k, 2 * DP]
                r[i, :DP] -= f * pub[k, a:a + DP]
                e[i] += 1
    D = r[:, 0].copy()
    L = np.eye(nv)
    for i in range(nv):
        for c in range(1, DP):
            if AT[c, i] >= 0:
                L[i, AT[c, i]] = r[i, c] / D[i]
    np.testing.assert_allclose(L.T @ np.diag(D) @ L, H, rtol=0, atol=1e-11)

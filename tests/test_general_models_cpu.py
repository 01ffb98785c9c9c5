"""CPU: the oracle's round-4 physics - friction-loss rows, ball and free joints, joint anchors, boxes, equalities, tendon
limits - held to mechanics and to itself (MuJoCo is absent: PARITY UNPINNED, see oracle/reacher_ref.c), the constraint
solver with its three row kinds against an independent minimiser, the MJCF loader's new features, and the two model
compilers (oracle C / mjmpc_amd.models.compile_tree) against each other on the synthetic models."""
import textwrap

import numpy as np
import pytest

from mjmpc_amd.models.compile_tree import compile_tree
from mjmpc_amd.models.mjcf import load_mjcf
from mjmpc_amd.models.raw import TASK_REACH
from mjmpc_amd.models.synthetic import FRAME_SKIP, start_state, synthetic_raw
from oracle.physics_ref import RefArm

HEAD = '<mujoco><compiler angle="radian" coordinate="local" inertiafromgeom="auto"/>'


def _model(tmp_path, body, name="m.xml", gravity="0 0 -9.81", extra="", timestep="0.002", head=HEAD, **kw):
    xml = head + '<option timestep="%s" gravity="%s" integrator="Euler"/>' % (timestep, gravity) + \
        '<default><geom contype="0" conaffinity="0"/></default><worldbody><site name="target" pos="0 0 0"/>' + \
        textwrap.dedent(body) + "</worldbody>" + extra + "</mujoco>"
    (tmp_path / name).write_text(xml)
    raw = load_mjcf(str(tmp_path / name), task=TASK_REACH, **kw)
    return raw, RefArm(raw.to_flat())


def _run(ref, q, v, u, n):
    for _ in range(n):
        q, v, site, diag = ref.step(q, v, u)
    return q, v, diag


# ------------------------------------------------------------------------------------------ friction loss
BLOCK = """
<body name="block"><joint name="x" type="slide" axis="1 0 0" frictionloss="%g"/>
  <geom type="sphere" size="0.1" mass="%g"/><site name="finger"/></body>"""
ACT = '<actuator><motor joint="x" gear="1" ctrlrange="-100 100" ctrllimited="true"/></actuator>'


def test_friction_loss_holds_below_its_bound_and_slides_above(tmp_path):
    """A 2 kg block on a slide joint with frictionloss 3 N: a 2 N push only makes it creep (the soft constraint's
    steady state: D B v = F), a 5 N push accelerates it at (5 - 3) / m, and it coasts to a stop at 3 / m."""
    raw, ref = _model(tmp_path, BLOCK % (3.0, 2.0), extra=ACT)
    m = ref.inertial()[0][1]
    assert abs(m - 2.0) < 1e-12
    q, v, diag = _run(ref, np.zeros(1), np.zeros(1), np.array([2.0]), 500)
    D = 1.0 / ((1 - 0.9) / 0.9 * (1.0 / m))             # R = (1 - imp) / imp * dof_invweight0 at imp(0) = dmin
    B = 2.0 / (0.95 * 0.02)
    assert 0 < v[0] < 1.05 * 2.0 / (D * B) and abs(v[0] - 2.0 / (D * B)) < 1e-6      # creeping, at the soft constraint's rate
    assert diag[0] == 1
    q, v, _ = _run(ref, np.zeros(1), np.zeros(1), np.array([5.0]), 500)
    assert abs(v[0] - (5.0 - 3.0) / m * 1.0) < 1e-9                                   # 500 steps of 2 ms
    q, v, _ = _run(ref, np.zeros(1), np.array([1.0]), np.zeros(1), 200)               # coasting: -f / m for 0.4 s
    assert abs(v[0] - (1.0 - 3.0 / m * 0.4)) < 1e-9
    q, v, _ = _run(ref, q, v, np.zeros(1), 400)                                       # ... and it stops, and stays
    assert abs(v[0]) < 1e-6


def test_constraint_solver_row_kinds_against_an_independent_minimiser(tmp_path):
    """solve_rows (Newton + exact line search over unilateral, equality and Huber rows) against scipy's minimiser of the
    same convex cost, random problems; and its KKT conditions exactly: M a - fs = J' f with f the rows' slopes."""
    from scipy.optimize import minimize
    raw, ref = _model(tmp_path, BLOCK % (0.0, 1.0), extra=ACT)
    rs = np.random.RandomState(0)
    for trial in range(30):
        nv, nc = rs.randint(2, 7), rs.randint(1, 9)
        A = rs.standard_normal((nv, nv))
        M = A @ A.T + nv * np.eye(nv)
        fs = 5 * rs.standard_normal(nv)
        J = rs.standard_normal((nc, nv)) * (rs.uniform(size=(nc, nv)) < 0.7)
        aref, D = 3 * rs.standard_normal(nc), rs.uniform(0.5, 20, nc)
        kind = rs.randint(0, 3, nc)
        fl = np.where(kind == 2, rs.uniform(0.1, 3.0, nc), 0.0)

        def cost(a):
            r = J @ a - aref
            c = 0.5 * a @ M @ a - fs @ a
            for i in range(nc):
                if kind[i] == 0:
                    c += 0.5 * D[i] * min(0.0, r[i]) ** 2
                elif kind[i] == 1:
                    c += 0.5 * D[i] * r[i] ** 2
                else:
                    Rf = fl[i] / D[i]
                    c += 0.5 * D[i] * r[i] ** 2 if abs(r[i]) < Rf else fl[i] * abs(r[i]) - 0.5 * Rf * fl[i]
            return c

        a, f = ref.solve_rows(M, fs, J, aref, D, kind, fl)
        r = J @ a - aref
        slope = np.where(kind == 0, D * np.minimum(0, r), np.where(kind == 1, D * r, np.clip(D * r, -fl, fl)))
        np.testing.assert_allclose(f, -slope, rtol=0, atol=1e-12)
        np.testing.assert_allclose(M @ a - fs, J.T @ f, rtol=0, atol=1e-9)            # stationarity of the convex cost
        best = minimize(cost, np.linalg.solve(M, fs), method="BFGS", options=dict(gtol=1e-10)).x
        assert cost(a) <= cost(best) + 1e-9
        np.testing.assert_allclose(a, best, rtol=0, atol=2e-5)
    assert ref.newton_stats()["fails"] == 0


# ------------------------------------------------------------------------------------------ free and ball joints
FREE_BODY = """
<body name="brick" pos="0.3 -0.2 1.0" quat="0.9 0.1 -0.3 0.2">
  <freejoint name="f"/>
  <geom name="g" type="box" size="0.1 0.05 0.02" pos="0.01 0 0.02" quat="0.95 0.2 0 0.1" density="700"/>
  <geom type="sphere" pos="0.1 0 0" size="0.03" density="2000"/><site name="finger"/></body>"""


def _world_inertial(ref, q):
    """(mass, com, world-frame inertia about com) of body 1 of a free-jointed model at qpos q."""
    mass, ipos, inertia = ref.inertial()
    w, x, y, z = q[3:7] / np.linalg.norm(q[3:7])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return mass[1], q[:3] + R @ ipos[1], R @ inertia[1] @ R.T, R


def test_free_body_momentum_and_free_fall(tmp_path):
    """A free-jointed body (qpos = position + quaternion, qvel = world linear + body-frame angular velocity) with an
    off-centre mass: without gravity its linear momentum, its angular momentum about the centre of mass and its kinetic
    energy are conserved to the integrator's order - the errors HALVE with the time step (the body origin is not the
    centre of mass, so M depends on q and semi-implicit Euler is first order); with gravity the centre of mass falls at g."""
    raw, ref = _model(tmp_path, FREE_BODY, gravity="0 0 0", timestep="0.0005")
    assert (ref.nv, ref.nq) == (6, 7)
    np.testing.assert_allclose(ref.qpos0, [0.3, -0.2, 1.0] + list(np.array([0.9, 0.1, -0.3, 0.2]) / np.linalg.norm([0.9, 0.1, -0.3, 0.2])))
    q, v = ref.qpos0.copy(), np.array([0.3, -0.1, 0.2, 2.0, -3.0, 1.5])

    def momenta(q, v):
        m, com, Iw, R = _world_inertial(ref, q)
        w = R @ v[3:]
        vc = v[:3] + np.cross(w, com - q[:3])
        return m * vc, Iw @ w, 0.5 * m * vc @ vc + 0.5 * w @ Iw @ w

    p0, L0, E0 = momenta(q, v)
    assert abs(ref.kinetic(q, v) - E0) < 1e-12          # the Jacobian-built mass matrix agrees with rigid-body mechanics
    q0, v0 = q.copy(), v.copy()
    for _ in range(2000):
        q, v, _, _ = ref.step(q, v, np.zeros(0))
    p1, L1, E1 = momenta(q, v)
    ep, eL, eE = np.linalg.norm(p1 - p0) / np.linalg.norm(p0), np.linalg.norm(L1 - L0) / np.linalg.norm(L0), abs(E1 - E0) / E0
    assert ep < 3e-3 and eL < 3e-3 and eE < 3e-3
    assert abs(np.linalg.norm(q[3:7]) - 1) < 1e-12
    raw, ref = _model(tmp_path, FREE_BODY, gravity="0 0 0", timestep="0.00025", name="half.xml")
    q, v = q0, v0
    for _ in range(4000):
        q, v, _, _ = ref.step(q, v, np.zeros(0))
    p2, L2, E2 = momenta(q, v)
    assert np.linalg.norm(p2 - p0) / np.linalg.norm(p0) < 0.6 * ep and np.linalg.norm(L2 - L0) / np.linalg.norm(L0) < 0.6 * eL
    assert abs(E2 - E0) / E0 < 0.6 * eE
    raw, ref = _model(tmp_path, FREE_BODY, name="g.xml")
    q, v = ref.qpos0.copy(), np.array([0.0, 0.0, 0.0, 1.0, 2.0, -1.0])
    _, com0, _, R = _world_inertial(ref, q)
    vc0 = v[:3] + np.cross(R @ v[3:], com0 - q[:3])             # the centre of mass moves with the spin about the origin
    for _ in range(100):
        q, v, _, _ = ref.step(q, v, np.zeros(0))
    _, com1, _, _ = _world_inertial(ref, q)
    t = 0.2
    np.testing.assert_allclose(com1 - com0, vc0 * t + np.array([0, 0, -0.5 * 9.81 * t * (t + 0.002)]), rtol=0, atol=3e-4)


BALL = """
<body name="bob" pos="0.1 0.2 1.0" quat="0.8 0.2 0.1 -0.3">
  <joint name="b" type="ball" pos="%s"/>
  <geom type="capsule" fromto="%s" size="0.02" density="900"/><site name="finger" pos="%s"/></body>"""


def test_ball_joint_pendulum_and_anchor_offset(tmp_path):
    """A ball-jointed pendulum: energy and the angular momentum about the vertical through the anchor are conserved to
    the integrator's order; and the SAME pendulum written with its body frame at the anchor (joint pos 0) instead of
    0.3 m away from it (joint pos = the anchor in the body frame) moves identically."""
    raw_a, ref_a = _model(tmp_path, BALL % ("0 0 0.3", "0 0 0.3 0.05 0 -0.1", "0.05 0 -0.1"), name="a.xml", timestep="0.0005")
    # the same body, frame moved to the anchor: body pos += R0 (0, 0, 0.3); geometry shifted by -(0, 0, 0.3)
    R0 = np.array(ref_a.mass_matrix(ref_a.qpos0)) * 0           # (placeholder to keep numpy import used)
    from mjmpc_amd.models.compile import _quat2mat
    R0 = _quat2mat([0.8, 0.2, 0.1, -0.3])
    p = np.array([0.1, 0.2, 1.0]) + R0 @ np.array([0, 0, 0.3])
    body_b = BALL.replace('pos="0.1 0.2 1.0"', 'pos="%.17g %.17g %.17g"' % tuple(p)) % ("0 0 0", "0 0 0 0.05 0 -0.4", "0.05 0 -0.4")
    raw_b, ref_b = _model(tmp_path, body_b, name="b.xml", timestep="0.0005")
    qa, va = ref_a.qpos0.copy(), np.array([1.0, -2.0, 0.5])
    qb, vb = qa.copy(), va.copy()
    anchor = p

    def energy_Lz(ref, q, v):
        m, ipos, inertia = ref.inertial()
        from mjmpc_amd.models.compile import _quat2mat as qm
        R = R0 @ qm(q)                                          # body orientation: parent o body quat o joint quat
        w = R @ v
        # body origin from the anchor (joint pos fixed in the body frame)
        jpos = np.array([0, 0, 0.3]) if ref is ref_a else np.zeros(3)
        origin = anchor - R @ jpos
        com = origin + R @ ipos[1]
        vc = np.cross(w, com - anchor)
        Iw = R @ inertia[1] @ R.T
        return (0.5 * m[1] * vc @ vc + 0.5 * w @ Iw @ w + m[1] * 9.81 * com[2],
                (m[1] * np.cross(com - anchor, vc) + Iw @ w)[2], com)

    E0, L0, c0 = energy_Lz(ref_a, qa, va)
    Eb, Lb, cb = energy_Lz(ref_b, qb, vb)
    assert abs(E0 - Eb) < 1e-12 and abs(L0 - Lb) < 1e-12
    swing = 0.0
    for _ in range(2000):
        qa, va, sa, _ = ref_a.step(qa, va, np.zeros(0))
        qb, vb, sb, _ = ref_b.step(qb, vb, np.zeros(0))
        swing = max(swing, np.linalg.norm(energy_Lz(ref_a, qa, va)[2] - c0))
    np.testing.assert_allclose(qa, qb, rtol=0, atol=1e-10)
    np.testing.assert_allclose(va, vb, rtol=0, atol=1e-9)
    np.testing.assert_allclose(sa, sb, rtol=0, atol=1e-10)      # the tracked site, through two different frames
    E1, L1, c1 = energy_Lz(ref_a, qa, va)
    assert swing > 0.1                                          # it did swing (about one period in this second)
    assert abs(E1 - E0) < 5e-3 * abs(E0) and abs(L1 - L0) < 5e-3 * max(abs(L0), 0.01)


def test_hinge_anchor_offset_and_explicit_inertial(tmp_path):
    """A door leaf hinged at its edge: joint pos off the body origin = the same leaf with its frame on the hinge line; and
    an explicit <inertial> = the geom it was computed from."""
    leaf_a = """<body name="leaf" pos="0.45 0 1"><joint name="h" type="hinge" axis="0 0.1 1" pos="-0.45 0 0" damping="0.1"/>
      <geom type="box" size="0.45 0.02 1.0" density="300"/><site name="finger" pos="0.4 0 0"/></body>"""
    leaf_b = """<body name="leaf" pos="0 0 1"><joint name="h" type="hinge" axis="0 0.1 1" damping="0.1"/>
      <geom type="box" size="0.45 0.02 1.0" pos="0.45 0 0" density="300"/><site name="finger" pos="0.85 0 0"/></body>"""
    m = 300 * 8 * 0.45 * 0.02 * 1.0
    leaf_c = leaf_a.replace('<geom type="box" size="0.45 0.02 1.0" density="300"/>',
                            '<inertial pos="0 0 0" mass="%.17g" diaginertia="%.17g %.17g %.17g"/><geom type="sphere" size="0.01" density="0"/>'
                            % (m, m / 3 * (0.02 ** 2 + 1.0), m / 3 * (0.45 ** 2 + 1.0), m / 3 * (0.45 ** 2 + 0.02 ** 2)))
    runs = []
    for k, body in enumerate((leaf_a, leaf_b, leaf_c)):
        raw, ref = _model(tmp_path, body, name="leaf%d.xml" % k, gravity="0 -3 -9.81")
        q, v = np.array([0.3]), np.array([1.0])
        sites = []
        for _ in range(300):
            q, v, s, _ = ref.step(q, v, np.zeros(0))
            sites.append(s)
        runs.append((q, v, np.array(sites)))
    for q, v, s in runs[1:]:
        np.testing.assert_allclose(q, runs[0][0], rtol=0, atol=1e-11)
        np.testing.assert_allclose(v, runs[0][1], rtol=0, atol=1e-10)
        np.testing.assert_allclose(s, runs[0][2], rtol=0, atol=1e-11)


# ------------------------------------------------------------------------------------------ boxes
def test_box_rests_and_slides_on_the_plane(tmp_path):
    """A free box on the world plane: four corner contacts carry it at rest (penetration of the soft contacts: a fraction
    of a millimetre), and sliding it decelerates at mu g on average (pyramidal cone along an axis)."""
    body = """<geom name="floor" type="plane" pos="0 0 0" size="5 5 0.1" contype="1" conaffinity="1" friction="0.5 0.005 0.0001" condim="3"/>
    <body name="box" pos="0 0 0.0501"><freejoint/>
      <geom name="b" type="box" size="0.1 0.08 0.05" density="800" contype="1" conaffinity="1" friction="0.5 0.005 0.0001" condim="3"/>
      <site name="finger"/></body>"""
    raw, ref = _model(tmp_path, body)
    q, v = ref.qpos0.copy(), np.zeros(6)
    for _ in range(500):
        q, v, _, diag = ref.step(q, v, np.zeros(0))
    assert diag[0] == 16                                        # 4 corners x 4 pyramid rows
    assert 0.0495 < q[2] < 0.0501 and np.abs(v).max() < 1e-6
    np.testing.assert_allclose(q[3:7], [1, 0, 0, 0], atol=1e-9)
    v = np.array([1.5, 0, 0, 0, 0, 0])
    v0 = []
    for _ in range(100):
        q, v, _, _ = ref.step(q, v, np.zeros(0))
        v0.append(v[0])
    dec = (v0[0] - v0[-1]) / (99 * 0.002)
    assert abs(dec - 0.5 * 9.81) < 0.08 * 0.5 * 9.81, dec       # (soft pyramid rows: a few per cent under mu g; the box pitches)


def test_sphere_rests_on_a_static_box(tmp_path):
    """Sphere / box contact against a STATIC geom of the world body: a ball dropped on a block (turned about the vertical)
    settles on its top face; dropped just past an edge it touches the edge's nearest point - the normal is diagonal - and
    is pushed off sideways."""
    body = """<geom name="block" type="box" pos="0 0 0.2" size="0.2 0.2 0.05" euler="0 0 0.5" friction="1.5 0.005 0.0001" condim="3"/>
    <body name="ball" pos="%s 0 0.32"><freejoint/>
      <geom name="s" type="sphere" size="0.05" density="1000" friction="1.5 0.005 0.0001" condim="3"/><site name="finger"/></body>"""
    extra = '<contact><pair geom1="s" geom2="block"/></contact>'
    raw, ref = _model(tmp_path, body % "0.05", extra=extra)
    q, v = ref.qpos0.copy(), np.zeros(6)
    for _ in range(800):
        q, v, _, diag = ref.step(q, v, np.zeros(0))
    assert diag[0] == 4 and -1e-3 < q[2] - 0.30 < 1e-5 and np.abs(v).max() < 1e-5 and abs(q[0] - 0.05) < 1e-6
    # past the edge: in the block's frame the ball sits 0.02 m beyond the +x face, centre above the top face
    c, s_ = np.cos(0.5), np.sin(0.5)
    raw, ref = _model(tmp_path, (body % ("%.17g" % (0.22 * c))).replace('pos="%.17g 0 0.32"' % (0.22 * c), 'pos="%.17g %.17g 0.32"' % (0.22 * c, 0.22 * s_)),
                      extra=extra, name="edge.xml")
    q, v = ref.qpos0.copy(), np.zeros(6)
    hit = False
    for _ in range(300):
        q, v, _, diag = ref.step(q, v, np.zeros(0))
        hit = hit or diag[0] > 0
    out = np.array([c, s_, 0.0])                                # the face's outward direction
    assert hit and v[:3] @ out > 0.05 and q[2] < 0.28           # pushed off along the face normal, and falling
    assert abs(v[:3] @ np.array([-s_, c, 0.0])) < 0.02 * (v[:3] @ out)     # (next to) nothing along the edge: the friction
                                                                           # pyramid's axes are not the edge's


# ------------------------------------------------------------------------------------------ equalities, tendon
def test_fourbar_loop_stays_closed_and_the_tendon_limit_holds():
    raw = synthetic_raw("fourbar")
    ref = RefArm(raw.to_flat())
    q, v = ref.qpos0.copy(), np.zeros(6)
    tip0 = np.array([0.4, 0, 0.4])                              # the rocker's tip at qpos0: pinned there
    worst, tmax = 0.0, 0.0

    def tip(q):
        a0, a1, a2 = q[0], q[0] + q[1], q[0] + q[1] + q[2]
        # planar chain about y: crank 0.2 up, coupler 0.4 along x, rocker 0.3 down
        rot = lambda a, vec: np.array([np.cos(a) * vec[0] + np.sin(a) * vec[2], 0, -np.sin(a) * vec[0] + np.cos(a) * vec[2]])
        return np.array([0, 0, 0.5]) + rot(a0, [0, 0, 0.2]) + rot(a1, [0.4, 0, 0]) + rot(a2, [0, 0, -0.3])

    np.testing.assert_allclose(tip(q), tip0, atol=1e-12)
    for k in range(1500):
        u = np.array([0.1 if k < 700 else -0.1])
        q, v, _, diag = ref.step(q, v, u)
        worst = max(worst, np.linalg.norm(tip(q) - tip0))
        tmax = max(tmax, abs(q[0] + 0.5 * q[1]))
    assert worst < 3e-3, worst                                  # the soft constraint's violation (the crank reaches 10 rad/s)
    assert abs(q[0]) > 0.2                                      # the linkage did move
    assert 0.6 < tmax < 0.6 + 0.015, tmax                       # tendon length j0 + 0.5 j1 limited to +-0.6: reached, held
    assert ref.newton_stats()["fails"] == 0


def test_door_latch_follows_the_handle_and_holds_the_door():
    raw = synthetic_raw("door")
    ref = RefArm(raw.to_flat())
    q, v = ref.qpos0.copy(), np.zeros(3)
    # handle at rest, bolt out: pushing the door (1 N m against 0.3 N m of hinge friction) leaves it latched against the strike
    for _ in range(400):
        q, v, _, diag = ref.step(q, v, np.array([0.0, 0.25]))
    assert q[0] < 0.01 and diag[0] >= 6, (q, diag[0])          # equality + hinge friction + the bolt on the strike (4 rows)
    # turn the handle (servo to 1.05 rad): the joint equality retracts the bolt; the same push now opens the door
    worst = 0.0
    for _ in range(900):
        q, v, _, diag = ref.step(q, v, np.array([1.05, 0.25]))
        worst = max(worst, abs(q[2] + 0.02865 * q[1]))
    assert worst < 4e-3, worst                                  # (the bolt is dragged along the strike while it retracts)
    assert abs(q[2] + 0.02865 * q[1]) < 5e-4
    assert q[1] > 0.7 and q[2] < -0.02 and q[0] > 0.3, q
    assert ref.newton_stats()["fails"] == 0


# ------------------------------------------------------------------------------------------ loader and compilers
def test_degrees_and_radians_load_the_same_model(tmp_path):
    body_deg = """<body name="a" pos="0 0 1" euler="0 30 0"><joint name="j" type="hinge" axis="0 1 0" limited="true" range="-45 90" springref="10" stiffness="1"/>
      <geom type="capsule" size="0.02 0.1" axisangle="1 0 0 90" density="500"/><site name="finger"/></body>"""
    body_rad = body_deg.replace('euler="0 30 0"', 'euler="0 %.17g 0"' % np.deg2rad(30)).replace('range="-45 90"', 'range="%.17g %.17g"' % (np.deg2rad(-45), np.deg2rad(90))) \
        .replace('springref="10"', 'springref="%.17g"' % np.deg2rad(10)).replace('axisangle="1 0 0 90"', 'axisangle="1 0 0 %.17g"' % np.deg2rad(90))
    rd, _ = _model(tmp_path, body_deg, name="deg.xml", head=HEAD.replace("radian", "degree"))
    rr, _ = _model(tmp_path, body_rad, name="rad.xml")
    np.testing.assert_allclose(rd.to_flat(), rr.to_flat(), rtol=0, atol=1e-15)
    assert abs(rd.bodies[0].joint.range[1] - np.pi / 2) < 1e-15


def test_ball_range_in_degrees_inertial_under_inertiafromgeom_and_autolimits(tmp_path):
    """ADVICE r4 (MuJoCo's compiler semantics): a BALL joint's range is an angle too; inertiafromgeom="true" ignores an
    explicit <inertial>; limited="auto" follows <compiler autolimits>."""
    ball = """<body name="a" pos="0 0 1"><joint name="j" type="ball" limited="true" range="0 60"/>
      <geom type="capsule" size="0.02 0.1" density="500"/><site name="finger"/></body>"""
    rd, _ = _model(tmp_path, ball, name="bd.xml", head=HEAD.replace("radian", "degree"))
    assert abs(rd.bodies[0].joint.range[1] - np.pi / 3) < 1e-15
    rr, _ = _model(tmp_path, ball.replace("0 60", "0 %.17g" % (np.pi / 3)), name="br.xml")
    np.testing.assert_allclose(rd.to_flat(), rr.to_flat(), rtol=0, atol=1e-15)
    # an explicit inertial beside a geom: used under "auto", ignored under "true"
    body = """<body name="a" pos="0 0 1"><joint name="j" type="hinge" axis="0 1 0"/><inertial pos="0 0 0" mass="7" diaginertia="1 1 1"/>
      <geom type="sphere" size="0.1" density="1000"/><site name="finger"/></body>"""
    ra, _ = _model(tmp_path, body, name="ia.xml")
    rt, _ = _model(tmp_path, body, name="it.xml", head=HEAD.replace('inertiafromgeom="auto"', 'inertiafromgeom="true"'))
    assert ra.bodies[0].inertial is not None and ra.bodies[0].inertial.mass == 7.0
    assert rt.bodies[0].inertial is None
    m_geom = 1000 * 4 / 3 * np.pi * 0.1 ** 3
    from mjmpc_amd.models.compile_tree import body_inertials
    np.testing.assert_allclose(body_inertials(rt, {})[2][0], m_geom, rtol=1e-12)
    np.testing.assert_allclose(body_inertials(ra, {})[2][0], 7.0, rtol=1e-12)
    # limited="auto": a range implies the limit only with autolimits
    auto = """<body name="a" pos="0 0 1"><joint name="j" type="hinge" axis="0 1 0" limited="auto" range="-1 1"/>
      <joint name="k" type="hinge" axis="1 0 0" limited="auto"/><geom type="sphere" size="0.1"/><site name="finger"/></body>"""
    r1, _ = _model(tmp_path, auto, name="al.xml", head=HEAD.replace("<compiler ", '<compiler autolimits="true" '))
    assert [b.joint.limited for b in r1.bodies] == [True, False]
    with pytest.raises(ValueError, match="autolimits"):
        _model(tmp_path, auto, name="al2.xml")


def test_loader_refuses_what_is_not_modelled(tmp_path):
    for body, extra, msg in [
        ('<body name="a"><joint/><geom type="ellipsoid" size="0.1 0.1 0.2"/><site name="finger"/></body>', "", "geom type"),
        ('<body name="a"><joint name="j"/><geom type="sphere" size="0.1"/><site name="finger"/></body>',
         '<equality><distance geom1="a" geom2="b"/></equality>', "equality"),
        ('<body name="a"><joint name="j"/><geom type="sphere" size="0.1"/><site name="finger"/></body>',
         '<tendon><spatial/></tendon>', "fixed tendons"),
    ]:
        with pytest.raises(ValueError, match=msg):
            _model(tmp_path, body, extra=extra, name="bad.xml")
    # a cylinder against a box (or another cylinder): derived from the masks -> refused with advice; a box against the plane,
    # spheres and (round 5) capsules and boxes is fine, and so is a cylinder against spheres and capsules
    body = """<body name="a"><freejoint/><geom name="x" type="box" size="0.1 0.1 0.1" contype="1" conaffinity="1"/><site name="finger"/></body>
    <body name="b" pos="1 0 0"><freejoint/><geom name="y" type="%s" size="0.1 0.1 0.1" contype="1" conaffinity="1"/></body>"""
    rawbb, _ = _model(tmp_path, body % "box", name="bb.xml")            # (box-box: four contact records, round 5)
    assert rawbb.pairs == [("y", "x")]
    with pytest.raises(ValueError, match="cylinder collides with the plane, spheres and capsules only"):
        _model(tmp_path, (body % "cylinder").replace('size="0.1 0.1 0.1" contype="1" conaffinity="1"/></body>', 'size="0.1 0.1" contype="1" conaffinity="1"/></body>'), name="bc.xml")
    raw, _ = _model(tmp_path, (body % "capsule").replace('size="0.1 0.1 0.1" contype="1" conaffinity="1"/></body>', 'size="0.1 0.1" contype="1" conaffinity="1"/></body>'), name="bk.xml")
    assert raw.pairs == [("y", "x")]


@pytest.mark.parametrize("name", sorted(FRAME_SKIP))
def test_two_compilers_agree_on_the_synthetic_models(name):
    """The oracle's C model compile and the host compiler for the kernel: masses, inertias, dof / body invweight0 (with
    MuJoCo's averaging over ball and free joints), qpos0, dimensions."""
    raw = synthetic_raw(name)
    m, ref = compile_tree(raw), RefArm(raw.to_flat())
    assert m.general and (ref.nv, ref.nq, ref.d_obs) == (m.nv, m.nq, m.d_obs)
    mass, ipos, inertia = ref.inertial()
    np.testing.assert_allclose(mass[1:], m.body_mass, rtol=1e-13)
    np.testing.assert_allclose(inertia[1:], m.body_inertia, rtol=0, atol=1e-13)
    d, b = ref.invweight0()
    np.testing.assert_allclose(d, m.dof_invweight0, rtol=1e-10)
    np.testing.assert_allclose(b[1:], m.body_invweight0, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(ref.qpos0, m.qpos0, rtol=0, atol=1e-15)
    st = start_state(name, raw)
    assert st["qp"].shape == (m.nq,) and st["qv"].shape == (m.nv,)


WELDED = """
<body name="a" pos="0 0 1" quat="0.9 0.1 0.2 -0.1"><freejoint/>
  <geom type="box" size="0.1 0.06 0.03" density="600"/><site name="finger"/></body>
<body name="b" pos="0.25 0.05 1.1" quat="0.7 -0.3 0.1 0.4"><freejoint/>
  <geom type="capsule" fromto="0 0 0 0.1 0 0.05" size="0.03" density="900"/></body>
<body name="arm" pos="-0.5 0 1"><joint name="h" type="hinge" axis="0 1 0" damping="0.01"/>
  <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.02" density="500"/></body>"""
WELDS = ('<equality><weld body1="a" body2="b" solref="0.005 1"/><weld body1="arm" solref="0.005 1"/></equality>'
         '<actuator><motor joint="h" gear="1" ctrlrange="-1 1" ctrllimited="true"/></actuator>')


def test_weld_keeps_the_relative_pose(tmp_path):
    """Two free bodies welded to each other tumble as ONE rigid body (their relative pose at qpos0 is kept to a fraction
    of a millimetre / milliradian while the pair falls and spins), and a pendulum welded to the world stays put."""
    from mjmpc_amd.models.compile import _quat2mat
    raw, ref = _model(tmp_path, WELDED, extra=WELDS, timestep="0.001")
    m = compile_tree(raw)
    assert m.general and (m.nv, m.nq) == (13, 15) and int(m.field("n_sphere")[0]) == 4      # two records per weld
    np.testing.assert_allclose(ref.invweight0_rot()[1:], m.body_invweight0_rot, rtol=1e-10)
    q, v = ref.qpos0.copy(), np.zeros(13)
    v[0:6] = [0.3, 0.0, 0.5, 2.0, -1.0, 1.5]
    v[6:12] = [0.3, 0.0, 0.5, 0.0, 0.0, 0.0]                    # (not consistent with a's spin: the weld pulls it along)

    def rel(q):
        Ra, Rb = _quat2mat(q[3:7]), _quat2mat(q[10:14])
        return Ra.T @ (q[7:10] - q[0:3]), Ra.T @ Rb

    p0, R0 = rel(q)
    worst_p, worst_R = 0.0, 0.0
    for k in range(600):
        q, v, _, diag = ref.step(q, v, np.zeros(1))
        if k > 100:                                             # (after the inconsistent start has been absorbed)
            p, R = rel(q)
            worst_p = max(worst_p, np.linalg.norm(p - p0))
            worst_R = max(worst_R, np.linalg.norm(R - R0))
    assert diag[0] == 12                                        # 2 welds x (3 + 3) rows
    assert worst_p < 1e-3 and worst_R < 5e-3, (worst_p, worst_R)
    assert abs(q[14]) < 2e-3                                    # the welded pendulum did not fall
    assert q[2] < 1.0 - 0.5 * 9.81 * 0.6 ** 2 * 0.8            # ... while the pair did
    assert ref.newton_stats()["fails"] == 0


# ------------------------------------------------------------------------------------------ actuators
ARM2 = """
<body name="a" pos="0 0 1"><joint name="j1" type="hinge" axis="0 1 0" damping="0.1" armature="0.01"/>
  <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.03"/>
  <body name="b" pos="0.3 0 0"><joint name="j2" type="hinge" axis="0 1 0" damping="0.1" armature="0.01"/>
    <geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.02"/><site name="finger" pos="0.2 0 0"/></body></body>"""


def test_velocity_and_general_actuators(tmp_path):
    """mj_fwdActuation with gaintype fixed / biastype affine [EXT]: scalar force = gain ctrl + b0 + b1 (gear q) + b2 (gear v),
    joint torque = gear force.  A <velocity kv> servo settles at the commanded joint speed; a <general> with the
    parameters of a <position kp> IS that servo (same trajectory to rounding); the constant bias b0 acts like a motor at
    ctrl = b0 / gain; ctrllimited="false" leaves the control unclamped; defaults reach every shortcut."""
    acts = ('<actuator><velocity joint="j1" kv="5" gear="2" ctrlrange="-3 3" ctrllimited="true"/>'
            '<motor joint="j2" gear="1" ctrlrange="-1 1" ctrllimited="true"/></actuator>')
    raw, ref = _model(tmp_path, ARM2, gravity="0 0 0", extra=acts, name="vel.xml")
    a = raw.actuators[0]
    assert a.gain == 5.0 and a.bias == (0.0, 0.0, -5.0) and raw.actuators[1].gain == 1.0 and raw.actuators[1].bias == (0, 0, 0)
    q, v, _ = _run(ref, raw.qpos0.copy(), np.zeros(2), np.array([1.5, 0.0]), 3000)
    # steady state: gear kv (u - gear w) = damping w  ->  w = gear kv u / (damping + gear^2 kv)
    assert abs(v[0] - 2 * 5 * 1.5 / (0.1 + 4 * 5)) < 1e-6
    # general == position
    pos = '<actuator><position joint="j1" kp="30" gear="1.5" ctrlrange="-1 1" ctrllimited="true"/><motor joint="j2" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    gen = ('<actuator><general joint="j1" gainprm="30" biastype="affine" biasprm="0 -30 0" gear="1.5" ctrlrange="-1 1" ctrllimited="true"/>'
           '<motor joint="j2" ctrlrange="-1 1" ctrllimited="true"/></actuator>')
    _, rp = _model(tmp_path, ARM2, extra=pos, name="pos.xml")
    rg_raw, rg = _model(tmp_path, ARM2, extra=gen, name="gen.xml")
    assert rg_raw.actuators[0].gain == 30.0 and rg_raw.actuators[0].bias == (0.0, -30.0, 0.0)
    u = np.array([0.4, -0.3])
    qa, va, _ = _run(rp, rg_raw.qpos0.copy(), np.zeros(2), u, 400)
    qb, vb, _ = _run(rg, rg_raw.qpos0.copy(), np.zeros(2), u, 400)
    np.testing.assert_allclose(np.r_[qa, va], np.r_[qb, vb], rtol=0, atol=1e-10)
    # constant bias = a motor held at b0 / gain; unclamped control
    b0 = ('<actuator><general joint="j1" gainprm="2" biastype="affine" biasprm="0.6 0 0" gear="3" ctrlrange="-1 1" ctrllimited="false"/>'
          '<motor joint="j2" ctrlrange="-1 1" ctrllimited="true"/></actuator>')
    mot = '<actuator><motor joint="j1" gear="3" ctrlrange="-100 100" ctrllimited="true"/><motor joint="j2" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    rb_raw, rb = _model(tmp_path, ARM2, extra=b0, name="b0.xml")
    _, rm = _model(tmp_path, ARM2, extra=mot, name="mot.xml")
    assert not rb_raw.actuators[0].ctrllimited
    qa, va, _ = _run(rb, rb_raw.qpos0.copy(), np.zeros(2), np.array([2.5, 0.0]), 200)         # 2 * 2.5 + 0.6 = 5.6 (beyond the range)
    qb, vb, _ = _run(rm, rb_raw.qpos0.copy(), np.zeros(2), np.array([5.6, 0.0]), 200)
    np.testing.assert_allclose(np.r_[qa, va], np.r_[qb, vb], rtol=0, atol=1e-10)
    # one actuator default per class, whichever shortcut sets it
    dflt = HEAD.replace("<mujoco>", '<mujoco><default><general ctrllimited="true" ctrlrange="-2 2" gear="4"/></default>')
    rd, _ = _model(tmp_path, ARM2, extra='<actuator><velocity joint="j1" kv="3"/><motor joint="j2"/></actuator>', name="d.xml", head=dflt)
    assert rd.actuators[0].gear == 4.0 and list(rd.actuators[1].ctrlrange) == [-2.0, 2.0] and rd.actuators[0].ctrllimited
    # the compiled blocks: ctrl part through an effective gear, biases at the joint; unclamped -> infinite bounds in the kernel's block
    tm = compile_tree(rb_raw)
    assert tm.field("gear")[0] == 6.0 and tm.field("tau0")[0] == 3 * 0.6 and np.isinf(tm.field("ctrl_lo")[0]) and tm.ctrl_lo[0] == -1.0
    tv = compile_tree(raw)
    assert tv.field("kvg")[0] == 4 * 5.0 and tv.field("kpg")[0] == 0.0 and tv.field("gear")[0] == 10.0
    # forcerange: a position servo far from its target pushes with the bound - a motor held at that force
    sat = ('<actuator><position joint="j1" kp="200" gear="2" ctrlrange="-3 3" ctrllimited="true" forcelimited="true" forcerange="-1.5 0.7"/>'
           '<motor joint="j2" ctrlrange="-1 1" ctrllimited="true"/></actuator>')
    mot2 = '<actuator><motor joint="j1" gear="2" ctrlrange="-3 3" ctrllimited="true"/><motor joint="j2" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    rs_raw, rsat = _model(tmp_path, ARM2, extra=sat, name="sat.xml")
    _, rmo = _model(tmp_path, ARM2, extra=mot2, name="mot2.xml")
    assert tuple(rs_raw.actuators[0].forcerange) == (-1.5, 0.7)
    qa, va, _ = _run(rsat, rs_raw.qpos0.copy(), np.zeros(2), np.array([3.0, 0.0]), 30)         # 200 * (3 - 2 q) >> 0.7: saturated
    qb, vb, _ = _run(rmo, rs_raw.qpos0.copy(), np.zeros(2), np.array([0.7, 0.0]), 30)
    np.testing.assert_allclose(np.r_[qa, va], np.r_[qb, vb], rtol=0, atol=1e-12)
    ts = compile_tree(rs_raw)
    assert ts.field("tau_lo")[0] == -3.0 and ts.field("tau_hi")[0] == 1.4 and np.isinf(ts.field("tau_hi")[1])
    with pytest.raises(ValueError, match="dyntype"):
        _model(tmp_path, ARM2, extra='<actuator><general joint="j1" dyntype="integrator" ctrlrange="-1 1"/><motor joint="j2" ctrlrange="-1 1"/></actuator>', name="dyn.xml")


# ------------------------------------------------------------------------------------------ the layout of robot models
def test_includes_visual_meshes_excludes_and_sensors(tmp_path):
    """What real robot MJCF files are made of and the simulation does not depend on: <include>d parts, <asset> meshes on
    visual geoms (masked out of collisions, on bodies with explicit inertials), <sensor> / <keyframe> sections; and
    <contact><exclude>, which does change the collision candidates.  The model with all of that equals the plain one."""
    body = """
    <body name="a" pos="0 0 0.5"><joint name="j1" type="hinge" axis="0 1 0" damping="0.2"/>
      <inertial pos="0.1 0 0" mass="1.2" diaginertia="0.01 0.02 0.02"/>
      <geom name="ca" type="capsule" fromto="0 0 0 0.3 0 0" size="0.03" contype="1" conaffinity="1"/>%s
      <body name="b" pos="0.3 0 0"><joint name="j2" type="hinge" axis="0 1 0" damping="0.2"/>
        <inertial pos="0.1 0 0" mass="0.5" diaginertia="0.004 0.006 0.006"/>
        <geom name="cb" type="capsule" fromto="0 0 0 0.2 0 0" size="0.02" contype="1" conaffinity="1"/>
        <body name="c" pos="0.2 0 0"><joint name="j3" type="hinge" axis="0 1 0" damping="0.2"/>
          <inertial pos="0.05 0 0" mass="0.2" diaginertia="0.001 0.002 0.002"/>
          <geom name="cc" type="capsule" fromto="0 0 0 0.15 0 0" size="0.02" contype="1" conaffinity="1"/><site name="finger" pos="0.15 0 0"/>
        </body></body></body>"""
    acts = '<actuator>' + "".join('<motor joint="j%d" ctrlrange="-1 1" ctrllimited="true"/>' % k for k in (1, 2, 3)) + '</actuator>'
    plain, ref0 = _model(tmp_path, body % "", extra=acts, name="plain.xml")
    assert [tuple(p) for p in plain.pairs] == [("cc", "ca")]            # a and c are not parent and child: MuJoCo would collide them
    (tmp_path / "acts.xml").write_text("<mujocoinclude>" + acts + "</mujocoinclude>")
    (tmp_path / "vis.xml").write_text('<mujocoinclude><asset><mesh name="shell" file="shell.stl"/></asset>'
                                      '<sensor><jointpos name="j1_pos" joint="j1" noise="0.01"/></sensor></mujocoinclude>')
    vis = '<geom type="mesh" mesh="shell" contype="0" conaffinity="0"/><geom type="cylinder" size="0.05 0.1" contype="0" conaffinity="0"/>'
    extra = ('<include file="acts.xml"/><include file="vis.xml"/><keyframe><key qpos="0 0 0"/></keyframe>')
    rich, ref1 = _model(tmp_path, body % vis, extra=extra, name="rich.xml")
    assert np.array_equal(plain.to_flat(), rich.to_flat())
    assert rich.sensors == {"j1_pos": 0.01} and plain.sensors == {}     # (kept by name for randomize_dynamics' sensor_noise)
    ex, _ = _model(tmp_path, body % "", extra=acts + '<contact><exclude body1="a" body2="c"/></contact>', name="ex.xml")
    assert ex.pairs == []
    # a mesh that would collide, or whose body takes its mass from its geoms, is refused
    with pytest.raises(ValueError, match="visual"):
        _model(tmp_path, body % '<geom type="mesh" mesh="shell" contype="1" conaffinity="0"/>', extra=acts, name="bad1.xml")
    no_inertial = (body % vis).replace('<inertial pos="0.1 0 0" mass="1.2" diaginertia="0.01 0.02 0.02"/>', "")
    with pytest.raises(ValueError, match="visual"):
        _model(tmp_path, no_inertial, extra=acts, name="bad2.xml")
    with pytest.raises(ValueError, match="noslip"):
        _model(tmp_path, body % "", extra=acts, name="ns.xml", head=HEAD + '<option noslip_iterations="5"/>')


# ------------------------------------------------------------------------------------------ solver parameters per element
MIXED = """
<geom name="floor" type="plane" pos="0 0 0" size="5 5 0.1" contype="1" conaffinity="1" condim="3" friction="0.7"
      solref="0.03 1" solimp="0.8 0.9 0.002" solmix="3"/>
<body name="ball" pos="0 0 0.1"><freejoint name="ball_free"/>
  <geom name="ball" type="sphere" size="0.1" mass="0.5" contype="1" conaffinity="1" condim="3" solref="0.01 1" solimp="0.9 0.95 0.001"/>
  <site name="finger"/></body>
<body name="pa" pos="1 0 1"><joint name="ja" type="hinge" axis="0 1 0" limited="true" range="-0.3 0.3" frictionloss="0.05"
    solreflimit="%s" solimplimit="0.9 0.95 0.001" solreffriction="%s"/>
  <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.03" mass="0.4"/></body>
<body name="pb" pos="2 0 1"><joint name="jb" type="hinge" axis="0 1 0" limited="true" range="-0.3 0.3" frictionloss="0.05"
    solreflimit="%s" solimplimit="0.9 0.95 0.001" solreffriction="%s"/>
  <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.03" mass="0.4"/></body>"""
MIXED_ACT = ('<actuator><motor joint="ja" ctrlrange="-1 1" ctrllimited="true"/><motor joint="jb" ctrlrange="-1 1" ctrllimited="true"/></actuator>')


def test_solver_parameters_per_geom_and_per_joint(tmp_path):
    """MuJoCo mj_contactParam [EXT]: a contact's solref / solimp are its two geoms' averaged with weights solmix_1 : solmix_2
    (a higher priority wins outright); joint-limit and friction-loss rows carry their joint's own sets.  The mixed model
    equals, trajectory for trajectory, models that give every element the resulting numbers directly."""
    from mjmpc_amd.models.raw import mix_contact_solver
    r, i = mix_contact_solver(((0.01, 1.0), (0.9, 0.95, 0.001), 1.0, 0), ((0.03, 1.0), (0.8, 0.9, 0.002), 3.0, 0))
    np.testing.assert_allclose(r, (0.025, 1.0))
    np.testing.assert_allclose(i, (0.825, 0.9125, 0.00175, 0.5, 2.0))
    assert mix_contact_solver(((0.01, 1.0), (0.9, 0.95, 0.001), 1.0, 2), ((0.03, 1.0), (0.8, 0.9, 0.002), 3.0, 0))[0] == (0.01, 1.0)
    assert mix_contact_solver(((0.01, 1.0), (0.9,), 0.0, 0), ((0.03, 1.0), (0.8,), 0.0, 0))[0] == (0.02, 1.0)
    raw, ref = _model(tmp_path, MIXED % ("0.01 1", "0.02 1", "0.05 1", "0.08 1"), extra=MIXED_ACT, name="mixed.xml")
    assert raw.plane.solmix == 3.0 and {tuple(b.joint.solref_limit or ()) for b in raw.bodies[1:]} == {(), (0.05, 1.0)}
    tm = compile_tree(raw)
    tab = tm.field("soltab").reshape(8, 7)
    assert np.count_nonzero(tab[:, 0]) == 5         # the ball's own set (= pendulum a's limit set), the mixed one, b's limit set, two friction sets
    # the same numbers given directly
    direct = 'solref="0.025 1" solimp="0.825 0.9125 0.00175"'
    same = MIXED.replace('solref="0.03 1" solimp="0.8 0.9 0.002" solmix="3"', direct).replace('solref="0.01 1" solimp="0.9 0.95 0.001"', direct)
    _, ref_a = _model(tmp_path, same % ("0.01 1", "0.02 1", "0.01 1", "0.02 1"), extra=MIXED_ACT, name="a.xml")
    _, ref_b = _model(tmp_path, same % ("0.05 1", "0.08 1", "0.05 1", "0.08 1"), extra=MIXED_ACT, name="b.xml")
    q0 = raw.qpos0.copy()
    q0[2] = 0.12                                    # the ball drops 2 cm onto the floor, with some spin and drift
    v0 = np.zeros(raw.nv)
    v0[0], v0[4], v0[6], v0[7] = 0.3, 2.0, 3.0, -3.0    # both pendulums swing into their limits
    u = np.array([0.3, -0.3])
    qm, vm, _ = _run(ref, q0, v0, u, 400)
    qa, va, _ = _run(ref_a, q0, v0, u, 400)
    qb, vb, _ = _run(ref_b, q0, v0, u, 400)
    np.testing.assert_allclose(np.r_[qm[:7], vm[:6]], np.r_[qa[:7], va[:6]], rtol=0, atol=1e-11)      # the ball: same in all three
    np.testing.assert_allclose(np.r_[qm[7], vm[6]], np.r_[qa[7], va[6]], rtol=0, atol=1e-11)          # pendulum a: the stiff sets
    np.testing.assert_allclose(np.r_[qm[8], vm[7]], np.r_[qb[8], vb[7]], rtol=0, atol=1e-11)          # pendulum b: the soft sets
    assert abs(qa[8] - qb[8]) > 1e-4                # (and the sets do matter)
    nine = "".join('<body name="s%d" pos="%d 1 0.2"><joint type="slide" axis="0 0 1"/><geom type="sphere" size="0.1" contype="1" conaffinity="1" solref="%g 1"/></body>'
                   % (k, k, 0.01 + 0.002 * k) for k in range(9))
    rr, _ = _model(tmp_path, MIXED % ("0.01 1", "0.02 1", "0.05 1", "0.08 1") + nine, extra=MIXED_ACT, name="nine.xml", self_collision=False)
    with pytest.raises(NotImplementedError, match="distinct solref"):
        compile_tree(rr)


def test_ball_joint_limit_keeps_the_rotation_angle_below_its_range(tmp_path):
    """mj_instantiateLimit for a ball joint [EXT]: the angle of the joint quaternion's rotation stays below max(range) - a
    soft row over the joint's three dofs, J = -axis.  A spinning, swinging body: unlimited it turns through more than 2 rad,
    limited to 0.6 rad it is turned back there (soft row: 0.1 rad of overshoot at this speed), and inside the cone the two models agree."""
    body = """
    <body name="bob" pos="0 0 1"><joint name="bj" type="ball" damping="0.02"%s/>
      <geom type="capsule" fromto="0 0 0 0.3 0 -0.1" size="0.03" mass="0.5"/><site name="finger" pos="0.3 0 -0.1"/>
      <body name="tip" pos="0.3 0 -0.1"><joint name="h" type="hinge" axis="0 1 0" damping="0.05"/>
        <geom type="capsule" fromto="0 0 0 0.1 0 0" size="0.02" mass="0.1"/></body></body>"""
    BALL_LIMIT_ACT = '<actuator><motor joint="h" gear="0.2" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    raw_f, ref_f = _model(tmp_path, body % "", name="free.xml", timestep="0.001", extra=BALL_LIMIT_ACT)
    raw_l, ref_l = _model(tmp_path, body % ' limited="true" range="0 0.6"', name="lim.xml", timestep="0.001", extra=BALL_LIMIT_ACT)
    assert raw_l.bodies[0].joint.limited and max(raw_l.bodies[0].joint.range) == 0.6
    tm = compile_tree(raw_l)
    assert tm.general and int(tm.field("n_sphere")[0]) == 1 and tm.field("pext")[0] == 2.0

    def angle(q):
        return 2 * np.arctan2(np.linalg.norm(q[1:4]), q[0])

    qf, vf = raw_f.qpos0.copy(), np.array([1.0, 3.0, -2.0, 0.0])
    ql, vl = qf.copy(), vf.copy()
    worst_f = worst_l = 0.0
    agree_until = None
    for k in range(1500):
        qf, vf, _, _ = ref_f.step(qf, vf, np.zeros(1))
        ql, vl, _, dg = ref_l.step(ql, vl, np.zeros(1))
        worst_f, worst_l = max(worst_f, angle(qf)), max(worst_l, angle(ql))
        if agree_until is None and angle(qf) > 0.6:
            agree_until = k
            np.testing.assert_allclose(ql, qf, rtol=0, atol=1e-6)     # (identical until the cone is reached)
    assert agree_until is not None and worst_f > 2.0
    assert 0.6 < worst_l < 0.75, worst_l                # (a soft row with a 20 ms time constant, met at 3.7 rad/s)


def test_pair_elements_override_the_geoms_contact_parameters(tmp_path):
    """<contact><pair geom1 geom2 condim friction margin solref solimp>: what the element gives replaces mj_contactParam's
    mix of the two geoms' values, the rest stays mixed; the model with the override equals the one whose geoms carry the
    numbers themselves."""
    body = """
    <geom name="post" type="sphere" pos="0 0 0.2" size="0.2" friction="0.9"%s/>
    <body name="ball" pos="0.05 0 0.52"><freejoint/>
      <geom name="ball" type="sphere" size="0.1" mass="0.4" friction="0.3"%s/><site name="finger"/></body>
    <body name="arm" pos="1 0 1"><joint name="h" type="hinge" axis="0 1 0"/><geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.02"/></body>"""
    act = '<actuator><motor joint="h" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    over = '<contact><pair geom1="ball" geom2="post" condim="1" margin="0.004" solref="0.012 1"/></contact>'
    plain = '<contact><pair geom1="ball" geom2="post"/></contact>'
    c3 = ' condim="3"'
    ra, refa = _model(tmp_path, body % (c3, c3), extra=act + over, name="over.xml")
    assert ra.pair_params == {("ball", "post"): {"condim": 1, "margin": 0.004, "solref": (0.012, 1.0)}}
    assert ra.pair_contact(ra.bodies[0].geoms[0], ra.world_geoms[0], ("ball", "post"))[:3] == (1, 0.9, 0.004)
    direct = ' condim="1" margin="0.004" solref="0.012 1"'
    rb, refb = _model(tmp_path, body % (direct, direct), extra=act + plain, name="direct.xml")
    q, v = ra.qpos0.copy(), np.zeros(ra.nv)
    qa, va, _ = _run(refa, q, v, np.zeros(1), 300)
    qb, vb, _ = _run(refb, q, v, np.zeros(1), 300)
    np.testing.assert_allclose(np.r_[qa, va], np.r_[qb, vb], rtol=0, atol=1e-12)
    rc, refc = _model(tmp_path, body % (c3, c3), extra=act + plain, name="plain.xml")      # (friction cone: it does not slide off as fast)
    qc, vc, _ = _run(refc, q, v, np.zeros(1), 300)
    assert np.abs(qc[:3] - qa[:3]).max() > 1e-3
    ta, tb = compile_tree(ra), compile_tree(rb)
    np.testing.assert_array_equal(ta.field("spheres")[:21], tb.field("spheres")[:21])        # ([21]: the record's solver set, numbered per model)
    sa, sb = int(ta.field("spheres")[21]), int(tb.field("spheres")[21])
    np.testing.assert_array_equal(ta.field("soltab").reshape(8, 7)[sa], tb.field("soltab").reshape(8, 7)[sb])


# ------------------------------------------------------------------------------------------ round 5: joint margin / ref, geom gap
def test_joint_margin_ref_and_geom_gap_on_the_oracle(tmp_path):
    """MJCF joint ``margin`` (the limit row exists while dist < margin and acts on dist - margin), joint ``ref`` (qpos0: the
    kinematics turn by qpos - ref while range, springs and actuator lengths are stated on qpos) and geom ``gap`` (a contact
    enters the solver while dist < margin - gap): [EXT] semantics, held to what they must mean mechanically."""
    pend = """<body name="a" pos="0 0 1"><joint name="j" type="hinge" axis="0 1 0" limited="true" range="-0.5 0.5"%s/>
      <geom type="capsule" size="0.02 0.1" pos="0 0 -0.1" density="500"/><site name="finger" pos="0 0 -0.2"/></body>"""
    act = '<actuator><motor joint="j" gear="1" ctrlrange="-10 10" ctrllimited="true"/></actuator>'
    # 1. margin: at q = 0.45 (0.05 inside the limit) a margin of 0.1 already pushes back, no margin does not
    _, r0 = _model(tmp_path, pend % "", gravity="0 0 0", extra=act, name="m0.xml")
    _, r1 = _model(tmp_path, pend % ' margin="0.1"', gravity="0 0 0", extra=act, name="m1.xml")
    q0, v0, _, d0 = r0.step([0.45], [0.0], [0.0])
    q1, v1, _, d1 = r1.step([0.45], [0.0], [0.0])
    assert d0[0] == 0 and v0[0] == 0.0
    assert d1[0] == 1 and v1[0] < 0.0
    # ... and a row at the same VIOLATION pos - margin gives the same push: q = 0.55 without margin = q = 0.45 with margin 0.1
    q2, v2, _, d2 = r0.step([0.55], [0.0], [0.0])
    np.testing.assert_allclose(v1, v2, rtol=1e-12)
    # 2. ref: the model with ref = 0.3 in coordinates qpos - 0.3 IS the model without ref whose range / springref moved by -0.3
    spring = ' stiffness="2" springref="0.1" ref="0.3"'
    _, ra = _model(tmp_path, pend % spring, extra=act, name="r1.xml")
    _, rb = _model(tmp_path, (pend % ' stiffness="2" springref="-0.2"').replace('range="-0.5 0.5"', 'range="-0.8 0.2"'), extra=act, name="r0.xml")
    assert ra.qpos0[0] == 0.3
    qa, va, qb, vb = np.array([0.3 + 0.15]), np.array([0.4]), np.array([0.15]), np.array([0.4])
    for k in range(400):
        qa, va, sa, _ = ra.step(qa, va, [6.0])
        qb, vb, sb, _ = rb.step(qb, vb, [6.0])
    np.testing.assert_allclose(qa - 0.3, qb, rtol=0, atol=1e-12)
    np.testing.assert_allclose(sa, sb, rtol=0, atol=1e-12)
    assert qb[0] > 0.19                                     # (it did run into the upper limit)
    # ... position servos read qpos itself
    servo = '<actuator><position joint="j" kp="30" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    _, rc = _model(tmp_path, pend % ' ref="0.3" damping="1"', gravity="0 0 0", extra=servo, name="r2.xml")
    qc, vc = np.array([0.3]), np.array([0.0])
    for k in range(3000):
        qc, vc, _, _ = rc.step(qc, vc, [0.4])
    assert abs(qc[0] - 0.4) < 1e-6                          # the set point is a qpos
    # 3. gap: a sphere resting 1.5 mm above the plane - inside the margin of 4 mm, outside margin - gap = 1 mm: no force
    ball = """<body name="s" pos="0 0 0.1015"><joint type="slide" axis="0 0 1"/><geom type="sphere" size="0.1" contype="1" conaffinity="1"
      margin="0.004"%s/><site name="finger"/></body><geom type="plane" size="1 1 0.1" contype="1" conaffinity="1"/>"""
    a1 = '<actuator><motor joint="s_joint0" gear="1" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    _, g0 = _model(tmp_path, ball % "", gravity="0 0 0", extra=a1, name="g0.xml")
    _, g1 = _model(tmp_path, ball % ' gap="0.003"', gravity="0 0 0", extra=a1, name="g1.xml")
    assert g0.step([0.0], [0.0], [0.0])[3][0] >= 1 and g1.step([0.0], [0.0], [0.0])[3][0] == 0
    assert g0.step([0.0], [0.0], [0.0])[1][0] > 0.0 and g1.step([0.0], [0.0], [0.0])[1][0] == 0.0
    # ... and once inside margin - gap the row acts on dist - (margin - gap): the same push as margin 1 mm without a gap
    _, g2 = _model(tmp_path, (ball % "").replace('margin="0.004"', 'margin="0.001"'), gravity="0 0 0", extra=a1, name="g2.xml")
    np.testing.assert_allclose(g1.step([-0.001], [0.0], [0.0])[1], g2.step([-0.001], [0.0], [0.0])[1], rtol=1e-12)


def test_two_compilers_agree_on_joint_ref(tmp_path):
    """``compile_tree`` moves range / spring reference / actuator constants by the joint's ref (the kernel's coordinate is qpos -
    ref); the oracle keeps qpos and subtracts qpos0 in its kinematics: the same constants at qpos0."""
    body = """<body name="a" pos="0 0 1"><joint name="j" type="hinge" axis="0 1 0" limited="true" range="-0.5 0.9" ref="0.2" margin="0.05" stiffness="1" springref="0.4"/>
      <geom type="capsule" size="0.02 0.1" pos="0 0 -0.1" density="500"/>
      <body name="b" pos="0 0 -0.2"><joint name="k" type="slide" axis="0 0 1" limited="true" range="-0.1 0.1" ref="-0.03"/>
      <geom type="sphere" size="0.03"/><site name="finger"/></body></body>"""
    act = '<actuator><position joint="j" kp="5" ctrlrange="-1 1" ctrllimited="true"/><motor joint="k" gear="1" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    raw, ref = _model(tmp_path, body, extra=act, name="cr.xml")
    m = compile_tree(raw)
    assert m.general
    f = m.field
    np.testing.assert_allclose(f("range_lo")[:2], [-0.7, -0.07], atol=1e-15)
    np.testing.assert_allclose(f("range_hi")[:2], [0.7, 0.13], atol=1e-15)
    np.testing.assert_allclose(f("springref")[0], 0.2, atol=1e-15)
    np.testing.assert_allclose(f("qoff")[:2], [0.2, -0.03], atol=1e-15)
    np.testing.assert_allclose(f("jmargin")[:2], [0.05, 0.0], atol=1e-15)
    np.testing.assert_allclose(f("tau0")[0], -5.0 * 0.2, atol=1e-15)           # gear^2 b1 ref, b1 = -kp
    dof, _ = ref.invweight0()
    np.testing.assert_allclose(f("dof_invweight0")[:2], dof, rtol=1e-12)


# ------------------------------------------------------------------------------------------ round 5: colliders
def test_cylinder_rests_on_the_plane_lying_and_standing(tmp_path):
    """A cylinder on the plane (mjc_PlaneCylinder as restated: the lowest rim point, the point under it on the other cap, two
    more rim points 120 degrees apart): standing it rests on a triangle of three points, lying on the two rim points of its
    line of contact - at its radius / half height above the plane, without tipping; its mass and inertia are a cylinder's."""
    for name, euler, z_rest, rows in (("stand", "0 0 0", 0.15, 3), ("lie", "0 1.5707963267948966 0", 0.05, 2)):
        body = """<body name="c" pos="0 0 %g" euler="%s"><freejoint/><geom type="cylinder" size="0.05 0.15" density="800" contype="1" conaffinity="1"
          margin="0.002" friction="0.7 0.005 0.0001"/><site name="finger"/></body><geom type="plane" size="1 1 0.1" contype="1" conaffinity="1" margin="0.002"/>"""
        raw, ref = _model(tmp_path, body % (z_rest + 0.01, euler), name="cyl_%s.xml" % name)
        mass, _, inertia = ref.inertial()
        m = 800 * np.pi * 0.05 ** 2 * 0.3
        np.testing.assert_allclose(mass[1], m, rtol=1e-12)
        np.testing.assert_allclose(np.sort(np.diag(inertia[1])), np.sort([0.5 * m * 0.05 ** 2] + 2 * [m * (3 * 0.05 ** 2 + 0.3 ** 2) / 12]), rtol=1e-12)
        q, v = raw.qpos0.copy(), np.zeros(6)
        for k in range(1500):
            q, v, _, d = ref.step(q, v, [])
        assert abs(q[2] - z_rest) < 2e-3 and np.abs(v).max() < 1e-3, (name, q, v)
        assert d[0] == 4 * rows                                 # (pyramidal cones: four rows per contact point)
        up = np.array([2 * (q[4] * q[6] + q[3] * q[5]), 2 * (q[5] * q[6] - q[3] * q[4]), 1 - 2 * (q[4] ** 2 + q[5] ** 2)])     # the body's z axis
        want = np.array([0, 0, 1.0]) if name == "stand" else None
        if want is not None:
            assert up @ want > 1 - 1e-6
        else:
            assert abs(up[2]) < 1e-3                            # still lying
    assert ref.newton_stats()["fails"] == 0


def test_a_box_on_the_plane_keeps_four_corners_below_its_centre(tmp_path):
    """mjc_PlaneBox: corners on the upper side of the box centre are skipped and four contacts are kept - a thin slab pushed
    INTO the plane (all eight corners within the margin) gets four contacts, not eight."""
    body = """<body name="b" pos="0 0 0.004"><freejoint/><geom type="box" size="0.1 0.08 0.005" contype="1" conaffinity="1" condim="1" margin="0.02"/>
      <site name="finger"/></body><geom type="plane" size="1 1 0.1" contype="1" conaffinity="1" condim="1" margin="0.02"/>"""
    raw, ref = _model(tmp_path, body, name="slab.xml")
    q, v, _, d = ref.step(raw.qpos0.copy(), np.zeros(6), [])
    assert d[0] == 4


def test_capsule_on_a_box_rests_and_a_tilted_one_touches_once(tmp_path):
    """A capsule against a box (round 5; three candidate contacts: where its axis comes nearest to the box, its two ends): lying
    along the top of a static box it rests at its radius above the face on more than one contact and stays level; tilted,
    only its lower end touches; hanging over the edge across the box it is held by the contact at the edge region."""
    world = '<geom name="slab" type="box" pos="0 0 0.1" size="0.3 0.2 0.1" contype="1" conaffinity="1" friction="0.8 0.005 0.0001"/>'
    body = """<body name="c" pos="0 0 %g" euler="0 %g 0"><freejoint/><geom name="cap" type="capsule" size="0.03 0.12" density="600" contype="1" conaffinity="1"
      margin="0.002" friction="0.8 0.005 0.0001"/><site name="finger"/></body>"""
    # lying flat on the top face (z = 0.2): rest height 0.23
    raw, ref = _model(tmp_path, world + body % (0.235, np.pi / 2), name="cb0.xml")
    q, v = raw.qpos0.copy(), np.zeros(6)
    for k in range(2000):
        q, v, _, d = ref.step(q, v, [])
    assert abs(q[2] - 0.23) < 2e-3 and np.abs(v).max() < 1e-3 and d[0] >= 8          # two or three contact points
    axis_z = 1 - 2 * (q[4] ** 2 + q[5] ** 2)
    assert abs(axis_z) < 2e-3                                   # level
    # tilted by 0.4 rad: one contact, at the lower end
    raw, ref = _model(tmp_path, world + body % (0.30, np.pi / 2 - 0.4), name="cb1.xml")
    q, v, _, d = ref.step(raw.qpos0.copy(), np.zeros(6), [])
    assert d[0] == 0
    q0 = raw.qpos0.copy()
    q0[2] = 0.2 + 0.03 + 0.12 * np.sin(0.4) - 0.001             # the lower end 1 mm into the face
    q, v, _, d = ref.step(q0, np.zeros(6), [])
    assert d[0] == 4 and v[2] > -9.81 * 0.002                   # (pushed up against gravity)
    assert ref.newton_stats()["fails"] == 0


def test_segment_box_closest_parameter_against_brute_force():
    """The closed-form piecewise minimisation behind the capsule-box contact against a dense scan."""
    import ctypes
    from oracle.physics_ref import _lib, _p, _c
    L = _lib()
    L.or_seg_box_param.restype = ctypes.c_double
    L.or_seg_box_param.argtypes = [ctypes.POINTER(ctypes.c_double)] * 3
    rs = np.random.RandomState(0)
    ts = np.linspace(0, 1, 20001)
    for k in range(300):
        h = rs.uniform(0.05, 0.5, 3)
        a, b = rs.uniform(-1, 1, 3), rs.uniform(-1.5, 1.5, 3)
        if k % 5 == 0:
            b[rs.randint(3)] = 0.0                              # parallel to a pair of faces
        t = L.or_seg_box_param(_p(_c(h)), _p(_c(a)), _p(_c(b)))
        pts = a[None] + ts[:, None] * b[None]
        f = (np.maximum(np.abs(pts) - h[None], 0.0) ** 2).sum(1)
        ft = (np.maximum(np.abs(a + t * b) - h, 0.0) ** 2).sum()
        assert 0.0 <= t <= 1.0 and ft <= f.min() + 1e-12, (k, t, ft, f.min())


def test_a_box_rests_on_a_static_box_and_slides_off_a_tilted_one(tmp_path):
    """Two boxes (round 5: separating axes, then the incident face clipped against the reference face - up to four contacts - or
    the closest points of an edge pair): a box dropped on a static slab comes to rest on it, level, at its half height above
    the slab's top; rotated about the vertical (an octagonal overlap) it still rests on four contacts; on a slab tilted beyond
    the friction angle it slides."""
    slab = '<geom name="slab" type="box" pos="0 0 0.1" size="0.3 0.3 0.1" contype="1" conaffinity="1" friction="0.6 0.005 0.0001"%s/>'
    box = """<body name="b" pos="0 0 0.255" euler="0 0 %g"><freejoint/><geom name="cube" type="box" size="0.05 0.04 0.05" density="600" contype="1" conaffinity="1"
      margin="0.002" friction="0.6 0.005 0.0001"/><site name="finger"/></body>"""
    for yaw in (0.0, 0.6):
        raw, ref = _model(tmp_path, slab % "" + box % yaw, name="bb%d.xml" % int(10 * yaw))
        q, v = raw.qpos0.copy(), np.zeros(6)
        for k in range(1500):
            q, v, _, d = ref.step(q, v, [])
        assert abs(q[2] - 0.25) < 2e-3 and np.abs(v).max() < 1e-3 and d[0] == 16, (yaw, q, d[0])
        assert abs(1 - 2 * (q[4] ** 2 + q[5] ** 2) - 1) < 1e-5      # level
    raw, ref = _model(tmp_path, slab % ' euler="0.8 0 0"' + box % 0.0, name="bbt.xml")       # 0.8 rad > atan(0.6)
    q, v = raw.qpos0.copy(), np.zeros(6)
    q[1], q[2] = -0.1, 0.33
    q[3:7] = [np.cos(0.4), np.sin(0.4), 0, 0]
    y0 = q[1]
    for k in range(600):
        q, v, _, d = ref.step(q, v, [])
    assert q[1] < y0 - 0.1 and ref.newton_stats()["fails"] == 0       # it slid down the slope


def test_box_box_contacts_lie_between_the_boxes():
    """Random pairs of boxes: no contact when a separating axis leaves more than the margin; otherwise every contact point lies
    within its own distance (+ margin) of BOTH boxes, the normal is a unit vector from box 1 towards box 0, and pushing box 0
    along it increases every contact's distance."""
    import ctypes
    from oracle.physics_ref import _lib, _p, _c
    L = _lib()
    dp = ctypes.POINTER(ctypes.c_double)
    L.or_box_box.restype = ctypes.c_int
    L.or_box_box.argtypes = [dp] * 6 + [ctypes.c_double] + [dp] * 3

    def quat2mat(q):
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def call(c0, R0, h0, c1, R1, h1, margin):
        n, pos, d = np.zeros(3), np.zeros((4, 3)), np.zeros(4)
        k = L.or_box_box(_p(_c(c0)), _p(_c(R0)), _p(_c(h0)), _p(_c(c1)), _p(_c(R1)), _p(_c(h1)), margin, _p(n), _p(pos), _p(d))
        return k, n, pos[:k], d[:k]

    def outside(p, c, R, h):            # distance of a point from a solid box (0 inside)
        loc = R.T @ (p - c)
        return np.linalg.norm(np.maximum(np.abs(loc) - h, 0.0))

    rs = np.random.RandomState(1)
    hit = edge = 0
    for trial in range(400):
        h0, h1 = rs.uniform(0.03, 0.2, 3), rs.uniform(0.03, 0.2, 3)
        R0, R1 = quat2mat(rs.standard_normal(4)), quat2mat(rs.standard_normal(4))
        if trial % 4 == 0:
            R1 = R0 @ quat2mat(np.r_[np.cos(0.3), 0, 0, np.sin(0.3)])       # a shared axis: face contacts with edge-parallel cases
        c0, c1 = np.zeros(3), rs.standard_normal(3) * 0.18
        margin = 0.004
        k, n, pos, d = call(c0, R0, h0, c1, R1, h1, margin)
        # brute-force separation over the 15 axes
        axes = [R0[:, i] for i in range(3)] + [R1[:, i] for i in range(3)] + [np.cross(R0[:, i], R1[:, j]) for i in range(3) for j in range(3)]
        sep = -np.inf
        for a in axes:
            if np.linalg.norm(a) < 1e-9:
                continue
            a = a / np.linalg.norm(a)
            sep = max(sep, abs(a @ (c1 - c0)) - sum(h0[i] * abs(a @ R0[:, i]) for i in range(3)) - sum(h1[i] * abs(a @ R1[:, i]) for i in range(3)))
        if sep >= margin:
            assert k == 0
            continue
        if k == 0:
            continue                    # (an edge pair whose closest points are further apart than the axis' separation says)
        hit += 1
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and n @ (c0 - c1) > 0
        for p, dd in zip(pos, d):
            assert dd < margin and dd >= sep - 0.06 * abs(sep) - 1e-9      # (an edge axis wins only by 5 %)
            assert outside(p, c0, R0, h0) <= abs(dd) / 2 + margin + 1e-9 and outside(p, c1, R1, h1) <= abs(dd) / 2 + margin + 1e-9
        k2, n2, pos2, d2 = call(c0 + 1e-4 * n, R0, h0, c1, R1, h1, margin + 1e-3)
        if k2 == k and np.allclose(n2, n):
            assert np.all(d2 > d - 1e-9)
    assert hit > 80


# ------------------------------------------------------------------------------------------ elliptic friction cones (round 5)
def _cone_cost(D0, Dt, fr, r):
    """The elliptic cone's cost from its definition: D0 / (2 mu^2) times the squared distance of U = (mu r0, fr r1, fr r2)
    from the cone N >= mu T - the distance computed here by projecting U onto the cone numerically (no zone formulas)."""
    mu = fr * np.sqrt(D0 / Dt)
    N, T = mu * r[0], fr * np.hypot(r[1], r[2])
    # nearest point of the 2-D cone {(n, t): n >= mu t, t >= 0} to (N, T): the point itself, the apex, or the foot on the edge
    if N >= mu * T:
        d2 = 0.0
    else:
        e = np.array([mu, 1.0]) / np.hypot(mu, 1.0)             # the edge's direction (n = mu t)
        s = max(0.0, e @ np.array([N, T]))
        d2 = (N - s * e[0]) ** 2 + (T - s * e[1]) ** 2
    return 0.5 * D0 / mu ** 2 * d2


def test_elliptic_cone_rows_against_an_independent_minimiser(tmp_path):
    """solve_rows with ELLIPTIC contacts (groups of three rows: kind 3 then 4, 4) beside the other row kinds, random
    problems: stationarity M a - fs = J' f, forces inside the friction cone |f_t| <= fr f_n, and the cost - computed here
    from the cone's definition as a distance, not from the oracle's zone formulas - not above scipy's minimiser's."""
    from scipy.optimize import minimize
    raw, ref = _model(tmp_path, BLOCK % (0.0, 1.0), extra=ACT)
    rs = np.random.RandomState(5)
    zones = set()
    for trial in range(60):
        nv, ng, nx = rs.randint(3, 8), rs.randint(1, 4), rs.randint(0, 4)
        nc = 3 * ng + nx
        A = rs.standard_normal((nv, nv))
        M = A @ A.T + nv * np.eye(nv)
        fs = 8 * rs.standard_normal(nv)
        J = rs.standard_normal((nc, nv)) * (rs.uniform(size=(nc, nv)) < 0.8)
        aref, D = 3 * rs.standard_normal(nc), rs.uniform(0.5, 20, nc)
        kind = np.r_[np.tile([3, 4, 4], ng), rs.randint(0, 3, nx)].astype(int)
        fl = np.where(kind == 2, rs.uniform(0.1, 3.0, nc), 0.0)
        for g in range(ng):
            D[3 * g + 1] = D[3 * g + 2] = D[3 * g] * rs.choice([1.0, 4.0, 0.3])       # impratio
            fl[3 * g] = rs.uniform(0.2, 1.5)                                          # friction coefficient
            aref[3 * g] = abs(aref[3 * g]) * rs.choice([1.0, 1.0, -1.0])              # (mostly: pushed into the surface)

        def cost(a):
            r = J @ a - aref
            c = 0.5 * a @ M @ a - fs @ a
            for g in range(ng):
                c += _cone_cost(D[3 * g], D[3 * g + 1], fl[3 * g], r[3 * g:3 * g + 3])
            for i in range(3 * ng, nc):
                if kind[i] == 0:
                    c += 0.5 * D[i] * min(0.0, r[i]) ** 2
                elif kind[i] == 1:
                    c += 0.5 * D[i] * r[i] ** 2
                else:
                    Rf = fl[i] / D[i]
                    c += 0.5 * D[i] * r[i] ** 2 if abs(r[i]) < Rf else fl[i] * abs(r[i]) - 0.5 * Rf * fl[i]
            return c

        a, f = ref.solve_rows(M, fs, J, aref, D, kind, fl)
        np.testing.assert_allclose(M @ a - fs, J.T @ f, rtol=0, atol=1e-9)
        r = J @ a - aref
        for g in range(ng):
            fn, ft = f[3 * g], np.hypot(f[3 * g + 1], f[3 * g + 2])
            assert fn >= -1e-12 and ft <= fl[3 * g] * fn + 1e-9
            zones.add(0 if fn == 0 else (2 if ft > fl[3 * g] * fn - 1e-9 else 1))
            # the force is the cost's negative gradient (central differences of the cost defined by distance)
            for k in range(3):
                e = np.zeros(3); e[k] = 1e-6
                rr = r[3 * g:3 * g + 3]
                num = (_cone_cost(D[3 * g], D[3 * g + 1], fl[3 * g], rr + e) - _cone_cost(D[3 * g], D[3 * g + 1], fl[3 * g], rr - e)) / 2e-6
                assert abs(num + f[3 * g + k]) < 1e-5 * (1 + abs(num))
        best = minimize(cost, a + 0.01 * rs.standard_normal(nv), method="BFGS", options=dict(gtol=1e-9)).x
        assert cost(a) <= cost(best) + 1e-9
        np.testing.assert_allclose(a, best, rtol=0, atol=5e-5)
    assert zones == {0, 1, 2}                   # no force, sticking (inside the cone) and sliding (on the cone) all occurred
    assert ref.newton_stats()["fails"] == 0


ELL_SPHERE = """<geom name="floor" type="plane" pos="0 0 0" size="5 5 0.1" contype="1" conaffinity="1" friction="%g 0.005 0.0001" condim="3"/>
    <body name="ball" pos="0 0 0.0999"><freejoint/>
      <geom name="b" type="sphere" size="0.1" density="800" contype="1" conaffinity="1" friction="%g 0.005 0.0001" condim="3"/>
      <site name="finger"/></body>"""


def _opt_model(tmp_path, body, option, name="m.xml", **kw):
    xml = HEAD + '<option timestep="0.002" integrator="Euler" %s/>' % option + \
        '<default><geom contype="0" conaffinity="0"/></default><worldbody><site name="target" pos="0 0 0"/>' + \
        textwrap.dedent(body) + "</worldbody></mujoco>"
    (tmp_path / name).write_text(xml)
    raw = load_mjcf(str(tmp_path / name), task=TASK_REACH, **kw)
    return raw, RefArm(raw.to_flat())


def test_elliptic_cone_friction_is_isotropic_and_bounded_by_mu(tmp_path):
    """A sphere thrown into sliding on the plane (it skips: a sliding contact under MuJoCo's cone cost pushes out, and the
    contact exists only while the sphere touches) loses speed at mu g on average - under an ELLIPTIC cone the same along an
    axis of the contact frame and across it (three rows per contact), under the pyramidal one 1 / sqrt(2) of it across
    (four rows: |f_1| + |f_2| <= mu f_n)."""
    mu = 0.4

    def deceleration(cone, name, rows):
        raw, ref = _opt_model(tmp_path, ELL_SPHERE % (mu, mu), 'cone="%s"' % cone, name=name)
        assert raw.cone == cone
        out = []
        for d in (np.array([1.0, 0.0]), np.array([1.0, 1.0]) / np.sqrt(2)):
            q, v = ref.qpos0.copy(), np.zeros(6)
            for _ in range(300):
                q, v, _, diag = ref.step(q, v, np.zeros(0))
            assert diag[0] == rows
            v[:2] = 4.0 * d                                     # (sliding turns into rolling after 2 v / (7 mu g) = 0.29 s)
            for _ in range(100):
                q, v, _, _ = ref.step(q, v, np.zeros(0))
            out.append((4.0 - v[:2] @ d) / (100 * 0.002))
        return out

    axis, diagonal = deceleration("elliptic", "e.xml", 3)       # normal + two tangents
    assert abs(axis - diagonal) < 1e-9 * axis                   # isotropic
    assert 0.85 * mu * 9.81 < axis < 1.02 * mu * 9.81, axis     # (in the air part of the time)
    axis, diagonal = deceleration("pyramidal", "p.xml", 4)
    assert 0.85 * mu * 9.81 < axis < 1.02 * mu * 9.81 and abs(diagonal / axis - np.sqrt(0.5)) < 0.06, (axis, diagonal)


def test_elliptic_cone_box_sticks_below_the_friction_angle_and_impratio_hardens_it(tmp_path):
    """A box on the plane with gravity tilted by an angle: below atan(mu) it creeps at the soft friction rows' rate -
    which impratio 10 divides by about ten - and above it slides at g (sin - mu cos)."""
    mu = 0.5
    body = """<geom name="floor" type="plane" pos="0 0 0" size="5 5 0.1" contype="1" conaffinity="1" friction="%g 0.005 0.0001" condim="3"/>
    <body name="box" pos="0 0 0.0499"><freejoint/>
      <geom name="b" type="box" size="0.1 0.08 0.05" density="800" contype="1" conaffinity="1" friction="%g 0.005 0.0001" condim="3"/>
      <site name="finger"/></body>""" % (mu, mu)

    def slide(angle, option, name):
        g = 9.81 * np.array([np.sin(angle), 0.0, -np.cos(angle)])
        raw, ref = _opt_model(tmp_path, body, 'gravity="%.17g %.17g %.17g" %s' % (g[0], g[1], g[2], option), name=name)
        q, v = ref.qpos0.copy(), np.zeros(6)
        vs = []
        for _ in range(400):
            q, v, _, diag = ref.step(q, v, np.zeros(0))
            vs.append(v[0])
        return np.array(vs), diag

    lo, diag = slide(np.arctan(mu) * 0.8, 'cone="elliptic"', "a.xml")
    assert diag[0] == 12                                        # 4 corners x 3 rows
    assert 0 < lo[-1] < 0.02 and abs(lo[-1] - lo[-50]) < 1e-5   # creeping at a constant, small rate
    lo10, _ = slide(np.arctan(mu) * 0.8, 'cone="elliptic" impratio="10"', "b.xml")
    assert 0 < lo10[-1] < 0.2 * lo[-1]
    ang = np.arctan(mu) * 1.3
    hi, _ = slide(ang, 'cone="elliptic"', "c.xml")
    acc = (hi[-1] - hi[-101]) / (100 * 0.002)
    assert abs(acc - 9.81 * (np.sin(ang) - mu * np.cos(ang))) < 0.03 * 9.81 * np.sin(ang), acc


# ------------------------------------------------------------------------------------------ spheres / capsules on cylinders (round 5)
def test_cylinder_nearest_surface_point_against_brute_force():
    """cyl_point (the closed form behind sphere / capsule against cylinder) against a dense sampling of the cylinder's surface:
    distance and nearest point for points outside (side, caps, rims) and inside."""
    import ctypes
    from oracle.physics_ref import build
    L = ctypes.CDLL(build())
    dp = ctypes.POINTER(ctypes.c_double)
    L.or_cyl_point.restype = ctypes.c_int
    L.or_cyl_point.argtypes = [dp, dp, ctypes.c_double, dp, dp, dp, dp]
    _c = lambda a: np.ascontiguousarray(a, float)
    _p = lambda a: a.ctypes.data_as(dp)
    rs = np.random.RandomState(2)
    for trial in range(40):
        p0, d = rs.standard_normal(3), rs.standard_normal(3) * rs.uniform(0.2, 1.5)
        r = rs.uniform(0.1, 0.8)
        Lc = np.linalg.norm(d); u = d / Lc
        e1 = np.cross(u, [1.0, 0, 0]); e1 /= np.linalg.norm(e1); e2 = np.cross(u, e1)
        # surface samples: side, two caps
        th = np.linspace(0, 2 * np.pi, 400, endpoint=False)
        zz = np.linspace(0, Lc, 200)
        side = (p0 + zz[:, None, None] * u + r * (np.cos(th)[None, :, None] * e1 + np.sin(th)[None, :, None] * e2)).reshape(-1, 3)
        rr = np.linspace(0, r, 60)
        disk = (rr[:, None, None] * (np.cos(th)[None, :, None] * e1 + np.sin(th)[None, :, None] * e2)).reshape(-1, 3)
        surf = np.concatenate([side, p0 + disk, p0 + d + disk])
        for _ in range(6):
            c = p0 + 0.5 * d + rs.standard_normal(3) * rs.choice([0.2, 1.0, 2.0])
            q, n, ln = np.zeros(3), np.zeros(3), np.zeros(1)
            assert L.or_cyl_point(_p(_c(p0)), _p(_c(d)), r, _p(_c(c)), _p(q), _p(n), _p(ln)) == 1
            z, rho = (c - p0) @ u, np.linalg.norm((c - p0) - ((c - p0) @ u) * u)
            inside = 0 < z < Lc and rho < r
            brute = np.linalg.norm(surf - c, axis=1).min()
            assert abs(abs(ln[0]) - brute) < 0.01 + 0.02 * brute and (ln[0] < 0) == inside
            assert abs(np.linalg.norm(n) - 1) < 1e-12 and abs(np.linalg.norm(q - c) - abs(ln[0])) < 1e-12
            np.testing.assert_allclose(c - q, n * ln[0], atol=1e-12)            # outside: n points at c; inside: away from it


def test_sphere_and_capsule_rest_on_a_static_cylinder(tmp_path):
    """A ball dropped on the flat top of a standing cylinder rests on it; a capsule laid across a lying cylinder balances on
    its ridge for a while with one contact at the crossing, and a capsule lying ALONG the top cap rests on it with contacts
    at its ends."""
    cyl = '<geom name="post" type="cylinder" fromto="0 0 0 0 0 0.3" size="0.2" contype="1" conaffinity="1" condim="3" friction="0.8 0.005 0.0001"/>'
    ball = """<body name="ball" pos="0.05 0.02 0.381"><freejoint/>
      <geom name="b" type="sphere" size="0.08" density="900" contype="1" conaffinity="1" condim="3"/><site name="finger"/></body>"""
    raw, ref = _model(tmp_path, cyl + ball)
    assert [tuple(p) for p in raw.pairs] == [("b", "post")]
    q, v = ref.qpos0.copy(), np.zeros(6)
    for _ in range(600):
        q, v, _, diag = ref.step(q, v, np.zeros(0))
    assert diag[0] == 4 and 0.3795 < q[2] < 0.3801 and np.abs(v).max() < 1e-5          # on the cap: centre one radius above it
    # beside the post: the ball touches the curved side and is pushed away horizontally (no gravity)
    raw, ref = _model(tmp_path, cyl + ball.replace('pos="0.05 0.02 0.381"', 'pos="0.275 0 0.15"'), gravity="0 0 0", name="side.xml")
    q, v = ref.qpos0.copy(), np.zeros(6)
    q, v, _, diag = ref.step(q, v, np.zeros(0))
    assert diag[0] == 4 and v[0] > 0 and abs(v[1]) < 1e-12 and abs(v[2]) < 1e-12
    cap = """<body name="rod" pos="0 0 0.341"><freejoint/>
      <geom name="r" type="capsule" fromto="-0.12 0 0 0.12 0 0" size="0.04" density="900" contype="1" conaffinity="1" condim="3"/><site name="finger"/></body>"""
    raw, ref = _model(tmp_path, cyl + cap, name="rod.xml")
    q, v = ref.qpos0.copy(), np.zeros(6)
    for _ in range(600):
        q, v, _, diag = ref.step(q, v, np.zeros(0))
    assert diag[0] >= 8 and 0.3395 < q[2] < 0.3401 and np.abs(v).max() < 1e-5          # two or three contacts along the rod


# ------------------------------------------------------------------------------------------ direct solref (round 5)
def test_direct_solref_gives_stiffness_and_damping(tmp_path):
    """solref = (-stiffness, -damping), MuJoCo's direct format [EXT: k = -solref[0] / dmax^2, b = -solref[1] / dmax, no refsafe
    clamp]: a 2 kg mass on a vertical slide joint resting in its lower limit sinks in by m g (1 - d) dmax^2 / (d^2 k m) ... -
    with a flat impedance d = dmax = 0.9 that is g (1 - d) / k - and twice the stiffness halves it; the standard format's
    answer is different; a pair of mixed signs is refused."""
    def sink(solref):
        body = """<body name="m" pos="0 0 1"><joint name="z" type="slide" axis="0 0 1" limited="true" range="-0.1 0.5"
            solreflimit="%s" solimplimit="0.9 0.9 0.001"/><geom type="sphere" size="0.05" mass="2"/><site name="finger"/></body>""" % solref
        raw, ref = _model(tmp_path, body, extra='<actuator><motor joint="z" ctrlrange="-1 1" ctrllimited="true"/></actuator>', name="d%d.xml" % abs(hash(solref)))
        q, v = np.array([-0.1]), np.zeros(1)
        for _ in range(3000):
            q, v, _, diag = ref.step(q, v, np.zeros(1))
        assert abs(v[0]) < 1e-9 and diag[0] == 1
        return -0.1 - q[0]
    s1, s2 = sink("-2000 -200"), sink("-4000 -300")
    assert abs(s1 - 9.81 * 0.1 / 2000) < 1e-9 and abs(s2 - 9.81 * 0.1 / 4000) < 1e-9
    assert abs(sink("0.02 1") - s1) > 1e-5
    with pytest.raises(ValueError, match="both negative"):
        sink("-2000 1")


def test_two_compilers_agree_on_direct_solref(tmp_path):
    """compile_tree's K, B for direct-format sets = what the oracle works with (one substep from a state inside the limit and
    in contact agrees - through the CPU-side consistency check: the blob's solver table)."""
    body = """<geom name="floor" type="plane" size="2 2 0.1" contype="1" conaffinity="1" solref="-8000 -120"/>
    <body name="m" pos="0 0 0.2"><joint name="z" type="slide" axis="0 0 1" limited="true" range="-0.1 0.5" solreflimit="-2000 -200"/>
      <geom name="b" type="sphere" size="0.05" mass="2" contype="1" conaffinity="1" solref="-3000 -50"/><site name="finger"/></body>"""
    raw, ref = _model(tmp_path, body, extra='<actuator><motor joint="z" ctrlrange="-1 1" ctrllimited="true"/></actuator>')
    m = compile_tree(raw)
    tab = m.field("soltab").reshape(-1, 7)
    got = sorted((round(r[0], 6), round(r[1], 6)) for r in tab if r[0] != 0)
    # limit row: k = 2000 / 0.95^2, b = 200 / 0.95 (default solimp dmax 0.95); contact: the element-wise minimum of the two geoms' sets
    want = sorted([(round(2000 / 0.95 ** 2, 6), round(200 / 0.95, 6)), (round(8000 / 0.95 ** 2, 6), round(120 / 0.95, 6))])
    assert all(any(abs(g[0] - w[0]) < 1e-6 * w[0] and abs(g[1] - w[1]) < 1e-6 * w[1] for g in got) for w in want), (got, want)


# ------------------------------------------------------------------------------------------ <option><flag> (round 5)
def test_option_flags_switch_parts_of_the_model_off(tmp_path):
    """<option><flag>: gravity / contact / limit / frictionloss / equality / clampctrl / actuation / constraint "disable" take the
    corresponding parts out of the model; flags nothing here depends on (warmstart, energy, fwdinv, sensornoise, midphase) are
    accepted either way; override on, passive / filterparent / refsafe off and unknown flags are refused."""
    body = """<geom name="floor" type="plane" size="2 2 0.1" contype="1" conaffinity="1"/>
    <body name="a" pos="0 0 0.3"><joint name="j" type="hinge" axis="0 1 0" limited="true" range="-0.5 0.5" frictionloss="0.2"/>
      <geom name="g" type="capsule" fromto="0 0 0 0.3 0 0" size="0.03" contype="1" conaffinity="1"/><site name="finger" pos="0.3 0 0"/></body>"""
    acts = '<actuator><motor joint="j" gear="3" ctrlrange="-1 1" ctrllimited="true"/></actuator>'

    def load(flag, name):
        head = HEAD + ("<option><flag %s/></option>" % flag if flag else "")
        xml = head + '<default><geom contype="0" conaffinity="0"/></default><worldbody><site name="target" pos="0 0 0"/>' + \
            textwrap.dedent(body) + "</worldbody>" + acts + "</mujoco>"
        (tmp_path / name).write_text(xml)
        return load_mjcf(str(tmp_path / name), task=TASK_REACH)

    plain = load("", "p.xml")
    assert plain.plane is not None and plain.bodies[0].joint.limited and plain.actuators[0].ctrllimited
    same = load('warmstart="disable" energy="enable" fwdinv="enable" sensornoise="enable" midphase="disable" override="disable" passive="enable"', "s.xml")
    assert np.array_equal(plain.to_flat(), same.to_flat())
    off = load('gravity="disable" contact="disable" limit="disable" frictionloss="disable" clampctrl="disable"', "o.xml")
    assert tuple(off.gravity) == (0.0, 0.0, 0.0) and off.plane is None and not off.bodies[0].joint.limited
    assert off.bodies[0].joint.frictionloss == 0.0 and not off.actuators[0].ctrllimited
    cons = load('constraint="disable"', "c.xml")
    assert cons.plane is None and not cons.bodies[0].joint.limited and cons.bodies[0].joint.frictionloss == 0.0 and tuple(cons.gravity) != (0.0, 0.0, 0.0)
    assert load('actuation="disable"', "a.xml").actuators[0].gear == 0.0
    # ... and the oracle runs them: without gravity, limits and contacts the arm coasts at its initial speed
    ref = RefArm(off.to_flat())
    q, v = np.array([0.2]), np.array([1.5])
    for _ in range(100):
        q, v, _, diag = ref.step(q, v, np.zeros(1))
    assert diag[0] == 0 and abs(v[0] - 1.5) < 1e-12 and abs(q[0] - 0.2 - 1.5 * 0.2) < 1e-12
    for bad in ('override="enable"', 'passive="disable"', 'filterparent="disable"', 'refsafe="disable"', 'island="enable"', 'gravity="off"'):
        with pytest.raises(ValueError, match="flag"):
            load(bad, "bad.xml")


def test_option_collision_selects_derived_or_predefined_pairs(tmp_path):
    """<option collision>: all (default) = derived + explicit pairs, predefined = the <pair>s only (nothing meets the plane
    through its masks), dynamic = the derived ones only."""
    body = """<geom name="floor" type="plane" size="2 2 0.1" contype="1" conaffinity="1"/>
    <body name="a" pos="0 0 0.3"><freejoint/><geom name="ga" type="sphere" size="0.05" contype="1" conaffinity="1"/><site name="finger"/></body>
    <body name="b" pos="0.5 0 0.3"><freejoint/><geom name="gb" type="sphere" size="0.05" contype="1" conaffinity="1"/></body>
    <body name="c" pos="1 0 0.3"><freejoint/><geom name="gc" type="sphere" size="0.05" contype="2" conaffinity="2"/></body>"""
    pair = '<contact><pair geom1="gc" geom2="ga"/></contact>'

    def load(mode, name):
        xml = HEAD + ('<option collision="%s"/>' % mode if mode else "") + \
            '<default><geom contype="0" conaffinity="0"/></default><worldbody><site name="target" pos="0 0 0"/>' + \
            textwrap.dedent(body) + "</worldbody>" + pair + "</mujoco>"
        (tmp_path / name).write_text(xml)
        return load_mjcf(str(tmp_path / name), task=TASK_REACH)

    every = load("", "all.xml")
    assert sorted(tuple(p) for p in every.pairs) == [("gb", "ga"), ("gc", "ga")] and every.plane is not None
    assert [g.collide for b in every.bodies for g in b.geoms] == [True, True, False]
    pre = load("predefined", "pre.xml")
    assert [tuple(p) for p in pre.pairs] == [("gc", "ga")] and not any(g.collide for b in pre.bodies for g in b.geoms)
    dyn = load("dynamic", "dyn.xml")
    assert [tuple(p) for p in dyn.pairs] == [("gb", "ga")] and dyn.plane is not None
    with pytest.raises(ValueError, match="collision"):
        load("some", "bad.xml")


def test_xyaxes_and_zaxis_orientations(tmp_path):
    """Body / geom orientation as ``xyaxes`` (x axis + a vector in the xy plane, orthogonalised) and ``zaxis`` (the minimal rotation
    of (0, 0, 1) onto the direction): the same model as with the equivalent quaternions."""
    body = """<body name="a" pos="0 0 1" %s><joint name="j" type="hinge" axis="0 1 0"/>
      <geom name="g" type="box" size="0.1 0.05 0.02" %s density="700"/><site name="finger" pos="0.1 0 0"/></body>"""
    acts = '<actuator><motor joint="j" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    # xyaxes = a rotation by 90 degrees about z (x -> y, y -> -x) given with a non-orthogonal second vector; zaxis = z -> x,
    # i.e. 90 degrees about y
    a, _ = _model(tmp_path, body % ('xyaxes="0 2 0 -1 0.7 0"', 'zaxis="3 0 0"'), extra=acts, name="a.xml")
    q1 = "%.17g 0 0 %.17g" % (np.cos(np.pi / 4), np.sin(np.pi / 4))
    q2 = "%.17g 0 %.17g 0" % (np.cos(np.pi / 4), np.sin(np.pi / 4))
    b, _ = _model(tmp_path, body % ('quat="%s"' % q1, 'quat="%s"' % q2), extra=acts, name="b.xml")
    np.testing.assert_allclose(a.to_flat(), b.to_flat(), rtol=0, atol=1e-15)
    c, _ = _model(tmp_path, body % ('zaxis="0 0 -2"', ''), extra=acts, name="c.xml")         # the antipode: half a turn about x
    np.testing.assert_allclose(np.abs(c.bodies[0].quat), [0, 1, 0, 0], atol=1e-15)


def test_joint_ref_or_margin_alone_asks_for_the_general_instantiation(tmp_path):
    """A chain of plain hinges whose only general feature is a joint ``ref`` (or ``margin``) must be compiled for the general
    kernels - the others read neither (round-5 soak, seed 8666: such a model ran with ref = 0)."""
    body = """<body name="a" pos="0 0 1"><joint name="j" type="hinge" axis="0 1 0" limited="true" range="-1 1" %s/>
      <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.03"/><site name="finger" pos="0.3 0 0"/></body>"""
    acts = '<actuator><motor joint="j" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    plain, _ = _model(tmp_path, body % "", extra=acts, name="p.xml")
    assert not compile_tree(plain).general
    for attr in ('ref="0.3"', 'margin="0.05"'):
        raw, _ = _model(tmp_path, body % attr, extra=acts, name="r.xml")
        assert compile_tree(raw).general, attr

"""Self-consistency checks of the FP64 physics oracle (oracle/reacher_ref.c).

MuJoCo itself cannot run here (parity unpinned, see the oracle header), so the oracle is held to
properties that need no MuJoCo: two independent model compilers agree, M is SPD and consistent
with RNE, the bias force matches the Lagrangian derivative of the kinetic energy, energy is
conserved without damping, and the soft-constraint solution satisfies its optimality conditions.
CPU only."""
import copy

import numpy as np
import pytest

from mjmpc_amd.models.compile import compile_arm
from oracle.physics_ref import RefArm

# masses derived by hand from sawyer.xml (density 1000)
SURVEY_MASSES = [24.3117, 10.4720, 0.28484, 5.42867, 1.35717, 0.28484, 2.80911, 0.016755, 2.14466]


def test_masses_match_hand_derivation(raw_arm, ref_arm):
    """SURVEY appendix A derived its sanity masses with the textbook capsule volume (end caps 4/3 pi r^3); MuJoCo 2.0,
    which the reference pins, counts the caps pi r^3 (models/raw.py::MJ20_CAPSULE_CAP, the default): both are checked."""
    import dataclasses
    mass43, _, _ = RefArm(dataclasses.replace(raw_arm, capsule_cap_factor=4.0 / 3.0).to_flat()).inertial()
    np.testing.assert_allclose(mass43[1:], SURVEY_MASSES, rtol=2e-5)
    assert abs(mass43.sum() - 47.110) < 1e-3
    mass, _, _ = ref_arm.inertial()
    r, hl = 0.1, 0.3                                    # the pan link's capsule: (0,0,-.4) -> (0,0,.2), radius .1
    pan = 1000 * (np.pi * r * r * 2 * hl + np.pi * r ** 3) + 1000 * 4 / 3 * np.pi * (2 * 0.05 ** 3 + 2 * 0.03 ** 3)
    assert abs(mass[1] - pan) < 1e-9 and abs(mass.sum() - 44.414266) < 1e-5


def test_two_compilers_agree(raw_arm, ref_arm):
    am = compile_arm(raw_arm)
    mass, ipos, inertia = ref_arm.inertial()
    dof_w, body_w = ref_arm.invweight0()
    np.testing.assert_allclose(am.body_mass, mass[1:], rtol=1e-14)
    np.testing.assert_allclose(am.body_ipos, ipos[1:], atol=1e-15)
    np.testing.assert_allclose(am.body_inertia, inertia[1:], rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(am.dof_invweight0, dof_w, rtol=1e-12)
    np.testing.assert_allclose(am.body_invweight0, body_w[1:], rtol=1e-12, atol=1e-16)


def test_mass_matrix_spd_and_rne_consistent(ref_arm):
    rs = np.random.RandomState(0)
    arm_diag = 0.004 * np.eye(7)
    for _ in range(10):
        q, v, a = rs.uniform(-1.5, 1.5, 7), rs.uniform(-3, 3, 7), rs.uniform(-5, 5, 7)
        M = ref_arm.mass_matrix(q)
        assert np.abs(M - M.T).max() < 1e-14
        assert np.linalg.eigvalsh(M).min() > 0
        lhs = ref_arm.rne(q, v, a) - ref_arm.rne(q, v)
        np.testing.assert_allclose(lhs, (M - arm_diag) @ a, rtol=1e-11, atol=1e-12)


def test_bias_is_lagrangian_derivative(ref_arm):
    """c(q,v) = Mdot v - 1/2 d(v^T M v)/dq, by central differences of the Jacobian-built M."""
    rs = np.random.RandomState(1)
    eps = 1e-6
    E = np.eye(7)
    for _ in range(3):
        q, v = rs.uniform(-1, 1, 7), rs.uniform(-2, 2, 7)
        dM = [(ref_arm.mass_matrix(q + eps * E[i]) - ref_arm.mass_matrix(q - eps * E[i])) / (2 * eps) for i in range(7)]
        mdot = sum(dM[i] * v[i] for i in range(7))
        dT = np.array([0.5 * v @ dM[i] @ v for i in range(7)])
        np.testing.assert_allclose(ref_arm.rne(q, v), mdot @ v - dT, atol=5e-8)


def test_site_jacobian_and_pose(ref_arm):
    np.testing.assert_allclose(ref_arm.site(np.zeros(7)), [0.821, -0.6, 0.0], atol=1e-15)
    # rotating the pan joint by 90 deg swings the straight arm from +x to +y about (0,-0.6,0)
    q = np.zeros(7)
    q[0] = np.pi / 2
    np.testing.assert_allclose(ref_arm.site(q), [0.0, -0.6 + 0.821, 0.0], atol=1e-15)


def _free_arm(raw_arm, damping=0.0, limited=False, plane=True):
    raw = copy.deepcopy(raw_arm)
    for b in raw.bodies:
        if b.joint is not None:
            b.joint.damping = damping
            b.joint.limited = limited
    if not plane:
        raw.plane = None
    return RefArm(raw.to_flat())


def test_energy_conserved_without_damping(raw_arm):
    arm = _free_arm(raw_arm, plane=False)
    rs = np.random.RandomState(2)
    q, v = rs.uniform(-0.5, 0.5, 7), rs.uniform(-1, 1, 7)
    e0 = arm.kinetic(q, v)
    es = []
    for _ in range(200):
        q, v, _, _ = arm.step(q, v, np.zeros(7))
        es.append(arm.kinetic(q, v))
    # semi-implicit Euler at h = 0.01: energy error stays at the O(h) integrator level, no drift
    assert abs(es[-1] - e0) / e0 < 2e-2
    assert max(abs(e - e0) for e in es) / e0 < 5e-2


def test_damping_dissipates(raw_arm, ref_arm):
    rs = np.random.RandomState(3)
    q, v = np.array([0.2, 0.3, 0.1, -0.5, 0.2, -0.4, 0.1]), rs.uniform(-1, 1, 7)
    e = ref_arm.kinetic(q, v)
    for _ in range(50):
        q, v, _, _ = ref_arm.step(q, v, np.zeros(7))
        e2 = ref_arm.kinetic(q, v)
        assert e2 < e
        e = e2


def test_limit_rows_strict_inequality_and_push_back(ref_arm):
    # exactly ON the limit (elbow, wrist-flex at qpos0): dist = 0 is NOT < 0 -> no row
    _, _, _, diag = ref_arm.step(np.zeros(7), np.zeros(7), np.zeros(7))
    assert diag[0] == 0
    # beyond the elbow's upper limit (0): one row, force pushes q back (qacc < 0)
    q = np.zeros(7)
    q[3] = 0.01
    _, v1, _, diag = ref_arm.step(q, np.zeros(7), np.zeros(7))
    assert diag[0] == 1
    assert diag[1 + 3] < 0 and v1[3] < 0
    # below the lower limit of the pan joint
    q = np.zeros(7)
    q[0] = -2.2854 - 0.02
    _, v1, _, diag = ref_arm.step(q, np.zeros(7), np.zeros(7))
    assert diag[0] == 1 and v1[0] > 0


def test_ctrl_is_clamped(ref_arm):
    q = np.array([0.1, 0.2, 0.1, -0.6, 0.1, -0.5, 0.1])
    a = ref_arm.step(q, np.zeros(7), np.full(7, 1.0))
    b = ref_arm.step(q, np.zeros(7), np.full(7, 7.5))
    np.testing.assert_array_equal(a[1], b[1])


def test_table_contact_pushes_up(ref_arm):
    # fold the arm down so the wrist ball (r = 0.08) dips below the table plane z = -0.425
    q = np.array([0.0, 0.75, 0.0, -0.3, 0.0, -0.2, 0.0])
    z = ref_arm.site(q)[2]
    assert z - 0.08 < -0.425 + 0.05
    # find a pose in shallow penetration by scanning the lift joint
    for lift in np.linspace(0.3, 1.39, 400):
        q[1] = lift
        if ref_arm.site(q)[2] - 0.08 + 0.03 * 0 < -0.425:
            break
    qn, vn, _, diag = ref_arm.step(q, np.zeros(7), np.zeros(7))
    assert diag[0] >= 1
    assert ref_arm.site(qn)[2] > ref_arm.site(q)[2] - 1e-12       # pushed up, not down


def test_site_lags_one_substep(ref_arm):
    q0 = np.array([0.1, 0.2, 0.1, -0.6, 0.1, -0.5, 0.1])
    u = np.full(7, 0.7)
    q1, v1, s1, _ = ref_arm.step(q0, np.zeros(7), u)
    q2, v2, s2, _ = ref_arm.step(q1, v1, u)
    np.testing.assert_array_equal(s1, ref_arm.site(q0))
    np.testing.assert_array_equal(s2, ref_arm.site(q1))
    tgt = np.array([0.1, 0.1, 0.1])
    qe, ve, r, obs = ref_arm.env_step(q0, np.zeros(7), u, tgt)
    np.testing.assert_array_equal(qe, q2)
    d = ref_arm.site(q1) - tgt
    assert r == -(np.abs(d).sum() + 5 * np.sqrt(d @ d))
    np.testing.assert_array_equal(obs, np.concatenate([q2, v2, ref_arm.site(q1), d]))


def test_rollout_layout(ref_arm):
    rs = np.random.RandomState(5)
    P, H = 6, 5
    mean, noise = rs.randn(H, 7) * 0.2, rs.randn(P, H, 7)
    tgt = np.array([0.2, -0.1, 0.15])
    obs, rew, act, done, nobs = ref_arm.rollout(np.zeros(7), np.zeros(7), tgt, mean, noise)
    assert np.array_equal(act, mean[None] + noise)                 # unclipped actions
    assert not done.any()
    np.testing.assert_array_equal(obs[:, 1:], nobs[:, :-1])        # obs[t] = next_obs[t-1]
    h0 = ref_arm.site(np.zeros(7))
    np.testing.assert_array_equal(obs[0, 0], np.concatenate([np.zeros(14), h0, h0 - tgt]))
    # particle 3 re-simulated step by step
    q, v = np.zeros(7), np.zeros(7)
    for t in range(H):
        q, v, r, o = ref_arm.env_step(q, v, act[3, t], tgt)
        assert r == rew[3, t]
        np.testing.assert_array_equal(o, nobs[3, t])
    assert ref_arm.newton_stats()["fails"] == 0


def _limit_rows(raw, ref, q, v):
    """Joint-limit rows of one substep, restated in numpy (impedance, reference acceleration): (J, D, aref) each."""
    joints = [b.joint for b in raw.bodies if b.joint is not None]
    dofw, _ = ref.invweight0()
    h = raw.timestep
    dmin, dmax, width, mid, power = raw.solimp
    tc, dr = max(raw.solref[0], 2 * h), raw.solref[1]
    rows = []
    for j, jt in enumerate(joints):
        for side, bound in ((-1, jt.range[0]), (1, jt.range[1])):
            dist = side * (bound - q[j])
            if jt.limited and dist < 0:
                x = abs(dist) / width
                y = 1.0 if x >= 1 else (x ** power / mid ** (power - 1) if x <= mid else 1 - (1 - x) ** power / (1 - mid) ** (power - 1))
                imp = dmin + y * (dmax - dmin)
                J = np.zeros(len(joints))
                J[j] = -side
                rows.append((J, 1.0 / max((1 - imp) / imp * dofw[j], 1e-15),
                             -2 / (dmax * tc) * (-side * v[j]) - imp * dist / (dmax * dmax * tc * tc * dr * dr)))
    return rows


def test_constraint_solver_matches_kkt_enumeration(raw_arm, ref_arm):
    """The oracle's Newton + line search against brute force: every subset of the limit rows is tried as the active
    set, the one consistent with its own solution is the unique minimiser.  States with two or three joints past
    their limits, incl. the one on which a zero polishing step once produced 0/0 in the line search."""
    import itertools
    joints = [b.joint for b in raw_arm.bodies if b.joint is not None]
    lo, hi = np.array([j.range[0] for j in joints]), np.array([j.range[1] for j in joints])
    damp, gear = np.array([j.damping for j in joints]), np.array([a.gear for a in raw_arm.actuators])
    cases = [(np.array([0.63395138, 0.94957624, 1.49866263, -2.32676372, -1.53308188, -1.11386727, -0.16821426]),
              np.array([-0.70879538, 1.14306517, -1.62665656, -0.02570151, 0.64803507, 0.28575956, -9.98607415]),
              np.array([-1.43339186, -1.61867497, -0.28029994, 1.9364087, -0.50533047, -1.14888822, 0.91139836]))]
    rs = np.random.RandomState(11)
    for _ in range(12):
        q = lo + (hi - lo) * rs.rand(7)
        for j in rs.choice(7, 3, replace=False):
            q[j] = (hi[j] + 0.02 * rs.rand()) if rs.rand() < 0.5 else (lo[j] - 0.02 * rs.rand())
        q[1] = min(q[1], 0.3)                       # hand above the table: limit rows only
        cases.append((q, 3.0 * rs.randn(7), 2.0 * rs.randn(7)))
    checked = 0
    for q, v, u in cases:
        for _ in range(2):                          # two substeps each
            M, bias = ref_arm.mass_matrix(q), ref_arm.rne(q, v)
            fs = -bias - damp * v + gear * np.clip(u, -1, 1)
            rows = _limit_rows(raw_arm, ref_arm, q, v)
            q2, v2, _, diag = ref_arm.step(q, v, u)
            if int(diag[0]) == len(rows) and rows:  # (no contact row in this substep)
                sols = []
                for mask in itertools.product([0, 1], repeat=len(rows)):
                    H, rhs = M.copy(), fs.copy()
                    for m, (J, D, ar) in zip(mask, rows):
                        if m:
                            H += D * np.outer(J, J)
                            rhs += D * ar * J
                    a = np.linalg.solve(H, rhs)
                    if all(((J @ a - ar) < 0) == bool(m) for m, (J, D, ar) in zip(mask, rows)):
                        sols.append(a)
                assert len(sols) == 1
                np.testing.assert_allclose(diag[1:8], sols[0], rtol=1e-9, atol=1e-9)
                checked += 1
            q, v = q2, v2
    assert checked >= 12 and ref_arm.newton_stats()["fails"] == 0

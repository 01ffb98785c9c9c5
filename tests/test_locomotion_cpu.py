"""CPU: the locomotion models (reference-vendored swimmer.xml / half_cheetah.xml) - the MJCF loader against the restated
constants, the two model compilers against each other, and MuJoCo-free physical properties of the oracle's new physics
(slide joints, floating roots, the inertia-box fluid model, pyramidal friction cones).  MuJoCo itself cannot run here
(parity unpinned, see the oracle header), so the oracle is held to identities and to textbook mechanics."""
import os

import numpy as np
import pytest

from mjmpc_amd.models.compile_tree import compile_tree
from mjmpc_amd.models.half_cheetah import half_cheetah_raw
from mjmpc_amd.models.mjcf import load_mjcf
from mjmpc_amd.models.raw import (GEOM_SPHERE, JOINT_SLIDE, TASK_FORWARD, RawActuator, RawBody, RawGeom, RawJoint, RawModel,
                                  RawPlane)
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from mjmpc_amd.models.swimmer import swimmer_raw

XML = "/root/reference/mjmpc/envs/assets/xml/"


@pytest.mark.skipif(not os.path.isdir(XML), reason="reference tree not present (GPU box)")
def test_loader_reproduces_the_restated_models():
    """load_mjcf on the reference's three vendored XMLs gives exactly the tables mjmpc_amd/models/*.py restate."""
    a = load_mjcf(XML + "swimmer.xml", task=TASK_FORWARD, frame_skip=4, ctrl_cost=1e-4, obs_skip=2).to_flat()
    np.testing.assert_allclose(a, swimmer_raw().to_flat(), rtol=0, atol=1e-15)
    a = load_mjcf(XML + "half_cheetah.xml", task=TASK_FORWARD, frame_skip=5, ctrl_cost=0.1, obs_skip=1).to_flat()
    np.testing.assert_allclose(a, half_cheetah_raw().to_flat(), rtol=0, atol=1e-15)
    np.testing.assert_allclose(load_mjcf(XML + "sawyer.xml").to_flat(), reacher7dof_raw().to_flat(), rtol=0, atol=1e-15)


@pytest.mark.parametrize("raw_fn,nv,nu,dobs,total", [(swimmer_raw, 7, 4, 12, None), (half_cheetah_raw, 9, 6, 17, 14.0)])
def test_two_compilers_agree(raw_fn, nv, nu, dobs, total):
    from oracle.physics_ref import RefArm
    raw = raw_fn()
    m, ref = compile_tree(raw), RefArm(raw.to_flat())
    assert (m.nv, m.nu, m.d_obs, ref.nv, ref.d_obs) == (nv, nu, dobs, nv, dobs)
    mass, ipos, inertia = ref.inertial()
    np.testing.assert_allclose(m.body_mass, mass[1:], rtol=1e-12)
    if total is not None:
        assert abs(mass.sum() - total) < 1e-12                  # settotalmass
    dof_iw, body_iw = ref.invweight0()
    np.testing.assert_allclose(m.dof_invweight0, dof_iw, rtol=1e-9)
    np.testing.assert_allclose(m.body_invweight0, body_iw[1:], rtol=1e-9, atol=1e-15)
    assert abs(ref.mass_matrix(np.zeros(nv))[0, 0] - mass.sum()) < 1e-12    # the root slide carries the whole mass
    assert dof_iw[0] >= 1.0 / mass.sum() - 1e-12
    act = m.field("act")[:nv].astype(int)
    assert sorted(a for a in act if a >= 0) == list(range(nu)) and (act[:3] == -1).all()     # the root is not actuated


def test_model_compilers_reproduce_mujoco_20_body_masses():
    """The one piece of the physics half that IS pinned on MuJoCo's own output: `model.body_mass` of HalfCheetah as
    mujoco-py 2.0 users print it (settotalmass 14 over capsules whose end caps MuJoCo 2.0 counts as pi r^3 - with
    4/3 pi r^3 the torso would weigh 6.2502), and the same rule on Hopper's four capsules (gym's 3.53429174 ...;
    MuJoCo >= 2.1.2 and gymnasium's -v4 give 3.6651914 ...).  Both model compilers, nine digits."""
    from oracle.physics_ref import RefArm
    published = [6.36031332, 1.53524804, 1.58093995, 1.0691906, 1.42558747, 1.17885117, 0.84986945]
    raw = half_cheetah_raw()
    np.testing.assert_allclose(compile_tree(raw).body_mass[2:], published, rtol=0, atol=5e-9)
    np.testing.assert_allclose(RefArm(raw.to_flat()).inertial()[0][3:], published, rtol=0, atol=5e-9)
    from mjmpc_amd.models.compile import _geom_inertial
    from mjmpc_amd.models.raw import GEOM_CAPSULE, MJ20_CAPSULE_CAP
    hopper = [(0.05, 0.4), (0.05, 0.45), (0.04, 0.5), (0.06, 0.39)]         # radius, length of gym's hopper.xml capsules
    m20 = [_geom_inertial(RawGeom(GEOM_CAPSULE, r, (0, 0, 0), (0, 0, h)), MJ20_CAPSULE_CAP)[0] for r, h in hopper]
    m21 = [_geom_inertial(RawGeom(GEOM_CAPSULE, r, (0, 0, 0), (0, 0, h)), 4.0 / 3.0)[0] for r, h in hopper]
    np.testing.assert_allclose(m20, [3.53429174, 3.92699082, 2.71433605, 5.0893801], rtol=0, atol=5e-9)
    np.testing.assert_allclose(m21, [3.6651914, 4.0578905, 2.7813567, 5.3155748], rtol=0, atol=5e-8)


@pytest.mark.parametrize("raw_fn", [swimmer_raw, half_cheetah_raw])
def test_oracle_identities_with_slide_joints(raw_fn):
    """M symmetric positive definite and RNE(q, v, a) - RNE(q, v, 0) = (M - armature) a: the Jacobian-built mass matrix
    and the Newton-Euler pass agree on trees with slide joints."""
    from oracle.physics_ref import RefArm
    raw = raw_fn()
    ref = RefArm(raw.to_flat())
    arm = np.array([b.joint.armature for b in raw.bodies if b.joint is not None])
    rs = np.random.RandomState(0)
    for _ in range(3):
        q, v, a = 0.5 * rs.standard_normal(ref.nv), rs.standard_normal(ref.nv), rs.standard_normal(ref.nv)
        M = ref.mass_matrix(q)
        np.testing.assert_allclose(M, M.T, atol=1e-13)
        assert np.linalg.eigvalsh(M).min() > 0
        np.testing.assert_allclose(ref.rne(q, v, a) - ref.rne(q, v), (M - np.diag(arm)) @ a, rtol=1e-9, atol=1e-11)


def _in_vacuum(raw):
    raw.density = raw.viscosity = 0.0
    return raw


def test_floating_root_conserves_momentum_without_a_medium():
    """A swimmer in vacuum driven by its own motors: the generalized momentum of the root slides (= total linear
    momentum) is conserved by the equations of motion - internal torques cannot push the centre of mass.  The Euler
    integrator (MuJoCo's too) keeps it only to first order in the time step, so the drift must shrink with it."""
    from oracle.physics_ref import RefArm

    def drift(h):
        raw = _in_vacuum(swimmer_raw())
        raw.timestep = h
        ref = RefArm(raw.to_flat())
        q, v = np.zeros(7), np.zeros(7)
        v[:2] = [0.3, -0.2]
        p0 = (ref.mass_matrix(q) @ v)[:2]
        for t in range(int(round(0.5 / h))):
            q, v, _, _ = ref.step(q, v, 0.05 * np.sin(10.0 * t * h + np.arange(4)))
        assert np.abs(q[3:]).max() > 0.05                       # the joints did move
        return np.abs((ref.mass_matrix(q) @ v)[:2] - p0).max() / np.abs(p0).max()

    d1, d2 = drift(0.005), drift(0.00125)
    assert d1 < 0.05 and d2 < 0.35 * d1


def test_medium_drains_kinetic_energy_and_propels_an_undulating_body():
    from oracle.physics_ref import RefArm
    ref = RefArm(swimmer_raw().to_flat())
    rs = np.random.RandomState(1)
    q, v = 0.2 * rs.standard_normal(7), rs.standard_normal(7)
    e = [ref.kinetic(q, v)]
    for _ in range(100):
        q, v, _, _ = ref.step(q, v, np.zeros(4))
        e.append(ref.kinetic(q, v))
    assert all(b <= a * (1 + 1e-9) for a, b in zip(e, e[1:])) and e[-1] < 0.5 * e[0]
    # a travelling wave along the body moves the swimmer
    q, v = np.zeros(7), np.zeros(7)
    for t in range(1600):
        q, v, _, _ = ref.step(q, v, np.sin(0.0125 * t + 1.2 * np.arange(4)))
    assert np.hypot(q[0], q[1]) > 0.5


def _block(mu, slope):
    """A 1 kg sphere on slides x / z over a plane with friction; gravity tilted by `slope` instead of the plane."""
    g = 9.81
    density = 1.0 / (4.0 / 3.0 * np.pi * 0.1 ** 3)
    bodies = [RawBody("x", -1, (0.0, 0.0, 0.1), joint=RawJoint((1, 0, 0), (0, 0), limited=False, name="x", type=JOINT_SLIDE)),
              RawBody("z", 0, (0.0, 0.0, 0.0), joint=RawJoint((0, 0, 1), (0, 0), limited=False, name="z", type=JOINT_SLIDE),
                      geoms=[RawGeom(GEOM_SPHERE, 0.1, (0, 0, 0), density=density, collide=True, friction=mu, condim=3)])]
    return RawModel(bodies=bodies, actuators=[RawActuator("x", 1.0, (-1, 1))], site_body=1, site_pos=(0, 0, 0),
                    target_pos=(0, 0, 0), plane=RawPlane((0, 0, 0), (0, 0, 1), 0.0, friction=mu, condim=3), timestep=0.002,
                    frame_skip=1, gravity=(g * np.sin(slope), 0.0, -g * np.cos(slope)), task=TASK_FORWARD, obs_skip=0)


def test_pyramidal_friction_is_coulomb_friction_along_the_pyramid_axes():
    """Textbook checks of the four-row friction pyramid: a sliding block decelerates at mu g, a block on a slope
    below the friction angle stays put (up to the soft constraint's creep), above it it accelerates at
    g (sin - mu cos)."""
    from oracle.physics_ref import RefArm
    g, mu, h = 9.81, 0.4, 0.002
    ref = RefArm(_block(mu, 0.0).to_flat())
    q, v = np.zeros(2), np.array([8.0, 0.0])
    for _ in range(500):            # (at this speed the cone's edges carry normal force too and the block hops; on average ...)
        q, v, _, _ = ref.step(q, v, np.zeros(1))
    assert abs((8.0 - v[0]) / (500 * h) - mu * g) < 0.05 * mu * g
    for _ in range(1500):
        q, v, _, _ = ref.step(q, v, np.zeros(1))
    assert abs(v[0]) < 1e-3 and abs(q[1]) < 1e-3                # it stopped, resting on the plane - and stays stopped
    for slope, moves in ((0.9 * np.arctan(mu), False), (1.5 * np.arctan(mu), True)):
        ref = RefArm(_block(mu, slope).to_flat())
        q, v = np.zeros(2), np.zeros(2)
        for _ in range(500):
            q, v, _, _ = ref.step(q, v, np.zeros(1))
        v0 = v[0]
        for _ in range(250):
            q, v, _, _ = ref.step(q, v, np.zeros(1))
        acc = (v[0] - v0) / (250 * h)
        if moves:
            assert abs(acc - g * (np.sin(slope) - mu * np.cos(slope))) < 0.03 * g * np.sin(slope)
        else:
            assert abs(acc) < 1e-2 and abs(v[0]) < 0.02
    assert ref.newton_stats()["fails"] == 0


def test_cheetah_settles_on_its_feet_and_the_tree_compiler_reads_its_contacts():
    from oracle.physics_ref import RefArm
    raw = half_cheetah_raw()
    m = compile_tree(raw)
    assert m.field("n_sphere")[0] == 16 and m.field("any_friction")[0] == 1 and m.max_path == 6
    sph = m.field("spheres").reshape(16, 24)
    assert np.allclose(sph[:, 7], 0.4) and np.allclose(sph[:, 4], 0.046) and np.allclose(np.linalg.norm(sph[:, 8:11], axis=1), 1)
    assert list(m.parent) == [-1, 0, 1, 2, 3, 4, 2, 6, 7]
    ref = RefArm(raw.to_flat())
    q, v = np.zeros(9), np.zeros(9)
    for _ in range(200):
        q, v, r, obs = ref.env_step(q, v, np.zeros(6), np.zeros(3))
    assert np.abs(v).max() < 1e-6 and -0.2 < q[1] < -0.05 and abs(q[2]) < 0.2 and ref.newton_stats()["fails"] == 0
    assert obs.shape == (17,) and np.allclose(obs[:8], q[1:]) and abs(r) < 1e-6


def _swimmer_segments(theta):
    """End points of the five 0.3-long segments for joint angles theta[4] (root at the origin, heading +x)."""
    pts, ang, segs = [np.zeros(2)], 0.0, []
    for k in range(5):
        ang += theta[k - 1] if k else 0.0
        e = pts[-1] + 0.3 * np.array([np.cos(ang), np.sin(ang)])
        segs.append((pts[-1], e))
        pts.append(e)
    return segs


def _gap(sa, sb, ra, rb):
    t = np.linspace(0.0, 1.0, 31)
    A = sa[0][None] + t[:, None] * (sa[1] - sa[0])[None]
    B = sb[0][None] + t[:, None] * (sb[1] - sb[0])[None]
    return np.sqrt(((A[:, None] - B[None]) ** 2).sum(-1)).min() - ra - rb


def test_swimmer_self_contact_reachability():
    """swimmer.xml's segments collide with each other (default contype / conaffinity; parent and child excluded).  Within
    the joint ranges of +-1.5 rad: segments TWO apart never touch; segments three or four apart do once the chain curls
    into a loop - every joint between them bent the same way beyond ~1.15 rad."""
    import itertools
    radii = (0.07, 0.065, 0.06, 0.055, 0.05)
    best = {}
    for lim in (1.1, 1.5):
        grid = np.linspace(-lim, lim, 7)
        for th in itertools.product(grid, repeat=4):
            segs = _swimmer_segments(th)
            for i in range(5):
                for j in range(i + 2, 5):
                    g = _gap(segs[i], segs[j], radii[i], radii[j])
                    best[(lim, j - i)] = min(best.get((lim, j - i), 9.0), g)
    assert best[(1.5, 2)] > 0.15                            # unreachable at any admissible posture
    assert best[(1.1, 3)] > 0.05 and best[(1.1, 4)] > 0.05  # ... and so is everything below ~1.1 rad of bend
    assert best[(1.5, 3)] < -0.04 and best[(1.5, 4)] < -0.04    # a curled chain does touch itself
    raw = swimmer_raw()
    assert sorted(raw.pairs) == sorted(("seg%d" % b, "seg%d" % a) for b in range(5) for a in range(b - 1))


def test_curled_swimmer_pushes_itself_apart():
    """Oracle: the chain curled onto itself (tail inside the torso's capsule).  With the pairs the contact force opens the
    loop; without them nothing does."""
    from oracle.physics_ref import RefArm
    q = np.zeros(7)
    q[3:] = -1.45
    radii = (0.07, 0.065, 0.06, 0.055, 0.05)

    def run(raw, steps):
        ref, qq, vv = RefArm(_in_vacuum(raw).to_flat()), q.copy(), np.zeros(7)
        for _ in range(steps):
            qq, vv = ref.step(qq, vv, np.zeros(4))[:2]
        return qq, vv

    g0 = _gap(*[_swimmer_segments(q[3:])[k] for k in (0, 3)], radii[0], radii[3])
    assert g0 < -0.01
    q1, v1 = run(swimmer_raw(), 40)
    q2, v2 = run(swimmer_raw(self_collision=False), 40)
    g1 = _gap(*[_swimmer_segments(q1[3:])[k] for k in (0, 3)], radii[0], radii[3])
    g2 = _gap(*[_swimmer_segments(q2[3:])[k] for k in (0, 3)], radii[0], radii[3])
    assert np.all(np.isfinite(q1)) and g1 > g0 + 0.01 and g1 > -0.005, (g0, g1)
    assert abs(g2 - g0) < 1e-9 and np.allclose(v2, 0.0, atol=1e-12)       # at rest in vacuum nothing moves it

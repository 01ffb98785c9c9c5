"""CPU: model description, MJCF-subset loader and the host compiler."""
import os
import textwrap

import numpy as np
import pytest

from mjmpc_amd.models.compile import ARM_BLOB_LEN, compile_arm
from mjmpc_amd.models.mjcf import load_mjcf
from mjmpc_amd.models.reacher7dof import reacher7dof_raw

SAWYER = "/root/reference/mjmpc/envs/assets/xml/sawyer.xml"


def test_blob_layout_and_kinematic_constants():
    m = compile_arm(reacher7dof_raw())
    assert m.blob.shape == (ARM_BLOB_LEN,) and ARM_BLOB_LEN == 255
    assert (m.nv, m.nu, m.d_obs, m.frame_skip) == (7, 7, 20, 2)
    off = m.field("off").reshape(3, 8).T
    np.testing.assert_allclose(off[:7].sum(0), [0.821, -0.6, 0.0], atol=1e-15)     # hand at qpos0
    assert m.field("armature")[7] == 1.0 and m.field("mass")[7] == 0.0             # spare lane
    # capsule volume as MuJoCo 2.0 computes it (end caps pi r^3, models/raw.py::MJ20_CAPSULE_CAP); with the 4/3 of later
    # versions the arm weighs SURVEY appendix A's 47.110 kg
    np.testing.assert_allclose(m.field("mass")[:7].sum(), 44.414266, rtol=1e-6)
    import dataclasses
    m43 = compile_arm(dataclasses.replace(reacher7dof_raw(), capsule_cap_factor=4.0 / 3.0))
    np.testing.assert_allclose(m43.field("mass")[:7].sum(), 47.10975, rtol=1e-6)
    assert m.field("sph_margin")[0] == 0.002 and m.field("n_sphere")[0] == 1


@pytest.mark.skipif(not os.path.exists(SAWYER), reason="reference tree not mounted")
def test_loader_reproduces_builtin_table_from_the_reference_xml():
    a = load_mjcf(SAWYER).to_flat()
    b = reacher7dof_raw().to_flat()
    np.testing.assert_array_equal(a, b)


def test_loader_on_a_small_model_and_rejections(tmp_path):
    xml = textwrap.dedent("""
    <mujoco>
      <compiler inertiafromgeom="true" angle="radian" coordinate="local"/>
      <option timestep="0.005" gravity="0 0 -9.81" integrator="Euler"/>
      <default><joint armature="0.01" damping="0.5" limited="true"/><geom margin="0.001" contype="0" conaffinity="0"/></default>
      <worldbody>
        <site name="target" pos="0.3 0 0.2"/>
        <body name="a" pos="0 0 0.1">
          <geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.03"/>
          <joint name="j0" axis="0 0 1" range="-1 1"/>
          <body name="b" pos="0.2 0 0">
            <geom type="sphere" pos="0.1 0 0" size="0.04"/>
            <joint name="j1" axis="0 1 0" range="-2 2" damping="0.1"/>
            <site name="finger" pos="0.1 0 0"/>
          </body>
        </body>
      </worldbody>
      <actuator>
        <motor joint="j0" gear="5" ctrlrange="-1 1" ctrllimited="true"/>
        <motor joint="j1" gear="2" ctrlrange="-1 1" ctrllimited="true"/>
      </actuator>
    </mujoco>""")
    p = tmp_path / "two_link.xml"
    p.write_text(xml)
    raw = load_mjcf(str(p))
    m = compile_arm(raw)
    assert (m.nv, m.nu, m.timestep) == (2, 2, 0.005)
    np.testing.assert_array_equal(m.field("gravity"), [0, 0, -9.81])
    np.testing.assert_array_equal(m.field("damping")[:2], [0.5, 0.1])
    np.testing.assert_allclose(m.field("site_pos"), [0.1, 0, 0])
    assert m.field("n_sphere")[0] == 0
    bad = xml.replace('type="sphere"', 'type="box"')
    (tmp_path / "bad.xml").write_text(bad)
    with pytest.raises(ValueError):
        load_mjcf(str(tmp_path / "bad.xml"))
    ball = xml.replace('<joint name="j1"', '<joint name="j1" type="ball"')
    (tmp_path / "ball.xml").write_text(ball)        # (a limited ball joint loads since round 4; the arm kernel cannot take it)
    with pytest.raises(ValueError):
        compile_arm(load_mjcf(str(tmp_path / "ball.xml")))
    bad = xml.replace('<joint name="j1"', '<joint name="j1" ref="0.1"')
    (tmp_path / "bad2.xml").write_text(bad)
    raw_ref = load_mjcf(str(tmp_path / "bad2.xml"))        # (joint ref loads since round 5: the tree engine runs it, the arm kernel says so)
    assert raw_ref.bodies[1].joint.ref == 0.1 and raw_ref.qpos0[1] == 0.1
    with pytest.raises(ValueError, match="tree engine"):
        compile_arm(raw_ref)
    # a slide joint loads; since round 6 the arm kernels' extended-joint build takes it too (jtype in the blob)
    slide = xml.replace('<joint name="j1"', '<joint name="j1" type="slide"')
    (tmp_path / "slide.xml").write_text(slide)
    raw2 = load_mjcf(str(tmp_path / "slide.xml"))
    assert [b.joint.type for b in raw2.bodies] == [1, 2]
    assert list(compile_arm(raw2).field("jtype")[:2]) == [0.0, 1.0]


def test_systematic_resampling_matches_the_serial_walk(golden):
    """Host logic of PFMPC: the vectorised cumulative-sum search selects exactly the particles the
    reference's serial pointer walk selects (oracle restatement pinned on the golden vectors)."""
    import random
    from mjmpc_amd.control.particle_filter_controller import systematic_resample_indices
    from oracle import controllers_ref as cr
    g = golden("updates")
    for i in range(int(g["pf_n"])):
        t = "pf%d" % i
        lam, gamma, _, _ = g[t + "_cfg"]
        s0 = g[t + "_samples0"]
        w = cr.pf_weights(g[t + "_costs"], cr.gamma_seq(gamma, s0.shape[1]), lam)
        random.seed(123)
        r = random.uniform(0.0, 1.0 / s0.shape[0] * 1.0)
        idx = systematic_resample_indices(w, r)
        assert np.array_equal(s0[idx], g[t + "_samples1"])
    # degenerate pointers: at 0 the reference indexes [-1]; past the total it sticks to the last particle
    w = np.array([0.5, 0.25, 0.25])
    assert list(systematic_resample_indices(w, 0.0)) == [-1, 0, 1]
    assert list(systematic_resample_indices(w * 0.5, 0.3)) == [1, 2, 2]


def test_impedance_power_must_be_a_small_integer():
    """The kernel evaluates solimp's power by repeated multiplication: the model compiler rejects the rest."""
    import dataclasses
    import pytest
    from mjmpc_amd.models.compile import compile_arm
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    raw = reacher7dof_raw()
    for ok in (1.0, 2.0, 3.0, 6.0):
        compile_arm(dataclasses.replace(raw, solimp=(0.9, 0.95, 0.001, 0.5, ok)))
    for bad in (2.5, 0.5, 100.0):
        with pytest.raises(NotImplementedError):
            compile_arm(dataclasses.replace(raw, solimp=(0.9, 0.95, 0.001, 0.5, bad)))


def test_loader_position_servos_pairs_and_self_collision(tmp_path):
    """Round 3's MJCF additions on a small model: an object on a slide + hinge, a two-link manipulator with <position> servos,
    an explicit <contact><pair>, and MuJoCo's contype / conaffinity rule for body geoms against each other."""
    from mjmpc_amd.models.compile_tree import compile_tree
    from mjmpc_amd.models.raw import TASK_REACH
    xml = textwrap.dedent("""
    <mujoco>
      <compiler inertiafromgeom="true" angle="radian" coordinate="local"/>
      <option timestep="0.002" gravity="0 0 -9.81" integrator="Euler"/>
      <default><joint limited="true" damping="0.1"/><geom contype="0" conaffinity="0" condim="3" friction="0.8 0.005 0.0001"/></default>
      <worldbody>
        <site name="target" pos="0 0 0.3"/>
        <body name="obj" pos="0.1 0 0.2">
          <joint name="oz" type="slide" axis="0 0 1" range="-1 1"/>
          <joint name="ory" type="hinge" axis="0 1 0" range="-3 3"/>
          <geom name="pen" type="capsule" fromto="-0.05 0 0 0.05 0 0" size="0.01"/>
          <site name="finger" pos="0 0 0"/>
        </body>
        <body name="a" pos="0 0 0.1">
          <joint name="j0" axis="0 1 0" range="-1 1"/>
          <geom name="ga" type="capsule" fromto="0 0 0 0.2 0 0" size="0.02"/>
          <body name="b" pos="0.2 0 0">
            <joint name="j1" axis="0 1 0" range="-2 2"/>
            <geom name="gb" type="sphere" pos="0.05 0 0" size="0.03"/>
          </body>
        </body>
      </worldbody>
      <contact><pair geom1="ga" geom2="pen"/><pair geom1="gb" geom2="pen"/></contact>
      <actuator>
        <position joint="j0" kp="40" ctrlrange="-1 1" ctrllimited="true"/>
        <position joint="j1" kp="10" gear="2" ctrlrange="-2 2" ctrllimited="true"/>
      </actuator>
    </mujoco>""")
    p = tmp_path / "obj_arm.xml"
    p.write_text(xml)
    raw = load_mjcf(str(p), task=TASK_REACH)
    assert [(a.joint, a.gear, a.kp) for a in raw.actuators] == [("j0", 1.0, 40.0), ("j1", 2.0, 10.0)]
    assert raw.pairs == [("ga", "pen"), ("gb", "pen")]          # explicit pairs only: the masks are 0
    m = compile_tree(raw)
    assert m.nv == 4 and int(m.field("n_sphere")[0]) == 2 and int(m.field("any_friction")[0]) == 1
    # servo: effective gear gear * kp, stiffness gear^2 kp at the joint
    np.testing.assert_allclose(m.field("gear")[2:4], [40.0, 20.0])
    np.testing.assert_allclose(m.field("kpg")[2:4], [40.0, 40.0])
    # the manipulator's root hangs under the object's last link in the elimination tree
    assert list(m.field("eparent")[:4].astype(int)) == [-1, 0, 1, 2] and m.max_path == 4
    np.testing.assert_allclose(m.field("spheres").reshape(16, 24)[:2, 7], [0.8, 0.8])      # mu of both pairs
    # masks that match: MuJoCo's rule adds the body-geom pairs that are not parent and child (here: both manipulator geoms
    # against the object's; ga - gb are parent and child); later geom first
    auto = load_mjcf(_write(tmp_path, "auto.xml", xml.replace('contype="0" conaffinity="0"', 'contype="1" conaffinity="1"')
                                                      .replace('<contact><pair geom1="ga" geom2="pen"/><pair geom1="gb" geom2="pen"/></contact>', "")),
                     task=TASK_REACH)
    assert sorted(auto.pairs) == [("ga", "pen"), ("gb", "pen")]
    assert load_mjcf(str(tmp_path / "auto.xml"), task=TASK_REACH, self_collision=False).pairs == []
    # a pair across two branches of one tree (round 4): the elimination tree - the symbolic Cholesky of M's pattern plus
    # every row's clique - chains the two branches, so that the row's dofs lie on one path of it (rounds 1-3 refused it)
    bad = xml.replace('<body name="b" pos="0.2 0 0">', '<body name="c" pos="0 0.1 0"><joint name="j2" axis="0 1 0" range="-1 1"/>'
                      '<geom name="gc" type="sphere" pos="0.05 0 0" size="0.03"/></body><body name="b" pos="0.2 0 0">')
    bad = bad.replace('<pair geom1="gb" geom2="pen"/>', '<pair geom1="gb" geom2="gc"/>')
    mb = compile_tree(load_mjcf(_write(tmp_path, "branch_pair.xml", bad), task=TASK_REACH))
    ep = list(mb.field("eparent")[:mb.nv].astype(int))
    # links: 0, 1 the object (untouched by the remaining pairs ga-pen ... gb-gc), 2 = j0, 3 = j2 (body c), 4 = j1 (body b);
    # gb (link 4) against gc (link 3): link 4's row now reaches link 3, which becomes its parent in the elimination tree
    assert mb.nv == 5 and ep[4] == 3 and ep[3] == 2 and mb.max_path == 5
    assert list(mb.parent) == [-1, 0, -1, 2, 2]                     # (the kinematic tree is what it was)


def _write(tmp_path, name, text):
    (tmp_path / name).write_text(text)
    return str(tmp_path / name)

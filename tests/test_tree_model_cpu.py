"""CPU: the tree model compiler (mjmpc_amd/models/compile_tree.py) against the oracle's own compile of the same
flat description, topology tables of the kernel block, the MJCF loader on a branching model, and MuJoCo-free
properties of the oracle on a tree (what tests/test_oracle_physics.py checks on the serial arm)."""
import textwrap

import numpy as np
import pytest

from mjmpc_amd.models.compile_tree import TL, TREE_BLOB_LEN, compile_tree
from mjmpc_amd.models.hand24 import hand24_raw
from mjmpc_amd.models.mjcf import load_mjcf
from mjmpc_amd.models.raw import GEOM_CAPSULE, RawActuator, RawBody, RawGeom, RawJoint, RawModel


@pytest.fixture(scope="module")
def hand():
    from oracle.physics_ref import RefArm
    raw = hand24_raw()
    return raw, compile_tree(raw), RefArm(raw.to_flat())


def test_two_compilers_agree_on_the_hand(hand):
    raw, m, ref = hand
    assert m.blob.shape == (TREE_BLOB_LEN,) and TREE_BLOB_LEN == 3961
    assert (m.nv, m.nu, m.d_obs) == (24, 24, 54)
    mass, ipos, inertia = ref.inertial()
    np.testing.assert_allclose(m.body_mass, mass[1:], rtol=1e-12)
    dof_iw, body_iw = ref.invweight0()
    np.testing.assert_allclose(m.dof_invweight0, dof_iw, rtol=1e-9)
    np.testing.assert_allclose(m.body_invweight0, body_iw[1:], rtol=1e-9)
    assert m.field("n_sphere")[0] == 5


def test_topology_tables(hand):
    raw, m, ref = hand
    par = m.parent
    assert list(par[:5]) == [-1, 0, 1, 2, 3] and list(par[[4, 8, 12, 16, 20]]) == [3] * 5       # five fingers on the wrist link
    sub = m.field("subsize")[:24].astype(int)
    assert sub[0] == 24 and sub[3] == 21 and list(sub[4:8]) == [4, 3, 2, 1]
    anc = m.field("anc").reshape(5, TL).astype(int)
    assert list(anc[0, :24]) == list(par)                                  # distance 1 = parent
    assert anc[1, 7] == 5 and anc[2, 7] == 3 and anc[3, 7] == -1           # f0_dist: 7 -> 6 -> 5 -> 4 -> 3 -> 2 -> 1 -> 0
    mask = m.field("ancmask").reshape(2, TL).astype(np.int64)
    full = mask[0] | (mask[1] << 16)
    assert full[7] == sum(1 << k for k in (0, 1, 2, 3, 4, 5, 6, 7))
    assert full[23] == sum(1 << k for k in (0, 1, 2, 3, 20, 21, 22, 23))
    assert m.field("jumps")[0] == 3 and m.max_path == 8                     # longest path: 8 links
    # elimination lists of the tree-sparse L'DL: descendants sorted by height, packed k | distance << 8 | height << 16
    assert m.field("n_rounds")[0] == 8 and list(m.field("depth")[[0, 3, 4, 7]]) == [0, 3, 4, 7]
    el = m.field("elim").reshape(TL - 1, TL).astype(int)
    first = [(x & 255, (x >> 8) & 255, x >> 16) for x in el[:6, 3]]        # the wrist link: five fingertips first
    assert first == [(7, 4, 0), (11, 4, 0), (15, 4, 0), (19, 4, 0), (23, 4, 0), (6, 3, 1)]
    assert (el[:, 3] >= 0).sum() == 20 and (el[:, 7] >= 0).sum() == 0 and (el[:, 0] >= 0).sum() == 23
    assert (m.field("parent")[24:] == -1).all() and (m.field("subsize")[24:] == 0).all()


def test_links_must_be_depth_first():
    def body(name, parent):
        return RawBody(name, parent, (0.1, 0, 0), joint=RawJoint((0, 0, 1), (-1, 1), name=name),
                       geoms=[RawGeom(GEOM_CAPSULE, 0.02, (0, 0, 0), (0.1, 0, 0))])
    bodies = [body("a", -1), body("b", 0), body("c", 0), body("d", 1)]      # d (child of b) listed after c
    raw = RawModel(bodies=bodies, actuators=[RawActuator(b.name, 1.0, (-1, 1)) for b in bodies], site_body=3,
                   site_pos=(0, 0, 0), target_pos=(0, 0, 0), plane=None, timestep=0.01, frame_skip=1)
    with pytest.raises(ValueError, match="depth-first"):
        compile_tree(raw)
    bodies = [body("a", -1), body("b", 0), body("d", 1), body("c", 0)]
    raw.bodies = bodies
    raw.actuators = [RawActuator(b.name, 1.0, (-1, 1)) for b in bodies]
    raw.site_body = 2
    assert list(compile_tree(raw).parent) == [-1, 0, 1, 0]


def test_mjcf_loader_on_a_branching_model(tmp_path):
    xml = textwrap.dedent("""
    <mujoco>
      <compiler inertiafromgeom="true" angle="radian" coordinate="local"/>
      <option timestep="0.004" gravity="0 0 -9.81" integrator="Euler"/>
      <default><joint armature="0.01" damping="0.2" limited="true"/><geom margin="0.001" contype="0" conaffinity="0" condim="1"/></default>
      <worldbody>
        <geom type="plane" pos="0 0 -0.2" size="1 1 1" contype="1" conaffinity="1"/>
        <site name="target" pos="0.3 0 0.2"/>
        <body name="base" pos="0 0 0.1">
          <geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.03"/>
          <joint name="j0" axis="0 0 1" range="-1 1"/>
          <body name="left" pos="0.2 0.05 0">
            <geom type="capsule" fromto="0 0 0 0.1 0 0" size="0.02"/>
            <joint name="j1" axis="0 1 0" range="-2 2"/>
            <body name="left_tip" pos="0.1 0 0">
              <geom type="sphere" pos="0.05 0 0" size="0.02" contype="1" conaffinity="1"/>
              <joint name="j2" axis="0 1 0" range="-2 2"/>
              <site name="finger" pos="0.05 0 0"/>
            </body>
          </body>
          <body name="right" pos="0.2 -0.05 0">
            <geom type="capsule" fromto="0 0 0 0.1 0 0" size="0.02"/>
            <joint name="j3" axis="0 1 0" range="-2 2"/>
          </body>
        </body>
      </worldbody>
      <actuator>
        <motor joint="j0" gear="5" ctrlrange="-1 1" ctrllimited="true"/>
        <motor joint="j1" gear="2" ctrlrange="-1 1" ctrllimited="true"/>
        <motor joint="j2" gear="2" ctrlrange="-1 1" ctrllimited="true"/>
        <motor joint="j3" gear="2" ctrlrange="-1 1" ctrllimited="true"/>
      </actuator>
    </mujoco>""")
    p = tmp_path / "gripper.xml"
    p.write_text(xml)
    raw = load_mjcf(str(p))
    m = compile_tree(raw)
    assert list(m.parent) == [-1, 0, 1, 0] and m.nv == 4
    assert m.field("n_sphere")[0] == 1 and m.field("spheres")[0] == 2          # the sphere rides on link 2
    np.testing.assert_allclose(m.field("site_pos"), [0.05, 0, 0])
    from mjmpc_amd.models.compile import compile_arm
    with pytest.raises(ValueError, match="serial chain"):
        compile_arm(raw)                                                       # the 8-lane kernel stays a chain kernel


def test_oracle_properties_on_the_tree(hand):
    """M symmetric positive definite; RNE(q, v, a) - RNE(q, v, 0) = (M - armature) a  at random configurations of the
    branching model - the identities that tie the oracle's two independent dynamics routines together."""
    raw, m, ref = hand
    rs = np.random.RandomState(0)
    arm = np.array([b.joint.armature for b in raw.bodies if b.joint is not None])
    for _ in range(3):
        q = 0.5 * rs.standard_normal(24)
        v = rs.standard_normal(24)
        a = rs.standard_normal(24)
        M = ref.mass_matrix(q)
        np.testing.assert_allclose(M, M.T, atol=1e-14)
        assert np.linalg.eigvalsh(M).min() > 0
        np.testing.assert_allclose(ref.rne(q, v, a) - ref.rne(q, v), (M - np.diag(arm)) @ a, rtol=1e-9, atol=1e-12)
        # entries between links on different fingers vanish (tree sparsity)
        assert abs(M[5, 9]) < 1e-15 and abs(M[7, 23]) < 1e-15 and abs(M[3, 9]) > 0

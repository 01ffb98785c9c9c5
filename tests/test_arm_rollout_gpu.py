"""GPU parity: the HIP arm rollout (through the C ABI) against the FP64 C oracle on the same
seeded inputs.  Tolerances:
  f64 kernel vs oracle: costs rel <= 1e-9, observations abs <= 1e-9
  f32 kernel vs oracle: measured, stated below (abs <= 2e-3 on costs, a damped 7-dof arm over
  64 substeps with discontinuous limit activation)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _noise(P, H, A, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    eps = scale * rs.standard_normal((P, H, A))
    for t in range(2, H):
        eps[:, t] = 0.25 * eps[:, t] + 0.8 * eps[:, t - 1]
    return eps


@pytest.fixture(scope="module")
def engines(raw_arm):
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    return {dt: ArmRolloutEngine(raw_arm, dtype=dt) for dt in ("f64", "f32")}


STATES = [
    dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1])),
    dict(qp=np.array([0.3, 0.5, -0.2, -1.0, 0.4, -0.6, 0.2]), qv=np.array([0.5, -1.0, 0.3, 2.0, -0.5, 1.0, 0.1]),
         target_pos=np.array([-0.25, 0.15, 0.2])),
    # folded down onto the table: the wrist ball touches the plane during the rollout
    dict(qp=np.array([0.0, 0.7, 0.0, -0.2, 0.0, -0.1, 0.0]), qv=np.array([0.0, 1.5, 0.0, 0.0, 0.0, 0.0, 0.0]),
         target_pos=np.array([0.2, -0.1, -0.25])),
]


@pytest.mark.parametrize("si", range(len(STATES)))
def test_f64_matches_oracle(engines, ref_arm, si):
    eng = engines["f64"]
    st = STATES[si]
    P, H = 200, 32                     # P not a multiple of 8: ragged last wavefront
    rs = np.random.RandomState(10 + si)
    mean = 0.3 * rs.standard_normal((H, 7))
    noise = _noise(P, H, 7, 20 + si)
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o_obs, o_rew, o_act, o_done, o_nobs = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    assert np.array_equal(act, o_act)
    assert not done.any()
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0
    assert len(info) == 1 and "total_time" in info[0]


def test_contact_case_really_touches(ref_arm):
    st = STATES[2]
    mean = np.zeros((32, 7))
    obs, rew, act, done, nobs = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, None)
    assert nobs[0, :, 16].min() - 0.08 < -0.425 + 0.002      # hand z - ball radius below table + margin


def test_f32_within_stated_tolerance(engines, ref_arm):
    eng = engines["f32"]
    st = STATES[1]
    P, H = 512, 32
    mean = np.zeros((H, 7))
    noise = _noise(P, H, 7, 5).astype(np.float32).astype(np.float64)   # identical inputs to both sides
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o_obs, o_rew, o_act, o_done, o_nobs = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    err = np.abs(rew - o_rew)
    print("f32 cost error: max %.3e  mean %.3e" % (err.max(), err.mean()))
    assert err.max() < 2e-3
    assert np.abs(nobs - o_nobs).max() < 2e-2
    assert eng.solver_failures() == 0


def test_mean_only_and_tiny_shapes(engines, ref_arm):
    eng = engines["f64"]
    st = STATES[0]
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    mean = 0.5 * np.ones((3, 7))
    obs, rew, act, done, info, nobs = eng.rollout(1, 3, mean, None, "open_loop")
    o = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, None)
    np.testing.assert_allclose(rew, o[1], rtol=1e-9, atol=1e-9)
    assert np.array_equal(act[0], mean)
    with pytest.raises(ValueError):
        eng.rollout(1, 3, mean, None, "closed_loop")


def test_full_size_properties(engines):
    """BASELINE size (4096 x 32): size-independent properties instead of the (slow) oracle."""
    eng = engines["f64"]
    st = STATES[0]
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    P, H = 4096, 32
    noise = _noise(P, H, 7, 99)
    noise[P // 2:] = noise[:P // 2]                    # duplicated particles
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, 7)), noise, "open_loop")
    assert np.array_equal(rew[:P // 2], rew[P // 2:])   # determinism / particle independence
    assert np.array_equal(obs[:, 1:], nobs[:, :-1])     # obs[t] == next_obs[t-1]
    assert np.isfinite(rew).all() and (rew < 0).all()
    d = nobs[..., 17:20]
    np.testing.assert_allclose(-rew, np.abs(d).sum(-1) + 5 * np.sqrt((d ** 2).sum(-1)), rtol=1e-12)
    lo = np.array([-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5]) - 0.2
    hi = np.array([1.714602, 1.3963, 1.7, 0.0, 1.5, 0.0, 1.5]) + 0.2
    q = nobs[..., :7]
    assert (q > lo).all() and (q < hi).all()            # soft limits hold the joints near their range
    assert eng.solver_failures() == 0


def test_two_link_arm_with_gravity(tmp_path):
    """Generality of the kernel beyond sawyer.xml: nv = 2, gravity on, rotated body frame, no contact."""
    import textwrap
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.mjcf import load_mjcf
    from oracle.physics_ref import RefArm
    xml = textwrap.dedent("""
    <mujoco>
      <compiler inertiafromgeom="true" angle="radian" coordinate="local"/>
      <option timestep="0.005" gravity="0 0 -9.81" integrator="Euler"/>
      <default><joint armature="0.01" damping="0.5" limited="true"/><geom margin="0.001" contype="0" conaffinity="0"/></default>
      <worldbody>
        <site name="target" pos="0.3 0 0.2"/>
        <body name="a" pos="0 0 0.1" quat="0.9238795325112867 0 0.3826834323650898 0">
          <geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.03"/>
          <joint name="j0" axis="0 0 1" range="-1 1"/>
          <body name="b" pos="0.2 0 0">
            <geom type="sphere" pos="0.1 0 0" size="0.04"/>
            <geom type="capsule" fromto="0 0 0 0.1 0.05 0" size="0.02"/>
            <joint name="j1" axis="0 1 0" range="-0.4 2" damping="0.1"/>
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
    raw = load_mjcf(str(p), frame_skip=3)
    eng = ArmRolloutEngine(raw, dtype="f64")
    ref = RefArm(raw.to_flat())
    P, H = 40, 20
    rs = np.random.RandomState(3)
    mean, noise = 0.2 * rs.randn(H, 2), rs.randn(P, H, 2)
    st = dict(qp=np.array([0.2, -0.1]), qv=np.array([0.5, 0.0]), target_pos=np.array([0.3, 0.0, 0.2]))
    eng.set_env_state(st)
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    assert obs.shape == (P, H, 10)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    assert nobs[..., 1].min() < -0.4                      # the lower limit of j1 was hit under gravity
    assert eng.solver_failures() == 0


def test_closed_loop_linear_mode(engines, ref_arm):
    """mode='closed_loop_linear' (gym_env_wrapper.py:135-136): action = W^T [obs; 1] + noise."""
    eng = engines["f64"]
    st = STATES[1]
    P, H = 64, 24
    rs = np.random.RandomState(8)
    W = 0.3 * rs.standard_normal((21, 7))
    W[14:20] *= 2.0                                   # make the (lagged) hand position matter
    noise = 0.3 * _noise(P, H, 7, 31)
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, W, noise, "closed_loop_linear")
    o_obs, o_rew, o_act, _, o_nobs = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], W, noise,
                                                     mode="closed_loop_linear")
    np.testing.assert_allclose(act, o_act, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-9)
    # the first action is a function of the fresh observation only
    np.testing.assert_allclose(act[:, 0] - noise[:, 0], (W.T @ np.append(o_obs[0, 0], 1.0))[None].repeat(P, 0),
                               rtol=1e-12, atol=1e-12)
    # noise=None: every particle follows the deterministic closed loop
    obs1, rew1, act1, _, _, _ = eng.rollout(1, H, W, None, "closed_loop_linear")
    o1 = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], W, None, mode="closed_loop_linear", horizon=H)
    np.testing.assert_allclose(rew1, o1[1], rtol=1e-9, atol=1e-9)
    assert eng.solver_failures() == 0


def test_dynamics_randomization_per_shard(raw_arm):
    """SubprocVecEnv.randomize_dynamics semantics: every shard simulates its own perturbed model
    (gym_env_wrapper.py:367-416).  The kernel's per-shard model blocks + the host compiler's overrides
    are checked against the oracle edited with the same values through its own setters."""
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.compile import principal_inertia
    from oracle.physics_ref import RefArm
    S = 4
    eng = ArmRolloutEngine(raw_arm, dtype="f64", num_shards=S)
    cfg = {"body_mass": {"r_forearm_link": [0.3, 0.5], "r_shoulder_lift_link": [0.2, 0.0]},
           "body_inertia": {"r_upper_arm_link": [0.4, 1.0]},
           "dof_damping": {"r_elbow_flex_joint": [0.5, 2.0], "r_shoulder_pan_joint": [0.1, -0.5]},
           "geom_size": {"wrist_ball": [0.2, 0.0]},
           "geom_friction": {"wrist_ball": [0.1, 0.0]},
           # sawyer.xml sets no frictionloss (MuJoCo default 0) and the randomization is multiplicative: stays 0
           "dof_frictionloss": {"r_wrist_flex_joint": [0.3, 1.0]}}
    defaults, rand = eng.randomize_dynamics(cfg, base_seed=123)
    assert all(r["dof_frictionloss"]["r_wrist_flex_joint"] == 0.0 for r in rand)
    assert len(rand) == S and rand[0]["body_mass"]["r_forearm_link"] != rand[1]["body_mass"]["r_forearm_link"]
    m0 = defaults[0]["body_mass"]["r_forearm_link"]
    for r in rand:                                        # uniform in m (1 +- noise), m = (1 + bias) default
        assert 1.5 * m0 * 0.7 <= r["body_mass"]["r_forearm_link"] <= 1.5 * m0 * 1.3
    P, H = 8 * 3 * S, 20
    rs = np.random.RandomState(4)
    mean, noise = 0.2 * rs.randn(H, 7), _noise(P, H, 7, 77)
    st = STATES[2]                                        # on the table: the sphere radius matters
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    assert len(info) == S
    names = [b.name for b in raw_arm.bodies]
    joints = raw_arm.joint_names
    n = P // S
    base_costs = None
    for k in range(S):
        ref = RefArm(raw_arm.to_flat())
        _, _, inertia = ref.inertial()
        r = rand[k]
        for name, v in r["body_mass"].items():
            ref.set_body_mass(1 + names.index(name), v)
        for name, v in r["body_inertia"].items():
            b = 1 + names.index(name)
            _, V = principal_inertia(inertia[b])
            ref.set_body_inertia(b, V @ np.diag(v) @ V.T)
        for name, v in r["dof_damping"].items():
            ref.set_dof_damping(joints.index(name), v)
        ref.set_sphere_radius(0, r["geom_size"]["wrist_ball"][0])
        sl = slice(k * n, (k + 1) * n)
        o = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise[sl])
        np.testing.assert_allclose(rew[sl], o[1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(nobs[sl], o[4], rtol=0, atol=1e-9)
        if k == 0:
            base_costs = rew[sl]
    # different shards really see different dynamics (same noise slice fed to shard 0 and 1)
    noise2 = noise.copy()
    noise2[n:2 * n] = noise[:n]
    _, rew2, _, _, _, _ = eng.rollout(P, H, mean, noise2, "open_loop")
    assert np.abs(rew2[n:2 * n] - base_costs).max() > 1e-4
    with pytest.raises(Exception):
        eng.rollout(4 * S, H, mean, noise[:4 * S], "open_loop")      # shard not a multiple of 8 particles
    assert eng.solver_failures() == 0


def test_per_shard_start_states(raw_arm, ref_arm):
    """set_env_state with one state dict per shard (subproc_vec_env.py:242-251)."""
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    S = 3
    eng = ArmRolloutEngine(raw_arm, dtype="f64", num_shards=S)
    P, H = 8 * 2 * S, 12
    mean, noise = np.zeros((H, 7)), _noise(P, H, 7, 9)
    eng.set_env_state([dict(STATES[k], qa=np.zeros(7), timestep=0) for k in range(S)])
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    n = P // S
    for k in range(S):
        st = STATES[k]
        o = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise[k * n:(k + 1) * n])
        np.testing.assert_allclose(rew[k * n:(k + 1) * n], o[1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(obs[k * n:(k + 1) * n], o[0], rtol=0, atol=1e-9)
    # back to one state for everybody
    eng.set_env_state(dict(STATES[1], qa=np.zeros(7), timestep=0))
    _, rew1, _, _, _, _ = eng.rollout(P, H, mean, noise, "open_loop")
    o = ref_arm.rollout(STATES[1]["qp"], STATES[1]["qv"], STATES[1]["target_pos"], mean, noise)
    np.testing.assert_allclose(rew1, o[1], rtol=1e-9, atol=1e-9)
    with pytest.raises(AssertionError):
        eng.set_env_state([dict(STATES[0]), dict(STATES[1])])


def test_cubic_impedance_and_fast_joints(raw_arm):
    """The two small code paths that stand in for library calls in the hot loop: an integer solimp power > 2
    (repeated multiplication instead of pow) with joints driven into their limits, and joint steps beyond
    0.25 rad per substep (doubling formula instead of sincos) from a 60 rad/s spin - f64 against the oracle."""
    import dataclasses
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from oracle.physics_ref import RefArm
    raw = dataclasses.replace(raw_arm, solimp=(0.9, 0.95, 0.001, 0.5, 3.0))
    eng, ref = ArmRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    P, H = 64, 16
    rs = np.random.RandomState(5)
    mean = 0.8 * rs.standard_normal((H, 7))
    noise = _noise(P, H, 7, 31, scale=1.5)
    for st in (STATES[1], dict(qp=np.array([0.0, 0.2, 0.1, -0.4, 0.0, -0.2, 0.0]),
                               qv=np.array([0.0, 0.0, 60.0, 0.0, -45.0, 0.0, 30.0]),
                               target_pos=np.array([0.1, 0.1, 0.1]))):
        eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
        obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
        o_obs, o_rew, o_act, o_done, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
        np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-8)
    assert eng.solver_failures() == 0
    assert ref.newton_stats()["calls"] > 0                    # the limits really were active


def test_get_env_state_after_per_shard_states(raw_arm):
    """SubprocVecEnv.get_env_state returns one state per worker; after a per-shard set_env_state so does the engine."""
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    eng = ArmRolloutEngine(raw_arm, dtype="f64", num_shards=2)
    s0 = dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1]))
    s1 = dict(qp=0.1 * np.ones(7), qv=np.zeros(7), target_pos=np.array([0.2, 0.1, 0.1]))
    eng.set_env_state([s0, s1])
    got = eng.get_env_state()
    assert len(got) == 2 and np.array_equal(got[1]["qp"], s1["qp"]) and np.array_equal(got[0]["target_pos"], s0["target_pos"])
    eng.set_env_state(s0)
    assert len(eng.get_env_state()) == 1


def test_two_wave_and_one_wave_kernels_agree(tmp_path):
    """The same 4096 x 32 rollout through both execution shapes of the arm kernel (MJMPC_ARM_DUO=1: two wavefronts per
    particle group, explicit inverses, Sherman-Morrison re-iteration; =0: one wavefront does everything): costs agree
    to 1e-11 relative - two algebraically different solver organisations on identical inputs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from mjmpc_amd.envs.arm_engine import ArmRolloutEngine\n"
        "from mjmpc_amd.models.reacher7dof import reacher7dof_raw\n"
        "eng = ArmRolloutEngine(reacher7dof_raw(), dtype='f64')\n"
        "eng.set_env_state(dict(qp=np.array([0.0, 0.7, 0.0, -0.2, 0.0, -0.1, 0.0]), qv=np.array([0.0, 1.5, 0, 0, 0, 0, 0.0]),\n"
        "                       target_pos=np.array([0.2, -0.1, -0.25])))\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "noise = torch.randn(4096, 32, 7, device='cuda', dtype=torch.float64, generator=g)\n"
        "c, a, _, _ = eng.rollout_device(4096, 32, np.zeros((32, 7)), noise)\n"
        "np.save(sys.argv[1], c.cpu().numpy()); assert eng.solver_failures() == 0\n" % root)
    out = {}
    for duo in ("0", "1"):
        path = str(tmp_path / ("c%s.npy" % duo))
        subprocess.run([sys.executable, str(script), path], check=True, timeout=300, env=dict(os.environ, MJMPC_ARM_DUO=duo))
        out[duo] = np.load(path)
    np.testing.assert_allclose(out["1"], out["0"], rtol=1e-11, atol=1e-12)
    assert np.abs(out["1"] - out["0"]).max() > 0        # ... and they ARE different programs

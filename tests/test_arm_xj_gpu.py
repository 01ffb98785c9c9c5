"""GPU: the EXTENDED-JOINT build of the arm kernels (csrc/arm_rollout_xj.hip, round 6) - slide joints and
dry friction (friction-loss rows) on the serial-chain kernel, for the reference's classic-control models
(examples/configs/classic_control/cartpole*.yml; cartpole_dyn_randomize.yml:23 randomizes dof_frictionloss): the HIP path
through the C ABI against oracle/reacher_ref.c on the same inputs, 1e-9; and against the general tree engine, which ran these
models before."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# a cart on two slide joints carrying a two-link pendulum whose tip sphere can meet a frictionless floor: slide joints UNDER
# hinges (the contact row's Jacobian has slide entries), limits on slide and hinge dofs, friction loss on three of four dofs,
# a motor on the first two dofs only
GANTRY = """
<mujoco model="gantry">
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.005" gravity="0 0 -9.81" integrator="Euler"/>
  <default>
    <joint damping="0.1" armature="0.01" solreffriction="0.03 1.1"/>
    <geom contype="0" conaffinity="0" density="800" condim="1"/>
  </default>
  <worldbody>
    <site name="target" pos="0.3 0 0.2"/>
    <geom name="floor" type="plane" pos="0 0 -0.55" size="2 2 0.1" contype="1" conaffinity="1" condim="1"/>
    <body name="xcart" pos="0 0 0.6">
      <joint name="sx" type="slide" axis="1 0 0" limited="true" range="-0.6 0.6" frictionloss="0.3"/>
      <geom type="box" size="0.1 0.08 0.05"/>
      <body name="zcart" pos="0 0 0">
        <joint name="sz" type="slide" axis="0 0 1" limited="true" range="-0.5 0.3" frictionloss="0.2"/>
        <geom type="capsule" fromto="0 0 0 0 0 -0.2" size="0.03"/>
        <body name="upper" pos="0 0 -0.2">
          <joint name="h1" type="hinge" axis="0 1 0" limited="true" range="-1.2 1.2" frictionloss="0.05"/>
          <geom type="capsule" fromto="0 0 0 0 0 -0.4" size="0.025"/>
          <body name="lower" pos="0 0 -0.4">
            <joint name="h2" type="hinge" axis="0 1 0"/>
            <geom type="capsule" fromto="0 0 0 0 0 -0.3" size="0.02"/>
            <geom name="tip" type="sphere" pos="0 0 -0.3" size="0.05" contype="1" conaffinity="1"/>
            <site name="finger" pos="0 0 -0.3"/>
          </body>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="sx" gear="20" ctrlrange="-1 1" ctrllimited="true"/>
    <motor joint="sz" gear="30" ctrlrange="-1 1" ctrllimited="true"/>
  </actuator>
</mujoco>
"""


def _models(tmp_path_factory):
    from mjmpc_amd.models.mjcf import load_mjcf
    from mjmpc_amd.models.synthetic import synthetic_raw
    d = tmp_path_factory.mktemp("xj")
    (d / "gantry.xml").write_text(GANTRY)
    return dict(cartpole=synthetic_raw("cartpole"), gantry=load_mjcf(str(d / "gantry.xml"), frame_skip=2))


@pytest.fixture(scope="module", params=["cartpole", "gantry"])
def rig(request, tmp_path_factory):
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _models(tmp_path_factory)[request.param]
    eng = ArmRolloutEngine(raw, dtype="f64")
    assert eng.model.field("jtype").any() and (eng.model.field("frictionloss") > 0).any()
    return request.param, raw, eng, RefArm(raw.to_flat())


def _state(name, raw, rs, k):
    q = raw.qpos0 + rs.uniform(-1.0, 1.0, raw.nq) * (0.6 if name == "gantry" else 1.5)
    v = rs.standard_normal(raw.nv) * (0.0 if k % 5 == 0 else (0.02 if k % 5 == 1 else 2.0))    # at rest / creeping (stiction zone) / moving
    if name == "gantry" and k % 3 == 0:
        q[1], q[2], q[3] = -0.45, 0.1 * rs.standard_normal(), 0.1 * rs.standard_normal()       # the tip near the floor
    return q, v


def test_one_env_step_from_random_states(rig):
    name, raw, eng, ref = rig
    rs = np.random.RandomState(3)
    tgt = np.asarray(raw.target_pos, float)
    nu = len(raw.actuators)
    worst = 0.0
    for k in range(60):
        q, v = _state(name, raw, rs, k)
        u = rs.uniform(-1.3, 1.3, nu) * (0.0 if k % 7 == 0 else 1.0)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("%s on the arm engine: one env step from 60 random states, worst relative error %.2e" % (name, worst))
    assert worst < 1e-9, worst
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0


@pytest.mark.parametrize("P", [64, 4096, 8200])         # DUO (two wavefronts per particle group) twice, SOLO
def test_rollouts_match_oracle(rig, P):
    name, raw, eng, ref = rig
    H, nu = 12, len(raw.actuators)
    rs = np.random.RandomState(5 + P)
    eps = 0.4 * rs.standard_normal((P, H, nu))
    for t in range(2, H):
        eps[:, t] = 0.25 * eps[:, t] + 0.8 * eps[:, t - 1]
    mean = 0.2 * rs.standard_normal((H, nu))
    q, v = _state(name, raw, rs, 2)
    tgt = np.asarray(raw.target_pos, float)
    eng.set_env_state(dict(qp=q, qv=0.3 * v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, eps, "open_loop")
    sl = slice(0, 96)
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(q, 0.3 * v, tgt, mean, eps[sl])
    assert np.array_equal(act[sl], o_act)
    np.testing.assert_allclose(rew[sl], o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs[sl], o_nobs, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_nobs).max()))
    np.testing.assert_allclose(obs[sl], o_obs, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_nobs).max()))
    assert eng.solver_failures() == 0


def test_arm_engine_agrees_with_the_tree_engine_and_is_chosen_by_make_engine(rig):
    """The same model on the general tree engine (which ran it before): costs agree at 1e-9; ``make_engine`` picks the arm one."""
    from mjmpc_amd.envs import make_engine
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    name, raw, eng, ref = rig
    tree = TreeRolloutEngine(raw, dtype="f64")
    P, H, nu = 256, 8, len(raw.actuators)
    rs = np.random.RandomState(1)
    eps = 0.3 * rs.standard_normal((P, H, nu))
    st = dict(qp=raw.qpos0 + 0.3 * rs.standard_normal(raw.nq), qv=rs.standard_normal(raw.nv), target_pos=np.asarray(raw.target_pos, float))
    eng.set_env_state(st)
    tree.set_env_state(st)
    _, r_a, _, _, _, n_a = eng.rollout(P, H, np.zeros((H, nu)), eps, "open_loop")
    _, r_t, _, _, _, n_t = tree.rollout(P, H, np.zeros((H, nu)), eps, "open_loop")
    np.testing.assert_allclose(r_a, r_t, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(n_a, n_t, rtol=1e-8, atol=1e-9)
    assert isinstance(make_engine(raw, dtype="f64"), ArmRolloutEngine)


def test_fused_mppi_iteration_and_device_env_on_the_cartpole():
    """The two-launch MPPI iteration (sampling in the kernel, env step in the finish launch) on the cart-pole = the oracle-driven
    loop on the same Philox samples."""
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.synthetic import start_state, synthetic_raw
    from oracle import controllers_ref as cr
    from oracle.physics_ref import RefArm
    raw = synthetic_raw("cartpole")
    eng, ref = ArmRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    st = start_state("cartpole", raw)
    eng.set_env_state(st)
    P, H, lam, A = 1024, 16, 0.05, 1
    c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=A, horizon=H, init_cov=0.3, base_action="null", lam=lam,
             num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
             action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=3, noise_mode="device")
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c.enable_graph(post_step=eng.step_state)
    q, v, tgt, mean = st["qp"].copy(), st["qv"].copy(), st["target_pos"], np.zeros((H, A))
    for step in range(4):
        a, _ = c.optimize({})
        assert c._mono, "the cart-pole should take the fused two-launch iteration on the arm engine"
        noise = c.dev.sample_noise(P, 0.3 * np.eye(A), [0.25, 0.8, 0.0], 3, step, filtered=True).cpu().numpy()
        _, rew, act, _, _ = ref.rollout(q, v, tgt, mean, noise, want_obs=False)
        mean = cr.mppi_update(-rew, act, mean, 0.3 * np.eye(A), cr.gamma_seq(1.0, H), lam, 1, 1.0)
        np.testing.assert_allclose(a, mean[0], rtol=0, atol=1e-9)
        q, v, _, _ = ref.env_step(q, v, a, tgt)
        mean = cr.shift_mean(mean, "null")
    assert eng.solver_failures() == 0


def test_f32_stays_close(rig):
    """The extended-joint build in single precision: one env step from random states within f32's reach of the f64 oracle (median
    error at single-precision level, nothing non-finite, the zone tests' rounding band keeps the iteration from cycling)."""
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    name, raw, _, ref = rig
    eng = ArmRolloutEngine(raw, dtype="f32")
    rs = np.random.RandomState(11)
    tgt = np.asarray(raw.target_pos, float)
    nu = len(raw.actuators)
    errs = []
    for k in range(40):
        q, v = _state(name, raw, rs, k)
        u = rs.uniform(-1.0, 1.0, nu)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        _, _, r1, o1 = ref.env_step(q, v, u, tgt)
        assert np.isfinite(nobs).all()
        errs.append(np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()))
    errs = np.sort(errs)
    print("%s f32: median %.1e, 95th percentile %.1e, worst %.1e" % (name, errs[len(errs) // 2], errs[int(0.95 * len(errs))], errs[-1]))
    assert errs[len(errs) // 2] < 1e-4 and errs[int(0.95 * len(errs))] < 1e-2
    P, H = 4096, 16
    eps = 0.3 * rs.standard_normal((P, H, nu)).astype(np.float32)
    q, v = _state(name, raw, rs, 2)
    eng.set_env_state(dict(qp=q, qv=0.3 * v, target_pos=tgt))
    _, rew, _, _, _, _ = eng.rollout(P, H, np.zeros((H, nu)), eps, "open_loop")
    assert np.isfinite(rew).all()
    assert eng.solver_failures() <= 2            # (f32: a handful of iteration-cap hits in 1.3e5 particle-substeps at most)

"""GPU: MuJoCo's reset on instability.

``mj_step`` [EXT] begins with ``mj_checkPos`` / ``mj_checkVel`` and follows ``mj_forward`` with ``mj_checkAcc``: a NaN or an entry
beyond mjMAXVAL = 1e10 in qpos / qvel / qacc makes MuJoCo call ``mj_resetData`` (qpos0, zero velocity, zero controls) and go
on - the reference's rollouts (mjmpc/envs/gym_env_wrapper.py:125-153 -> env.step -> sim.step()) therefore return FINITE
costs of a reset simulation for such particles (with mujoco-py's default warning callback the worker would raise instead;
what is emulated is MuJoCo's own behaviour).  The oracle restates that sequence literally (or_step_mj); the kernels reach
the same states with a check at the start of a substep, a check on the acceleration, and a per-model record of the state one
substep after the reset state.  Here: forced blow-ups through both paths, kernel = oracle at 1e-9, resets counted alike."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TGT = np.array([0.15, -0.1, 0.2])


def _arm_case(vel, P, H, seed):
    rs = np.random.RandomState(seed)
    qp = rs.uniform(-0.5, 0.5, 7)
    qv = rs.uniform(-1.0, 1.0, 7) * vel
    mean = 0.3 * rs.standard_normal((H, 7))
    noise = 0.5 * rs.standard_normal((P, H, 7))
    return qp, qv, mean, noise


# velocity scales: none / acceleration check (bias forces ~ v^2 beyond 1e10) / velocity check
@pytest.mark.parametrize("P", [24, 4104])                # two wavefronts per particle group (DUO) / one (SOLO)
@pytest.mark.parametrize("vel", [1.0, 3e5, 1e7, 1e9, 3e10, 1e200, np.nan])
def test_arm_rollout_through_a_reset_equals_the_oracle(raw_arm, ref_arm, vel, P):
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    H = 5
    eng = ArmRolloutEngine(raw_arm, dtype="f64")
    qp, qv, mean, noise = _arm_case(1.0 if np.isnan(vel) else vel, P, H, 11)
    if np.isnan(vel):
        qv[3] = np.nan
    eng.set_env_state(dict(qp=qp, qv=qv, target_pos=TGT))
    obs, rew, act, _, _, nobs = eng.rollout(P, H, mean, noise)
    r0 = ref_arm.resets()
    # (the oracle on a slice: every particle shares the start state, the samples differ)
    sl = slice(0, min(P, 64))
    o_obs, o_rew, o_act, _, o_nobs = ref_arm.rollout(qp, qv, TGT, mean, noise[sl])
    n_or = ref_arm.resets() - r0
    assert np.isfinite(rew).all() and np.isfinite(nobs).all()
    scale = max(1.0, np.abs(o_rew).max())
    np.testing.assert_allclose(rew[sl], o_rew, rtol=1e-9, atol=1e-9 * scale)
    np.testing.assert_allclose(nobs[sl], o_nobs, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_nobs).max()))
    # observations: obs[:, t] = next_obs[:, t - 1]; obs[:, 0] echoes the start state - its site entries are the kinematics of
    # that state in MuJoCo (set_env_state ends with sim.forward(), which checks nothing) and of the state the first substep
    # ran from here, which differ only where the start state itself is one MuJoCo resets (velocities beyond 1e10, NaN)
    np.testing.assert_allclose(obs[sl][:, 1:], o_obs[:, 1:], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_nobs).max()))
    start_ok = bool(np.all(np.abs(qv) <= 1e10))
    np.testing.assert_allclose(obs[sl][:, 0, :14 if not start_ok else 20], o_obs[:, 0, :14 if not start_ok else 20], rtol=1e-9, atol=1e-9,
                               equal_nan=True)
    per_particle = n_or / (sl.stop - sl.start)
    assert eng.diverged_substeps() == int(round(per_particle * P))
    if np.isnan(vel) or vel >= 1e7:
        assert n_or > 0
    if vel == 1.0:
        assert n_or == 0
    assert eng.solver_failures() == 0


def test_arm_per_shard_start_states_reset_independently(raw_arm, ref_arm):
    """Eight shards with eight start states - slow, fast, beyond every bound - in ONE launch: wave-mates that reset and
    wave-mates that do not (8 particles per wavefront, 8 per shard)."""
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    P, H, S = 64, 6, 8
    eng = ArmRolloutEngine(raw_arm, dtype="f64", num_shards=S)
    rs = np.random.RandomState(5)
    mean, noise = 0.3 * rs.standard_normal((H, 7)), 0.5 * rs.standard_normal((P, H, 7))
    vels = [1.0, 1e5, 1e6, 3e7, 1e9, 2e10, 1e11, 1e300]
    states = [dict(qp=rs.uniform(-0.5, 0.5, 7), qv=rs.uniform(-1, 1, 7) * vl, target_pos=TGT) for vl in vels]
    eng.set_env_state(states)
    _, rew, _, _, _, nobs = eng.rollout(P, H, mean, noise)
    assert np.isfinite(rew).all()
    r0, per = ref_arm.resets(), P // S
    for k, st in enumerate(states):
        _, o_rew, _, _, o_nobs = ref_arm.rollout(st["qp"], st["qv"], TGT, mean, noise[k * per:(k + 1) * per])
        np.testing.assert_allclose(rew[k * per:(k + 1) * per], o_rew, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_rew).max()), err_msg=str(vels[k]))
        np.testing.assert_allclose(nobs[k * per:(k + 1) * per], o_nobs, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_nobs).max()))
    assert eng.diverged_substeps() == ref_arm.resets() - r0 > 0


@pytest.mark.parametrize("vel", [1e9, 1e11])
def test_arm_device_env_and_fused_step_through_a_reset(raw_arm, ref_arm, vel):
    """The device-resident real env (mjmpc_arm_step_state) and the two-launch MPPI iteration (sampling in the kernel, env
    step in the finish launch) from a start state MuJoCo resets: actions and states as the oracle-driven loop's."""
    import torch
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from oracle import controllers_ref as cr
    rs = np.random.RandomState(2)
    qp, qv = rs.uniform(-0.5, 0.5, 7), rs.uniform(-1, 1, 7) * vel
    # 1. step_state
    eng = ArmRolloutEngine(raw_arm, dtype="f64")
    eng.set_env_state(dict(qp=qp, qv=qv, target_pos=TGT))
    u = rs.uniform(-1, 1, 7)
    q1, v1, r1, o1 = ref_arm.env_step(qp, qv, u, TGT)
    cost, nobs = eng.step_state(u)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nobs.cpu().numpy(), o1, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(float(cost.cpu()[0]), -r1, rtol=1e-9)
    assert eng.diverged_substeps() >= 1
    # 2. the fused iteration, two control steps
    P, H, lam = 512, 8, 0.5
    eng = ArmRolloutEngine(raw_arm, dtype="f64")
    eng.set_env_state(dict(qp=qp, qv=qv, target_pos=TGT))
    c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=H, init_cov=0.4, base_action="null", lam=lam,
             num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
             action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=3, noise_mode="device")
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c.enable_graph(post_step=eng.step_state)
    q, v, mean = qp.copy(), qv.copy(), np.zeros((H, 7))
    for step in range(2):
        a, _ = c.optimize({})
        noise = c.dev.sample_noise(P, 0.4 * np.eye(7), [0.25, 0.8, 0.0], 3, step, filtered=True).cpu().numpy()
        _, rew, act, _, _ = ref_arm.rollout(q, v, TGT, mean, noise, want_obs=False)
        mean = cr.mppi_update(-rew, act, mean, 0.4 * np.eye(7), cr.gamma_seq(1.0, H), lam, 1, 1.0)
        np.testing.assert_allclose(a, mean[0], rtol=0, atol=1e-9)
        q, v, _, _ = ref_arm.env_step(q, v, mean[0], TGT)
        mean = cr.shift_mean(mean, "null")
    torch.cuda.synchronize()
    assert c._mono          # (step 2's action came from the state the finish launch's env step left: the reset env's)
    _, nobs = eng.step_state(np.zeros(7))
    q, v, _, o = ref_arm.env_step(q, v, np.zeros(7), TGT)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nobs.cpu().numpy(), o, rtol=1e-9, atol=1e-11)


def _tree_models():
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.hand24 import hand24_raw
    from mjmpc_amd.models.pen_hand import pen_hand_raw
    from mjmpc_amd.models.synthetic import synthetic_raw
    return dict(cheetah=half_cheetah_raw, hand=hand24_raw, pen=pen_hand_raw, cartpole=lambda: synthetic_raw("cartpole"),
                tray=lambda: synthetic_raw("tray"), door=lambda: synthetic_raw("door"))


@pytest.mark.parametrize("vel", [1e6, 1e8, 1e11, 1e200])
@pytest.mark.parametrize("name", ["cheetah", "hand", "pen", "cartpole", "tray", "door"])
def test_tree_rollout_through_a_reset_equals_the_oracle(name, vel):
    """Every execution shape of the tree kernel (lean / full dense 16 lanes / dense 32 lanes / general incl. a free body with
    its quaternion): start states MuJoCo resets at the first substep (velocity check) or after it (acceleration check)."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _tree_models()[name]()
    eng = TreeRolloutEngine(raw, dtype="f64")
    ref = RefArm(raw.to_flat())
    rs = np.random.RandomState(7)
    P, H, A, nv = 37, 4, eng.d_action, eng.model.nv
    qp = ref.qpos0.copy()
    qv = rs.uniform(-1, 1, nv) * vel
    tgt = np.asarray(raw.target_pos, float)
    mean, noise = 0.2 * rs.standard_normal((H, A)), 0.3 * rs.standard_normal((P, H, A))
    eng.set_env_state(dict(qpos=qp, qvel=qv, target_pos=tgt))
    _, rew, _, _, _, nobs = eng.rollout(P, H, mean, noise)
    r0 = ref.resets()
    _, o_rew, _, _, o_nobs = ref.rollout(qp, qv, tgt, mean, noise)
    n_or = ref.resets() - r0
    assert np.isfinite(rew).all() and n_or >= P
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_rew).max()))
    np.testing.assert_allclose(nobs, o_nobs, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(o_nobs).max()))
    assert eng.diverged_substeps() == n_or
    assert eng.solver_failures() == 0


def test_tree_device_env_step_through_a_reset():
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _tree_models()["tray"]()
    eng = TreeRolloutEngine(raw, dtype="f64")
    ref = RefArm(raw.to_flat())
    rs = np.random.RandomState(9)
    qp, qv, tgt = ref.qpos0.copy(), rs.uniform(-1, 1, eng.model.nv) * 1e9, np.asarray(raw.target_pos, float)
    eng.set_env_state(dict(qpos=qp, qvel=qv, target_pos=tgt))
    u = rs.uniform(-0.2, 0.2, eng.d_action)
    q1, v1, _, _ = ref.env_step(qp, qv, u, tgt)
    eng.step_state(u)
    # (the controlled env itself blows up here: by default the engine raises where the host reads the state back - the
    # reference's MujocoException; this test is about the emulated state behind it)
    from mjmpc_amd.envs.tree_engine import SimulationUnstableError
    with pytest.raises(SimulationUnstableError):
        eng.get_state_device()
    eng.on_env_reset = "ignore"
    got = eng.get_state_device()
    np.testing.assert_allclose(got["qp" if "qp" in got else "qpos"], q1, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(got["qv" if "qv" in got else "qvel"], v1, rtol=1e-9, atol=1e-11)
    assert eng.diverged_substeps() >= 1


# ---- round 6: the real env's resets are surfaced, rollouts may keep +inf returns -----------------------------
def test_real_env_reset_is_counted_apart_and_raised(raw_arm):
    """The reference raises MujocoException out of sim.step() when the CONTROLLED env blows up; here the device-resident env
    counts its resets apart from the rollouts' (mjmpc_arm_env_resets) and the engine raises where the host synchronises."""
    import torch
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, SimulationUnstableError
    rs = np.random.RandomState(4)
    qp, qv = rs.uniform(-0.5, 0.5, 7), rs.uniform(-1, 1, 7) * 1e11
    eng = ArmRolloutEngine(raw_arm, dtype="f64")
    assert eng.on_env_reset == "raise" and eng.env_resets() == 0
    # rollouts from a state MuJoCo resets: counted as rollout resets only
    eng.set_env_state(dict(qp=qp, qv=qv, target_pos=TGT))
    eng.rollout(16, 3, np.zeros((3, 7)), 0.1 * rs.standard_normal((16, 3, 7)))
    assert eng.diverged_substeps() > 0 and eng.env_resets() == 0
    assert eng.check_env_resets() == 0
    # the device-resident env stepped from it: counted as the real env's, raised at the check
    eng.step_state(np.zeros(7))
    torch.cuda.synchronize()
    assert eng.env_resets() >= 1
    with pytest.raises(SimulationUnstableError, match="MujocoException"):
        eng.check_env_resets()
    assert eng.check_env_resets() == 0                      # (reported once)
    eng.on_env_reset = "warn"
    eng.set_env_state(dict(qp=qp, qv=qv, target_pos=TGT))
    eng.step_state(np.zeros(7))
    with pytest.warns(UserWarning, match="was reset"):
        assert eng.check_env_resets() >= 1


def test_host_env_step_raises_on_a_reset(raw_arm):
    """Reacher7DOFEnv.step (a host-synchronous one-particle rollout) from a state MuJoCo resets."""
    from mjmpc_amd.envs.arm_engine import SimulationUnstableError
    from mjmpc_amd.envs.reacher_env import Reacher7DOFEnv
    env = Reacher7DOFEnv()
    env.reset(seed=0)
    env.step(np.zeros(7))                                   # a sane step passes
    st = env.get_env_state()
    st["qv"] = np.full(7, 1e11)
    env.set_env_state(st)
    with pytest.raises(SimulationUnstableError):
        env.step(np.zeros(7))
    env.engine.on_env_reset = "ignore"
    env.set_env_state(st)
    ob, rew, done, info = env.step(np.zeros(7))
    assert np.isfinite(ob).all() and np.isfinite(rew)


@pytest.mark.parametrize("P", [64, 4096, 8192])               # four waves + flags / DUO / SOLO launches
def test_reset_returns_inf_option_arm(raw_arm, P):
    """engine.set_reset_returns("inf"): particles that reset cost +inf from the env step of the reset on, the others are
    untouched; with per-shard start states only the shards that blow up are affected."""
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    H, S = 6, 8
    Pq = (P // S) * S
    eng = ArmRolloutEngine(raw_arm, dtype="f64", num_shards=S)
    rs = np.random.RandomState(9)
    mean, noise = 0.3 * rs.standard_normal((H, 7)), 0.5 * rs.standard_normal((Pq, H, 7))
    vels = [1.0, 3.0, 1.0, 1e9, 1.0, 2e10, 1.0, 1e300]
    states = [dict(qp=rs.uniform(-0.5, 0.5, 7), qv=rs.uniform(-1, 1, 7) * vl, target_pos=TGT) for vl in vels]
    eng.set_env_state(states)
    _, rew_f, _, _, _, _ = eng.rollout(Pq, H, mean, noise)
    assert np.isfinite(rew_f).all()
    eng.set_reset_returns("inf")
    _, rew_i, _, _, _, _ = eng.rollout(Pq, H, mean, noise)
    per = Pq // S
    for k, vl in enumerate(vels):
        blk_f, blk_i = rew_f[k * per:(k + 1) * per], rew_i[k * per:(k + 1) * per]
        if vl <= 3.0:
            assert np.array_equal(blk_f, blk_i), vl                     # no reset: bit-identical
        else:
            assert np.isinf(blk_i[:, -1]).all() and (blk_i[:, -1] < 0).all(), vl   # reward = -cost = -inf at the end ...
            first = np.argmax(np.isinf(blk_i), axis=1)
            for p in range(min(per, 8)):                                # ... from the reset's env step on, finite before
                assert np.isinf(blk_i[p, first[p]:]).all() and np.array_equal(blk_i[p, :first[p]], blk_f[p, :first[p]])
    eng.set_reset_returns("finite")
    _, rew_b, _, _, _, _ = eng.rollout(Pq, H, mean, noise)
    assert np.array_equal(rew_b, rew_f)


def test_reset_returns_inf_option_tree():
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.hand24 import hand24_raw
    raw = hand24_raw()
    eng = TreeRolloutEngine(raw, dtype="f64")
    rs = np.random.RandomState(3)
    P, H, A = 16, 4, eng.d_action
    qp = np.asarray(raw.qpos0, float).copy()
    eng.set_env_state(dict(qp=qp, qv=1e11 * rs.uniform(-1, 1, eng.model.nv), target_pos=np.asarray(raw.target_pos, float)))
    noise = 0.1 * rs.standard_normal((P, H, A))
    _, rew_f, _, _, _, _ = eng.rollout(P, H, np.zeros((H, A)), noise)
    assert np.isfinite(rew_f).all() and eng.diverged_substeps() > 0
    eng.set_reset_returns("inf")
    _, rew_i, _, _, _, _ = eng.rollout(P, H, np.zeros((H, A)), noise)
    assert np.isinf(rew_i).all()                            # (the start state itself resets: every env step costs +inf)
    # and the device-resident env
    assert eng.env_resets() == 0
    eng.step_state(np.zeros(A))
    assert eng.env_resets() >= 1

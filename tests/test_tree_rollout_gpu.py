"""GPU parity of the TREE rollout kernel (mjmpc_amd/csrc/tree_rollout.hip, through the C ABI) against the FP64 C
oracle, which walks the same parent-indexed tree with an independent formulation (Jacobian-built mass matrix,
inertial-frame Newton-Euler, dense Cholesky).  Model: the synthetic 24-dof hand-on-an-arm tree
(mjmpc_amd/models/hand24.py: branching, gravity, limits on every joint, five fingertip spheres over a table).
Tolerance: f64 costs rel <= 1e-9, observations abs <= 1e-9 (SURVEY 8d's gate for the f64 kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hand():
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.hand24 import hand24_raw
    from oracle.physics_ref import RefArm
    raw = hand24_raw()
    return raw, TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())


def _noise(P, H, A, seed, scale):
    rs = np.random.RandomState(seed)
    eps = scale * rs.standard_normal((P, H, A))
    for t in range(2, H):
        eps[:, t] = 0.25 * eps[:, t] + 0.8 * eps[:, t - 1]
    return eps


STATES = [
    dict(qp=np.zeros(24), qv=np.zeros(24)),
    # arm lowered so that the fingertips press on the table during the rollout, fingers half curled and moving
    dict(qp=np.concatenate([[0.2, 0.55, -0.3, 0.2], np.tile([0.1, 0.4, 0.5, 0.3], 5)]),
         qv=np.concatenate([[0.3, 1.0, -0.5, 0.2], np.tile([0.5, -1.0, 2.0, 1.0], 5)])),
]


@pytest.mark.parametrize("si", range(len(STATES)))
def test_tree_f64_matches_oracle(hand, si):
    raw, eng, ref = hand
    st = dict(STATES[si], target_pos=np.array(raw.target_pos))
    P, H, A = 101, 24, 24                    # P odd: a half-empty last wavefront
    mean = 0.2 * np.random.RandomState(3 + si).standard_normal((H, A))
    noise = _noise(P, H, A, 30 + si, 0.7)
    eng.set_env_state(dict(st, qa=np.zeros(24), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o_obs, o_rew, o_act, o_done, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    assert np.array_equal(act, o_act)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


def test_tree_contact_case_really_touches(hand):
    raw, eng, ref = hand
    st = STATES[1]
    flat = raw.to_flat()
    # the oracle's Newton statistics distinguish substeps with constraint rows; here: at least one fingertip centre
    # comes within radius + margin of the table along the mean-only rollout
    from mjmpc_amd.models.compile_tree import compile_tree
    m = compile_tree(raw)
    obs, rew, act, done, nobs = ref.rollout(st["qp"], st["qv"], np.array(raw.target_pos), np.zeros((24, 24)), None)
    assert np.isfinite(rew).all()
    assert m.field("n_sphere")[0] == 5
    hand_z = nobs[0, :, 2 * 24 + 2]
    assert hand_z.min() < -0.12 + 0.03       # index fingertip site within 3 cm of the table plane


def test_tree_engine_serial_chain_equals_arm_engine(raw_arm, ref_arm):
    """The tree kernel on the serial 7-dof arm (a tree without branches) reproduces the arm oracle too."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    eng = TreeRolloutEngine(raw_arm, dtype="f64")
    st = dict(qp=np.array([0.0, 0.7, 0.0, -0.2, 0.0, -0.1, 0.0]), qv=np.array([0.0, 1.5, 0.0, 0.0, 0.0, 0.0, 0.0]),
              target_pos=np.array([0.2, -0.1, -0.25]))          # touches the table (arm test state 2)
    P, H = 64, 32
    noise = _noise(P, H, 7, 5, 1.0)
    mean = np.zeros((H, 7))
    eng.set_env_state(dict(st, qa=np.zeros(7), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o_obs, o_rew, o_act, o_done, o_nobs = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


def test_tree_f32_within_stated_tolerance(hand):
    """f32 build of the tree kernel: measured against the FP64 oracle, bound stated here (costs 5e-3 absolute on a
    24-dof tree with gravity over 48 substeps)."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    raw, _, ref = hand
    eng = TreeRolloutEngine(raw, dtype="f32")
    st = dict(STATES[0], target_pos=np.array(raw.target_pos))
    P, H, A = 128, 24, 24
    noise = _noise(P, H, A, 9, 0.5).astype(np.float32).astype(np.float64)
    mean = np.zeros((H, A))
    eng.set_env_state(dict(st, qa=np.zeros(24), timestep=0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    err = np.abs(rew - o[1])
    print("tree f32 cost error: max %.3e mean %.3e" % (err.max(), err.mean()))
    assert err.max() < 5e-3


def test_tree_pen_config_size_65536x64(hand):
    """BASELINE config 5's size on the synthetic tree (one GPU): a 65536 x 64 rollout; size-independent properties
    on everything (determinism of duplicated particles, obs[t] = next_obs[t-1], cost = f(hand - target)) and the
    oracle itself on every 1021st particle at 1e-9."""
    raw, eng, ref = hand
    import torch
    P, H, A = 65536, 64, 24
    st = dict(STATES[1], target_pos=np.array(raw.target_pos))
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = 0.5 * torch.randn(P, H, A, device="cuda", dtype=torch.float64, generator=g)
    noise[P // 2:] = noise[:P // 2]
    mean = np.zeros((H, A))
    eng.set_env_state(dict(st, qa=np.zeros(24), timestep=0))
    costs, act, obs, nobs = eng.rollout_device(P, H, mean, noise, want_obs=True)
    assert torch.equal(costs[:P // 2], costs[P // 2:])
    assert torch.equal(obs[:, 1:], nobs[:, :-1])
    d = nobs[..., 2 * A + 3:2 * A + 6]
    want = d.abs().sum(-1) + 5 * (d * d).sum(-1).sqrt()
    assert torch.isfinite(costs).all() and float((costs - want).abs().max()) < 1e-12
    idx = np.arange(0, P // 2, 1021)
    sub = noise[idx].cpu().numpy()
    _, o_rew, _, _, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, sub)
    np.testing.assert_allclose(costs[idx].cpu().numpy(), -o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs[idx].cpu().numpy(), o_nobs, rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


def test_dmd_step_on_the_tree(hand):
    """DMD-MPC (the controller BASELINE config 5 names) over the tree engine: optimize() = dmd_update on oracle
    rollouts from the same host noise, 512 x 16."""
    from mjmpc_amd.control import DMDMPC
    from mjmpc_amd.envs.arm_engine import make_rollout_fn
    from oracle import controllers_ref as cr
    raw, eng, ref = hand
    P, H, A = 512, 16, 24
    ctrl = DMDMPC(d_state=eng.d_state, d_obs=eng.d_obs, d_action=A, horizon=H, init_cov=0.3, beta=0.1, base_action="null",
                  lam=0.5, num_particles=P, step_size=0.9, gamma=0.99, n_iters=1, update_cov=True, cov_type="diagonal",
                  action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=11)
    ctrl.set_sim_state_fn = eng.set_env_state
    ctrl.rollout_fn = make_rollout_fn(eng)
    st = dict(STATES[0], target_pos=np.array(raw.target_pos), qa=np.zeros(24), timestep=0)
    action, _ = ctrl.optimize(st)
    mean0, cov0 = np.zeros((H, A)), 0.3 * np.eye(A)
    noise = cr.generate_noise(cov0, [0.25, 0.8, 0.0], (P, H), 11)
    _, rew, act, _, _ = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean0, noise, want_obs=False)
    mean1, cov1 = cr.dmd_update(-rew, act, mean0, cov0, cr.gamma_seq(0.99, H), 0.5, 0.9, True, "diagonal")
    np.testing.assert_allclose(action, mean1[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ctrl.mean_action, cr.shift_mean(mean1, "null"), rtol=0, atol=1e-9)
    np.testing.assert_allclose(ctrl.cov_action, cr.dmd_shift_cov(cov1, 0.1, True), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n_links,full", [(12, False), (20, False), (12, True), (20, True)])
def test_deep_chains_use_the_wider_row_instantiations(n_links, full):
    """Root-to-leaf paths longer than 8 links run the DP = 16 / DP = 32 instantiations of the tree kernel (path-indexed
    rows of 16 / 32 entries): a 12-link and a 20-link chain with a two-link side branch, gravity on, against the oracle.
    `full`: joint springs and a friction cone on the contact sphere send the same models through the full instantiation
    (14 dofs: 16 lanes per particle, 22 dofs: 32)."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.raw import GEOM_CAPSULE, GEOM_SPHERE, RawActuator, RawBody, RawGeom, RawJoint, RawModel, RawPlane
    from oracle.physics_ref import RefArm
    axes = [(0, 0, 1), (0, 1, 0), (1, 0, 0)]
    bodies = []
    for i in range(n_links):
        r = 0.03 - 0.001 * i
        bodies.append(RawBody("l%d" % i, i - 1, (0.0, 0.0, 0.3) if i == 0 else (0.08, 0.0, 0.0),
                              joint=RawJoint(axes[i % 3], (-1.2, 1.2), True, 0.3, 0.002, "j%d" % i,
                                             stiffness=(2.0 if full and i % 4 == 1 else 0.0), springref=0.1),
                              geoms=[RawGeom(GEOM_CAPSULE, r, (0, 0, 0), (0.08, 0, 0), margin=0.001)]))
    # a side branch half way up, so that the model is a tree and not a chain
    mid = n_links // 2
    bodies.insert(mid + 1, RawBody("b0", mid, (0.0, 0.05, 0.0), joint=RawJoint((0, 0, 1), (-1, 1), True, 0.2, 0.001, "jb0"),
                                   geoms=[RawGeom(GEOM_CAPSULE, 0.015, (0, 0, 0), (0, 0.06, 0), margin=0.001)]))
    bodies.insert(mid + 2, RawBody("b1", mid + 1, (0.0, 0.06, 0.0), joint=RawJoint((1, 0, 0), (-1, 1), True, 0.2, 0.001, "jb1"),
                                   geoms=[RawGeom(GEOM_SPHERE, 0.02, (0, 0.03, 0), collide=True, margin=0.001,
                                                  friction=0.6, condim=3 if full else 1)]))
    for b in bodies[mid + 3:]:                      # the rest of the main chain hangs off link `mid`, after the branch
        b.parent = b.parent + 2 if b.parent > mid else b.parent
    bodies[mid + 3].parent = mid
    nv = len(bodies)
    raw = RawModel(bodies=bodies, actuators=[RawActuator(b.joint.name, 0.5, (-1, 1)) for b in bodies],
                   site_body=nv - 1, site_pos=(0.08, 0, 0), target_pos=(0.5, 0.2, 0.4),
                   plane=RawPlane((0, 0, -0.05), (0, 0, 1), 0.001, friction=0.3, condim=3 if full else 1), timestep=0.004,
                   frame_skip=2, gravity=(0, 0, -9.81))
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    assert eng.model.field("any_friction")[0] == (1.0 if full else 0.0)
    assert eng.model.max_path == n_links and nv == n_links + 2          # 12 -> DP = 16, 20 -> DP = 32
    rs = np.random.RandomState(n_links)
    P, H = 66, 12
    q0, v0 = 0.3 * rs.randn(nv), 0.5 * rs.randn(nv)
    noise, mean = 0.5 * rs.standard_normal((P, H, nv)), 0.1 * rs.standard_normal((H, nv))
    eng.set_env_state(dict(qp=q0, qv=v0, target_pos=np.array(raw.target_pos)))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    o = ref.rollout(q0, v0, np.array(raw.target_pos), mean, noise)
    np.testing.assert_allclose(rew, o[1], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o[4], rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


def test_hand_with_friction_runs_the_full_kernel_at_32_lanes():
    """The 24-dof hand with friction cones on its fingertips and springs in the finger joints: more than 16 dofs, so the
    full instantiation runs 32 lanes per particle (DP = 8); start state with the fingertips pressed on the table."""
    import dataclasses
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.hand24 import hand24_raw
    from oracle.physics_ref import RefArm
    raw = hand24_raw()
    for b in raw.bodies:
        if b.joint is not None and b.name.endswith("_mid"):
            b.joint.stiffness, b.joint.springref = 0.05, 0.3
        for g in b.geoms:
            if g.collide:
                g.friction, g.condim = 0.8, 3
    raw.plane = dataclasses.replace(raw.plane, friction=0.5, condim=3)
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    assert eng.model.field("any_friction")[0] == 1.0 and eng.model.nv == 24
    st = STATES[1]
    P, H, A = 67, 16, 24
    mean = 0.2 * np.random.RandomState(5).standard_normal((H, A))
    noise = _noise(P, H, A, 50, 0.7)
    eng.set_env_state(dict(st, target_pos=np.array(raw.target_pos)))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise)
    o = ref.rollout(st["qp"], st["qv"], np.array(raw.target_pos), mean, noise)
    assert ref.newton_stats()["iters"] > 0
    np.testing.assert_allclose(rew, o[1], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o[4], rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


@pytest.mark.parametrize("model", ["hand", "cheetah", "swimmer"])
def test_closed_loop_linear_mode_on_the_tree_engine(model):
    """``rollout(mode="closed_loop_linear")`` (gym_env_wrapper.py:135-136) on the tree engine: the nominal action of a step
    is weights' [observation the step starts from; 1] - reach-task observation incl. the lagging site on the hand, the
    forward task's [qpos[skip:], qvel] on the locomotion models - against the oracle's closed-loop rollout."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.hand24 import hand24_raw
    from mjmpc_amd.models.swimmer import swimmer_raw
    from oracle.physics_ref import RefArm
    raw = dict(hand=hand24_raw, cheetah=half_cheetah_raw, swimmer=swimmer_raw)[model]()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    nv, A, dobs = eng.model.nv, eng.d_action, eng.d_obs
    rs = np.random.RandomState(21)
    P, H = 37, 6
    W = 0.15 * rs.standard_normal((dobs + 1, A))
    noise = 0.3 * rs.standard_normal((P, H, A))
    if model == "hand":
        st = dict(STATES[1], target_pos=np.array(raw.target_pos))
        q0, v0, tgt = st["qp"], st["qv"], st["target_pos"]
    else:
        q0, v0, tgt = 0.1 * rs.standard_normal(nv), 0.5 * rs.standard_normal(nv), np.zeros(3)
        if model == "cheetah":
            q0[1] = -0.1
        st = dict(qpos=q0, qvel=v0)
    eng.set_env_state(st)
    obs, rew, act, done, info, nobs = eng.rollout(P, H, W, noise, "closed_loop_linear")
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(q0, v0, tgt, W, noise, mode="closed_loop_linear")
    np.testing.assert_allclose(act, o_act, rtol=0, atol=1e-9)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    assert np.abs(act - noise).max() > 1e-2                 # the feedback term is really there
    # mean-only closed-loop rollout (noise None), and the open-loop mode is untouched by the state vector's new tail
    obs1, rew1, act1, _, _, _ = eng.rollout(1, H, W, None, "closed_loop_linear")
    o1 = ref.rollout(q0, v0, tgt, W, None, mode="closed_loop_linear", horizon=H)
    np.testing.assert_allclose(rew1, o1[1], rtol=1e-9, atol=1e-9)
    with pytest.raises(ValueError):
        eng.rollout(P, H, W, noise, "closed_loop_quadratic")


def test_per_shard_start_states_on_the_tree_engine():
    """``SubprocVecEnv.set_env_state`` with one state dict per worker (subproc_vec_env.py:242-251) on the tree engine:
    shard k's particles start from states[k] (open loop and closed_loop_linear), ``get_env_state`` returns one state per
    shard, and a single dict brings every shard back to one state."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.hand24 import hand24_raw
    from oracle.physics_ref import RefArm
    raw = hand24_raw()
    S = 2
    eng, ref = TreeRolloutEngine(raw, dtype="f64", num_shards=S), RefArm(raw.to_flat())
    tgt = np.array(raw.target_pos)
    sts = [dict(STATES[k], target_pos=tgt + 0.05 * k) for k in range(S)]
    P, H, A = 12, 6, 24
    mean = 0.2 * np.random.RandomState(8).standard_normal((H, A))
    noise = _noise(P, H, A, 77, 0.5)
    eng.set_env_state([dict(s, qa=np.zeros(24), timestep=0) for s in sts])
    got = eng.get_env_state()
    assert len(got) == S and np.array_equal(got[1]["qp"], sts[1]["qp"]) and np.array_equal(got[1]["target_pos"], sts[1]["target_pos"])
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise, "open_loop")
    n = P // S
    for k, st in enumerate(sts):
        o = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise[k * n:(k + 1) * n])
        np.testing.assert_allclose(rew[k * n:(k + 1) * n], o[1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(obs[k * n:(k + 1) * n], o[0], rtol=0, atol=1e-9)
        np.testing.assert_allclose(nobs[k * n:(k + 1) * n], o[4], rtol=0, atol=1e-9)
    assert np.abs(rew[:n] - rew[n:]).max() > 1e-3                     # the shards really differ
    # closed_loop_linear: the fresh observation's site is taken per start state
    W = 0.05 * np.random.RandomState(9).standard_normal((eng.d_obs + 1, A))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, W, noise, "closed_loop_linear")
    for k, st in enumerate(sts):
        o = ref.rollout(st["qp"], st["qv"], st["target_pos"], W, noise[k * n:(k + 1) * n], mode="closed_loop_linear")
        np.testing.assert_allclose(rew[k * n:(k + 1) * n], o[1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(act[k * n:(k + 1) * n], o[2], rtol=1e-9, atol=1e-9)
    # back to one state for everybody
    eng.set_env_state(dict(sts[1], qa=np.zeros(24), timestep=0))
    assert len(eng.get_env_state()) == 1
    _, rew1, _, _, _, _ = eng.rollout(P, H, mean, noise, "open_loop")
    o = ref.rollout(sts[1]["qp"], sts[1]["qv"], sts[1]["target_pos"], mean, noise)
    np.testing.assert_allclose(rew1, o[1], rtol=1e-9, atol=1e-9)
    with pytest.raises(AssertionError):
        eng.set_env_state([dict(sts[0]), dict(sts[1]), dict(sts[0])])
    assert eng.solver_failures() == 0

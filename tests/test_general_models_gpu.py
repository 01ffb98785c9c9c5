"""GPU parity of the tree kernel's GENERAL instantiation (round 4, SURVEY 8f rank 4 / VERDICT r3 next #1) against the FP64
C oracle on four synthetic MJCF models written from scratch (mjmpc_amd/models/assets): friction-loss rows (cart-pole),
free and ball joints with quaternion state, box geoms and sphere / box pairs, explicit inertials, static geoms (tray,
door), joint anchors off the body origin, angles in degrees, joint and connect equalities, a limited fixed tendon (door,
four-bar).  Tolerances: one env step from random states 1e-9 (SURVEY 8d's gate); rollouts of 12-16 env steps: the stated
per-model tolerance (contact and closed-loop dynamics amplify rounding along a rollout like the cheetah's)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# (name:cone - the model with its friction cones switched: gripper.xml itself asks for elliptic cones with impratio 2)
NAMES = ["cartpole", "door", "tray", "fourbar", "gripper", "gripper:pyramidal", "tray:elliptic"]


def _quat(rs, scale):
    w = scale * rs.standard_normal(3)
    a = np.linalg.norm(w)
    return np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * w / max(a, 1e-12)])


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def random_state(name, raw, rs, big=1.0):
    """A random state around the model's working point, qpos in MuJoCo's layout."""
    q, v = raw.qpos0.copy(), np.zeros(raw.nv)
    if name == "cartpole":
        q[:] = [rs.uniform(-1.9, 1.9), rs.uniform(-np.pi, np.pi)]          # (beyond the slider's range now and then)
        v[:] = rs.standard_normal(2) * [1.0, 3.0] * rs.choice([0.0, 0.01, 1.0])     # (at rest: the friction rows hold)
    elif name == "door":
        handle = rs.uniform(-0.1, 1.1)
        q[:] = [rs.uniform(-0.05, 1.6), handle, -0.02865 * handle + 0.003 * rs.standard_normal()]
        v[:] = rs.standard_normal(3) * [0.5, 1.0, 0.05] * big
    elif name == "tray":
        q[0:3] += rs.standard_normal(3) * [0.02, 0.02, 0.004]
        q[3:7] = _quat(rs, 0.15)
        q[7:11] = 0.15 * rs.standard_normal(4)
        v[:] = rs.standard_normal(10) * np.r_[0.1 * np.ones(3), 0.5 * np.ones(3), 0.3 * np.ones(4)] * big
    elif name == "gripper":
        # the pen over / on / in the fingers at any attitude, the can standing, tilted or lying, the fingers anywhere in their range
        q[0:3] += rs.standard_normal(3) * [0.03, 0.01, 0.006]
        q[3:7] = _quat(rs, rs.choice([0.0, 0.05, 0.6]))
        q[7:10] += rs.standard_normal(3) * [0.02, 0.02, 0.004]
        q[10:14] = _quat(rs, rs.choice([0.0, 0.1, 1.6]))
        q[14:17] = [rs.uniform(-0.05, 0.1), rs.uniform(-0.1, 0.5), rs.uniform(-0.1, 0.5)]
        v[:] = rs.standard_normal(15) * np.r_[0.2 * np.ones(3), 1.0 * np.ones(3), 0.2 * np.ones(3), 1.0 * np.ones(3), 0.2, 1.0, 1.0] * big * rs.choice([0.0, 1.0])
    else:
        q[0:3] = 0.06 * rs.standard_normal(3)                              # the loop slightly open: the connect rows pull
        q[3:7] = _quat(rs, 1.0)
        v[:] = rs.standard_normal(6) * np.r_[0.5 * np.ones(3), 2.0 * np.ones(3)] * big
    return q, v


@pytest.fixture(scope="module", params=NAMES)
def rig(request):
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.synthetic import synthetic_raw
    from oracle.physics_ref import RefArm
    name, _, cone = request.param.partition(":")
    raw = synthetic_raw(name)
    if cone:
        raw.cone, raw.impratio = cone, (1.0 if cone == "pyramidal" else 5.0)
    assert raw.cone == ("elliptic" if (cone == "elliptic" or request.param == "gripper") else "pyramidal")
    eng = TreeRolloutEngine(raw, dtype="f64")
    assert eng.model.general
    return name, raw, eng, RefArm(raw.to_flat())


def test_one_env_step_from_random_states(rig):
    name, raw, eng, ref = rig
    rs = np.random.RandomState(3)
    tgt = np.asarray(raw.target_pos, float)
    nu = len(raw.actuators)
    worst, rows = 0.0, 0
    for k in range(48):
        q, v = random_state(name, raw, rs)
        u = rs.uniform(-1.2, 1.2, nu) * (eng.action_highs - eng.action_lows) / 2
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        if name in ("tray", "fourbar"):         # q and -q are one rotation: compare up to the sign the integration keeps
            assert np.sign(nobs[0, 0, 3 if name == "tray" else 3]) == np.sign(o1[3])
        err = np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max())
        worst = max(worst, err, abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("%s: one env step from 48 random states, worst relative error %.2e" % (name, worst))
    assert worst < 1e-9, worst
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0


ROLLOUT_TOL = dict(cartpole=1e-9, door=1e-8, tray=1e-7, fourbar=1e-8, gripper=1e-7)


def test_rollouts_match_oracle(rig):
    """64 particles x 12 env steps of filtered noise from the model's start state: costs, observations, next observations."""
    from mjmpc_amd.models.synthetic import start_state
    name, raw, eng, ref = rig
    st = start_state(name, raw)
    P, H, nu = 64, 12, len(raw.actuators)
    rs = np.random.RandomState(5)
    eps = 0.3 * rs.standard_normal((P, H, nu)) * (eng.action_highs - eng.action_lows) / 2
    for t in range(2, H):
        eps[:, t] = 0.25 * eps[:, t] + 0.8 * eps[:, t - 1]
    mean = np.zeros((H, nu))
    eng.set_env_state(st)
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, eps, "open_loop")
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], mean, eps)
    assert np.array_equal(act, o_act)
    tol = ROLLOUT_TOL[name]
    np.testing.assert_allclose(obs[:, 0], o_obs[:, 0], rtol=0, atol=1e-12)          # the fresh observation (site included)
    np.testing.assert_allclose(rew[:, :2], o_rew[:, :2], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rew, o_rew, rtol=tol, atol=tol)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=tol * 10)
    np.testing.assert_array_equal(obs[:, 1:], nobs[:, :-1])
    print("%s: 64 x 12 rollouts, cost error max %.2e, observation error max %.2e"
          % (name, np.abs(rew - o_rew).max(), np.abs(nobs - o_nobs).max()))
    assert eng.solver_failures() == 0


def test_device_resident_env_and_state_round_trip(rig):
    """set_env_state / get_state_device in MuJoCo's qpos layout (quaternions, absolute free-joint positions), and the
    device-resident real env (step_state) against the oracle over five env steps."""
    import torch
    name, raw, eng, ref = rig
    rs = np.random.RandomState(9)
    tgt = np.asarray(raw.target_pos, float)
    q, v = random_state(name, raw, rs, big=0.3)
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    st = eng.get_state_device()
    np.testing.assert_allclose(st["qp"], q, rtol=0, atol=1e-15)
    np.testing.assert_allclose(st["qv"], v, rtol=0, atol=0)
    nu = len(raw.actuators)
    for k in range(5):
        u = rs.uniform(-1, 1, nu) * (eng.action_highs - eng.action_lows) / 2
        cost, nobs = eng.step_state(u)
        q, v, r, o = ref.env_step(q, v, u, tgt)
        np.testing.assert_allclose(nobs.cpu().numpy(), o, rtol=0, atol=1e-9)
        np.testing.assert_allclose(float(cost.item()), -r, rtol=1e-9, atol=1e-9)
    st = eng.get_state_device()
    np.testing.assert_allclose(st["qp"], q, rtol=0, atol=1e-9)
    torch.cuda.synchronize()


def test_friction_loss_randomization_per_shard():
    """``dof_frictionloss`` randomization (reference examples/configs/classic_control/cartpole_dyn_randomize.yml:23,
    gym_env_wrapper.py:387-389) - refused until round 3 - runs: four shards of the cart-pole, each with its own draw of
    masses, inertias, damping and friction loss, each against the oracle edited through its own setters (1e-9)."""
    import yaml, os
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.compile import principal_inertia
    from mjmpc_amd.models.synthetic import start_state, synthetic_raw
    from oracle.physics_ref import RefArm
    raw = synthetic_raw("cartpole")
    eng = TreeRolloutEngine(raw, dtype="f64", num_shards=4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "examples", "configs", "cartpole_gpu_dyn_randomize.yml")) as f:
        spec = yaml.safe_load(f)
    default, rand = eng.randomize_dynamics(spec, base_seed=77)
    assert default[0]["dof_frictionloss"] == {"slider": 0.4, "hinge": 0.02}
    fl = [r["dof_frictionloss"]["slider"] for r in rand]
    assert len(set(np.round(fl, 12))) == 4 and all(0.4 * 1.2 * 0.5 <= x <= 0.4 * 1.2 * 1.5 for x in fl)
    st = start_state("cartpole", raw)
    st["qv"] = np.array([0.3, -1.0])
    P, H = 64, 10
    rs = np.random.RandomState(1)
    eps = 0.5 * rs.standard_normal((P, H, 1))
    eng.set_env_state(st)
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, 1)), eps)
    names = [b.name for b in raw.bodies]
    for k in range(4):
        ref = RefArm(raw.to_flat())
        for name, m in rand[k]["body_mass"].items():
            ref.set_body_mass(1 + names.index(name), m)
        mass, ipos, inertia = ref.inertial()
        for name, I3 in rand[k]["body_inertia"].items():
            b = 1 + names.index(name)
            _, V = principal_inertia(inertia[b])
            ref.set_body_inertia(b, V @ np.diag(I3) @ V.T)
        for j, name in enumerate(("slider", "hinge")):
            ref.set_dof_damping(j, rand[k]["dof_damping"][name])
            ref.set_dof_frictionloss(j, rand[k]["dof_frictionloss"][name])
        sl = slice(k * P // 4, (k + 1) * P // 4)
        _, o_rew, _, _, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], np.zeros((H, 1)), eps[sl])
        np.testing.assert_allclose(rew[sl], o_rew, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(nobs[sl], o_nobs, rtol=0, atol=1e-9)
    assert not np.allclose(rew[:16], rew[16:32])                # the shards do simulate different carts
    assert eng.solver_failures() == 0


@pytest.mark.parametrize("cfg,controller,needle", [("cartpole_gpu.yml", "mppi", None), ("tray_gpu.yml", "mppi", None),
                                                    ("door_gpu.yml", "dmd", None), ("gripper_gpu.yml", "mppi", None)])
def test_example_driver_runs_the_synthetic_models(tmp_path, cfg, controller, needle):
    """examples/example_mpc.py: a short MPC episode on each of the three synthetic MJCF models (VERDICT r3 next #1 'done')."""
    import os, subprocess, sys, yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "examples", "configs", cfg)) as f:
        exp = yaml.safe_load(f)
    exp["max_ep_length"] = 25
    for block in exp.values():
        if isinstance(block, dict) and "particles_per_cpu" in block:
            block["particles_per_cpu"] = 256
    p = tmp_path / cfg
    p.write_text(yaml.safe_dump(exp))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "example_mpc.py"), "--config", str(p),
                          "--controller", controller, "--noise_mode", "device"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Success Metric" in out.stdout and "solver failures 0" in out.stdout, out.stdout[-800:]


def test_f32_general_instantiation_statistics(rig):
    """The f32 build of the general instantiation against the FP64 oracle: 64 x 8 rollouts, stated tolerance - median cost
    error below 1e-4, all below 5e-2 (contacts and friction zones switch a substep apart in f32 and f64); under elliptic
    cones 1e-3 and 1e-1."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.synthetic import start_state
    name, raw, eng, ref = rig
    e32 = TreeRolloutEngine(raw, dtype="f32")
    st = start_state(name, raw)
    P, H, nu = 64, 8, len(raw.actuators)
    rs = np.random.RandomState(8)
    eps = 0.2 * rs.standard_normal((P, H, nu)) * (eng.action_highs - eng.action_lows) / 2
    e32.set_env_state(st)
    _, rew, _, _, _, nobs = e32.rollout(P, H, np.zeros((H, nu)), eps.astype(np.float32).astype(np.float64))
    _, o_rew, _, _, o_nobs = ref.rollout(st["qp"], st["qv"], st["target_pos"], np.zeros((H, nu)), eps.astype(np.float32).astype(np.float64))
    err = np.abs(rew - o_rew) / np.maximum(1.0, np.abs(o_rew))
    print("%s f32: cost error median %.2e max %.2e, failures %d" % (name, np.median(err), err.max(), e32.solver_failures()))
    # (elliptic cones: the cost is not piecewise quadratic, the f32 iteration stops at steps of 1e-6 of the acceleration - the
    # glass on the tray lands a few 1e-4 of its cost away on the median)
    med, top = (1e-3, 1e-1) if raw.cone == "elliptic" else (1e-4, 5e-2)
    assert np.isfinite(rew).all() and np.median(err) < med and err.max() < top


def test_weld_equality_matches_oracle(tmp_path):
    """<equality><weld>: two free bodies welded to each other and a pendulum welded to the world (tests/
    test_general_models_cpu.py::WELDED): 12-link elimination paths on the dense 16-lane general instantiation, six
    bilateral rows per weld - three at body 2's origin, three on the error quaternion's vector part.  One env step from 32
    random states (relative pose perturbed: the rows pull) at 1e-9, a 64 x 10 rollout at 1e-8."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("general_models_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_general_models_cpu.py"))
    cpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cpu)
    WELDED, WELDS, _model = cpu.WELDED, cpu.WELDS, cpu._model
    raw, ref = _model(tmp_path, WELDED, extra=WELDS, timestep="0.002", frame_skip=2)
    eng = TreeRolloutEngine(raw, dtype="f64")
    assert eng.model.general and eng.model.nv == 13 and eng.model.max_path == 12
    rs = np.random.RandomState(4)
    tgt = np.asarray(raw.target_pos, float)
    worst = 0.0
    for k in range(32):
        q, v = raw.qpos0.copy(), np.zeros(13)
        q[0:3] += 0.05 * rs.standard_normal(3)
        q[3:7] = _quat_mul(q[3:7], _quat(rs, 0.5))
        q[7:10] += 0.05 * rs.standard_normal(3) + 0.004 * rs.standard_normal(3)
        q[10:14] = _quat_mul(q[10:14], _quat(rs, 0.02 if k % 2 else 0.5))
        q[14] = 0.02 * rs.standard_normal()
        v[:] = rs.standard_normal(13) * np.r_[0.3 * np.ones(3), 2 * np.ones(3), 0.3 * np.ones(3), 2 * np.ones(3), 1.0]
        u = rs.uniform(-1, 1, 1)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("weld: one env step from 32 random states, worst relative error %.2e" % worst)
    assert worst < 1e-9, worst
    P, H = 64, 10
    q, v = raw.qpos0.copy(), np.zeros(13)
    v[0:6] = [0.3, 0.0, 0.5, 2.0, -1.0, 1.5]
    eps = 0.5 * rs.standard_normal((P, H, 1))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, 1)), eps, "open_loop")
    o_obs, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 1)), eps)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-7)
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0


def test_velocity_and_general_actuators_match_oracle(tmp_path):
    """<velocity kv>, <general gainprm biasprm biastype="affine"> with all three bias terms, an unclamped control
    (ctrllimited="false") and a force limit (forcerange, reached by a third of the random states) on a two-link arm under gravity: the kernel carries the bias as a stiffness, a velocity term
    and a constant torque at the joint (T_KPG / T_KVG / T_TAU0).  One env step from 32 random states at 1e-9, a 64 x 12
    rollout at 1e-9."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("general_models_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_general_models_cpu.py"))
    cpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cpu)
    acts = ('<actuator><velocity joint="j1" kv="4" gear="2" ctrlrange="-3 3" ctrllimited="true"/>'
            '<general joint="j2" gainprm="6" biastype="affine" biasprm="0.3 -5 -0.2" gear="1.5" ctrlrange="-1 1" ctrllimited="false" '
            'forcelimited="true" forcerange="-4 6"/></actuator>')
    raw, ref = cpu._model(tmp_path, cpu.ARM2, extra=acts, timestep="0.004", frame_skip=2)
    eng = TreeRolloutEngine(raw, dtype="f64")
    assert eng.model.field("kvg")[0] == 16.0 and eng.model.field("tau0")[1] == 1.5 * 0.3 and not eng.model.general
    rs = np.random.RandomState(11)
    tgt = np.asarray(raw.target_pos, float)
    worst = 0.0
    for _ in range(32):
        q, v, u = rs.uniform(-2, 2, 2), 3 * rs.standard_normal(2), rs.uniform(-4, 4, 2)      # (controls beyond both ranges)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("actuators: one env step from 32 random states, worst relative error %.2e" % worst)
    assert worst < 1e-9, worst
    P, H = 64, 12
    q, v = np.array([0.3, -0.5]), np.zeros(2)
    eps = 1.5 * rs.standard_normal((P, H, 2))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, 2)), eps, "open_loop")
    o_obs, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 2)), eps)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    assert np.array_equal(eng.action_lows, [-3.0, -1.0]) and np.array_equal(eng.action_highs, [3.0, 1.0])


def test_per_element_solver_parameters_match_oracle(tmp_path):
    """A model whose floor, ball and two pendulums each carry their own solref / solimp (contact sets mixed by solmix,
    joint-limit and friction-loss sets per joint - tests/test_general_models_cpu.py::MIXED): the kernel reads every row's
    set from the block's table (T_SOLTAB).  One env step from 48 random states at 1e-9, a 64 x 12 rollout at 1e-9."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("general_models_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_general_models_cpu.py"))
    cpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cpu)
    raw, ref = cpu._model(tmp_path, cpu.MIXED % ("0.01 1", "0.02 1", "0.05 1", "0.08 1"), extra=cpu.MIXED_ACT, timestep="0.002", frame_skip=2)
    eng = TreeRolloutEngine(raw, dtype="f64")
    assert eng.model.general and eng.model.nv == 8
    rs = np.random.RandomState(21)
    tgt = np.asarray(raw.target_pos, float)
    worst = 0.0
    for k in range(48):
        q, v = raw.qpos0.copy(), np.zeros(8)
        q[0:2] += 0.2 * rs.standard_normal(2)
        q[2] = 0.1 + rs.uniform(-0.004, 0.01)                       # pressed into / just above the floor
        q[3:7] = _quat(rs, 1.0)
        q[7:9] = rs.uniform(-0.4, 0.4, 2)                           # beyond the limits now and then
        v[:] = rs.standard_normal(8) * np.r_[0.5 * np.ones(3), 3 * np.ones(3), 2 * np.ones(2)] * (k % 4 > 0)
        u = rs.uniform(-1, 1, 2)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("per-element solver sets: one env step from 48 random states, worst relative error %.2e" % worst)
    assert worst < 1e-9, worst
    P, H = 64, 12
    q, v = raw.qpos0.copy(), np.zeros(8)
    q[2] = 0.12
    v[0], v[4], v[6], v[7] = 0.3, 2.0, 3.0, -3.0
    eps = 0.5 * rs.standard_normal((P, H, 2))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, 2)), eps, "open_loop")
    o_obs, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 2)), eps)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


def test_closed_loop_linear_mode_on_general_models(rig):
    """``rollout(mode="closed_loop_linear")`` (gym_env_wrapper.py:135-136) on models with quaternion coordinates: the
    observation the weights multiply is [qpos (MuJoCo's layout, nq entries), qvel, site, site - target]."""
    name, raw, eng, ref = rig
    rs = np.random.RandomState(31)
    A, dobs = eng.d_action, eng.d_obs
    assert dobs == raw.nq + raw.nv + 6
    P, H = 33, 5
    W = 0.1 * rs.standard_normal((dobs + 1, A))
    noise = 0.2 * rs.standard_normal((P, H, A))
    q, v = random_state(name, raw, rs, big=0.3)
    tgt = np.asarray(raw.target_pos, float)
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, W, noise, "closed_loop_linear")
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(q, v, tgt, W, noise, mode="closed_loop_linear")
    np.testing.assert_allclose(act, o_act, rtol=0, atol=1e-9)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-7)
    assert np.abs(act - noise).max() > 1e-2


def test_actuators_on_fixed_tendons_match_oracle(tmp_path):
    """A position servo and a force-limited general actuator, each pulling on a fixed tendon (over two joints / over one):
    tendon length and velocity across the two lanes, each dof takes its coefficient's share of the tendon force.  One env step
    from 32 random states at 1e-9, a 64 x 12 rollout at 1e-9."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("general_models_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_general_models_cpu.py"))
    cpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cpu)
    body = cpu.ARM2.replace('<site name="finger" pos="0.2 0 0"/></body></body>',
                            '<site name="finger" pos="0.2 0 0"/><body name="c" pos="0.2 0 0"><joint name="j3" type="hinge" axis="0 0 1" damping="0.05"/>'
                            '<geom type="capsule" fromto="0 0 0 0.1 0 0" size="0.015"/></body></body></body>')
    extra = ('<tendon><fixed name="t12"><joint joint="j1" coef="1.0"/><joint joint="j2" coef="-0.5"/></fixed>'
             '<fixed name="t3" limited="true" range="-0.8 0.8"><joint joint="j3" coef="2.0"/></fixed></tendon>'
             '<actuator><position tendon="t12" kp="15" gear="1.5" ctrlrange="-1 1" ctrllimited="true"/>'
             '<general tendon="t3" gainprm="3" biastype="affine" biasprm="0.1 -2 -0.1" gear="0.5" ctrlrange="-2 2" ctrllimited="true" '
             'forcelimited="true" forcerange="-1.5 1.5"/></actuator>')
    raw, ref = cpu._model(tmp_path, body, extra=extra, timestep="0.004", frame_skip=2)
    eng = TreeRolloutEngine(raw, dtype="f64")
    m = eng.model
    assert list(m.field("act")[:3]) == [0, 0, 1] and list(m.field("tpartner")[:3]) == [1, 0, -1] and m.field("tcoef")[2] == 2.0
    rs = np.random.RandomState(13)
    tgt = np.asarray(raw.target_pos, float)
    worst = 0.0
    for _ in range(32):
        q, v, u = rs.uniform(-1.5, 1.5, 3), 3 * rs.standard_normal(3), rs.uniform(-2.5, 2.5, 2)
        q[2] = rs.uniform(-0.5, 0.5)                    # (the limited tendon: 2 q3 in [-0.8, 0.8], crossed now and then)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("tendon actuators: one env step from 32 random states, worst relative error %.2e" % worst)
    assert worst < 1e-9, worst
    P, H = 64, 12
    q, v = np.array([0.3, -0.5, 0.1]), np.zeros(3)
    eps = 1.0 * rs.standard_normal((P, H, 2))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, 2)), eps, "open_loop")
    o_obs, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 2)), eps)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)


def test_ball_joint_limit_matches_oracle(tmp_path):
    """A limited ball joint (mj_instantiateLimit, mjJNT_BALL): one soft row over the joint's three dofs, J = -axis of the
    joint quaternion's rotation, evaluated by the record's lane from the quaternion its joint's first link holds.  One env
    step from 48 random orientations (inside and beyond the cone) at 1e-9, a 64 x 12 rollout at 1e-9."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("general_models_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_general_models_cpu.py"))
    cpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cpu)
    body = """
    <body name="bob" pos="0 0 1" quat="0.9 0.1 -0.3 0.2"><joint name="bj" type="ball" damping="0.02" limited="true" range="0 0.6" solreflimit="0.015 1"/>
      <geom type="capsule" fromto="0 0 0 0.3 0 -0.1" size="0.03" mass="0.5"/><site name="finger" pos="0.3 0 -0.1"/>
      <body name="tip" pos="0.3 0 -0.1"><joint name="h" type="hinge" axis="0 1 0" damping="0.05" limited="true" range="-1 1"/>
        <geom type="capsule" fromto="0 0 0 0.1 0 0" size="0.02" mass="0.1"/></body></body>"""
    act = '<actuator><motor joint="h" gear="0.2" ctrlrange="-1 1" ctrllimited="true"/></actuator>'
    raw, ref = cpu._model(tmp_path, body, extra=act, timestep="0.002", frame_skip=2)
    eng = TreeRolloutEngine(raw, dtype="f64")
    assert eng.model.general and eng.model.nv == 4 and raw.nq == 5
    rs = np.random.RandomState(17)
    tgt = np.asarray(raw.target_pos, float)
    worst, beyond = 0.0, 0
    for k in range(48):
        q, v = raw.qpos0.copy(), np.zeros(4)
        q[0:4] = _quat(rs, 0.5 if k % 2 else 1.2)
        q[4] = rs.uniform(-1.2, 1.2)
        v[:] = rs.standard_normal(4) * [2, 2, 2, 3]
        beyond += 2 * np.arctan2(np.linalg.norm(q[1:4]), q[0]) > 0.6
        u = rs.uniform(-1, 1, 1)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("ball-joint limit: one env step from 48 random states (%d beyond the cone), worst relative error %.2e" % (beyond, worst))
    assert worst < 1e-9 and beyond >= 10, (worst, beyond)
    P, H = 64, 12
    q, v = raw.qpos0.copy(), np.array([1.0, 3.0, -2.0, 0.5])
    eps = 0.5 * rs.standard_normal((P, H, 1))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    obs, rew, act_, done, info, nobs = eng.rollout(P, H, np.zeros((H, 1)), eps, "open_loop")
    o_obs, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 1)), eps)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    assert eng.solver_failures() == 0


# ------------------------------------------------------------------------------------------ round 5: joint margin / ref, geom gap
MARGIN_REF_GAP = """<mujoco><compiler angle="radian" coordinate="local" inertiafromgeom="auto"/>
<option timestep="0.002" gravity="0 0 -9.81" integrator="Euler"/>
<default><geom contype="1" conaffinity="1" condim="3" friction="0.8 0.005 0.0001"/></default>
<worldbody><site name="target" pos="0.2 0 0.2"/>
  <geom name="floor" type="plane" size="2 2 0.1" margin="0.002" gap="0.0005"/>
  <body name="base" pos="0 0 0.45"><joint name="lift" type="slide" axis="0 0 1" limited="true" range="-0.25 0.1" ref="-0.05" margin="0.02" damping="2"/>
    <geom name="hub" type="sphere" size="0.04" contype="0" conaffinity="0"/>
    <body name="upper" pos="0 0 0"><joint name="sh" type="hinge" axis="0 1 0" limited="true" range="-0.6 1.4" ref="0.4" margin="0.08" damping="0.1" stiffness="0.5" springref="0.6"/>
      <geom name="u" type="capsule" fromto="0 0 0 0.18 0 0" size="0.02" margin="0.004" gap="0.0025"/>
      <body name="lower" pos="0.18 0 0"><joint name="el" type="hinge" axis="0 1 0" limited="true" range="-1.2 1.2" damping="0.05"/>
        <geom name="l" type="capsule" fromto="0 0 0 0.16 0 0" size="0.018" margin="0.003"/>
        <geom name="tip" type="sphere" pos="0.16 0 0" size="0.025" margin="0.006" gap="0.004"/><site name="finger" pos="0.16 0 0"/>
      </body></body></body>
</worldbody>
<actuator><position joint="lift" kp="60" ctrlrange="-0.3 0.15" ctrllimited="true"/><motor joint="sh" gear="2" ctrlrange="-1 1" ctrllimited="true"/>
  <general joint="el" gainprm="1.5" biastype="affine" biasprm="0.05 -2 -0.1" ctrlrange="-1 1" ctrllimited="true"/></actuator>
</mujoco>"""


def test_joint_margin_ref_and_geom_gap_match_the_oracle(tmp_path):
    """MJCF joint ``margin`` / ``ref`` and geom ``gap`` (VERDICT r4 next #1d) on a three-joint arm over a floor: limit rows that
    begin inside the range, coordinates measured from a reference pose (range, spring, servo and affine-actuator lengths
    stated on qpos), contacts that enter the solver at margin - gap.  One env step from 64 random states at 1e-9, a 64 x 10
    rollout, the state round trip in qpos."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.mjcf import load_mjcf
    from oracle.physics_ref import RefArm
    (tmp_path / "mrg.xml").write_text(MARGIN_REF_GAP)
    raw = load_mjcf(str(tmp_path / "mrg.xml"), self_collision=False)
    eng = TreeRolloutEngine(raw, dtype="f64")
    ref = RefArm(raw.to_flat())
    assert eng.model.general and np.allclose(raw.qpos0, [-0.05, 0.4, 0.0])
    rs = np.random.RandomState(1)
    tgt = np.asarray(raw.target_pos, float)
    worst, rows = 0.0, 0
    for k in range(64):
        q = raw.qpos0 + rs.uniform(-1, 1, 3) * [0.22, 1.1, 1.3]
        v = rs.standard_normal(3) * [0.5, 3.0, 3.0] * rs.choice([0.0, 1.0])
        u = rs.uniform(-1.2, 1.2, 3) * [0.3, 1.0, 1.0]
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        rows += ref.step(q, v, u)[3][0] > 0
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("joint margin / ref, geom gap: one env step from 64 random states (%d with rows), worst relative error %.2e" % (rows, worst))
    assert worst < 1e-9 and rows > 20
    P, H = 64, 10
    q, v = raw.qpos0 + [0.0, 0.5, -0.9], np.array([0.0, 1.0, -1.0])
    eps = 0.6 * rs.standard_normal((P, H, 3))
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    _, rew, _, _, _, nobs = eng.rollout(P, H, np.zeros((H, 3)), eps, "open_loop")
    _, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 3)), eps)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-8)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-8, atol=1e-8)
    # the device-resident env speaks qpos too
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    eng.step_state(np.array([0.1, 0.5, -0.5]))
    got = eng.get_state_device()
    q1, v1, _, _ = ref.env_step(q, v, np.array([0.1, 0.5, -0.5]), tgt)
    np.testing.assert_allclose(got["qp"], q1, rtol=0, atol=1e-11)
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0


CYL_PAIRS = """<mujoco><compiler angle="radian" coordinate="local" inertiafromgeom="auto"/>
<option timestep="0.002" gravity="0 0 -9.81" integrator="Euler" %s/>
<default><geom contype="0" conaffinity="0" condim="3" friction="0.7 0.005 0.0001" margin="0.003"/></default>
<worldbody><site name="target" pos="0 0 0"/>
  <geom name="floor" type="plane" pos="0 0 0" size="5 5 0.1" conaffinity="1"/>
  <geom name="post" type="cylinder" fromto="0 0 0 0 0 0.25" size="0.12"/>
  <body name="pusher" pos="-0.4 0 0.1"><joint name="push" type="slide" axis="1 0 0" range="-0.1 0.5" limited="true" damping="2"/>
    <geom name="tip" type="capsule" fromto="0 0 0 0.15 0 0" size="0.025" density="800"/></body>
  <body name="puck" pos="0.4 0 0.2"><freejoint name="puck_free"/>
    <geom name="puck" type="cylinder" fromto="0 -0.05 0 0 0.05 0" size="0.07" density="700" contype="1"/></body>
  <body name="ball" pos="0.02 0.01 0.34"><freejoint name="ball_free"/>
    <geom name="ball" type="sphere" size="0.06" density="900" contype="1"/></body>
  <body name="rod" pos="0.4 0.02 0.33"><freejoint name="rod_free"/>
    <geom name="rod" type="capsule" fromto="-0.1 0 0 0.1 0 0" size="0.03" density="900" contype="1"/><site name="finger"/></body>
</worldbody>
<contact><pair geom1="ball" geom2="post"/><pair geom1="puck" geom2="rod"/><pair geom1="tip" geom2="post"/><pair geom1="puck" geom2="ball"/></contact>
<actuator><motor joint="push" gear="20" ctrlrange="-1 1" ctrllimited="true"/></actuator></mujoco>"""


@pytest.mark.parametrize("option", ["", 'cone="elliptic" impratio="3"'])
def test_spheres_and_capsules_against_cylinders_match_the_oracle(tmp_path, option):
    """Round 5: sphere / capsule against cylinder, in both geom orders, against a static and a moving cylinder (a ball on a
    post's cap and beside it, a rod lying on / across / inside a puck, a capsule pushed into a post): one env step from 64
    random states at 1e-9 under pyramidal and elliptic cones, and a 32 x 8 rollout."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.mjcf import load_mjcf
    from oracle.physics_ref import RefArm
    (tmp_path / "cyl.xml").write_text(CYL_PAIRS % option)
    raw = load_mjcf(str(tmp_path / "cyl.xml"), self_collision=False)
    assert sorted(tuple(p) for p in raw.pairs) == [("ball", "post"), ("puck", "ball"), ("puck", "rod"), ("tip", "post")]
    eng = TreeRolloutEngine(raw, dtype="f64")
    ref = RefArm(raw.to_flat())
    rs = np.random.RandomState(4)
    tgt = np.asarray(raw.target_pos, float)
    worst, rows = 0.0, 0
    for k in range(64):
        q, v = raw.qpos0.copy(), np.zeros(raw.nv)
        # puck: on the floor or in the air near the rod / ball, any attitude; ball: on / beside / inside the post or at the puck;
        # rod: around the puck; the pusher anywhere in its range (its tip reaches the post beyond 0.13)
        q[0] = rs.uniform(-0.12, 0.52)
        q[1:4] = [0.4, 0.0, 0.2] + rs.standard_normal(3) * [0.04, 0.04, 0.06]
        q[4:8] = _quat(rs, rs.choice([0.0, 0.3, 2.0]))
        where = rs.randint(3)
        q[8:11] = ([0.0, 0.0, 0.31] if where == 0 else ([0.17, 0.0, 0.12] if where == 1 else q[1:4] + [0.0, 0.0, 0.12])) + rs.standard_normal(3) * 0.03
        q[11:15] = _quat(rs, 1.0)
        q[15:18] = q[1:4] + rs.standard_normal(3) * [0.06, 0.06, 0.08] + [0.0, 0.0, 0.06]
        q[18:22] = _quat(rs, rs.choice([0.1, 2.0]))
        v[:] = rs.standard_normal(raw.nv) * 0.5 * rs.choice([0.0, 1.0])
        u = rs.uniform(-1.2, 1.2, 1)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        rows += ref.step(q, v, np.clip(u, -1, 1))[3][0]
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("cylinder pairs (%s): one env step from 64 random states (%d rows in all), worst relative error %.2e" % (option or "pyramidal", rows, worst))
    assert worst < 1e-9 and rows > 300
    P, H = 32, 8
    eps = 0.8 * rs.standard_normal((P, H, 1))
    q, v = raw.qpos0.copy(), np.zeros(raw.nv)
    eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
    _, rew, _, _, _, nobs = eng.rollout(P, H, np.zeros((H, 1)), eps, "open_loop")
    _, o_rew, _, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, 1)), eps)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-7)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-7, atol=1e-7)
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0


def test_direct_solref_matches_the_oracle(tmp_path):
    """MuJoCo's direct solref format (-stiffness, -damping) on joint limits, on the floor and on a geom (round 5; the kernel sees
    only the K and B the model compiler works out): one env step from 64 random states of the margin / ref / gap model at 1e-9."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.mjcf import load_mjcf
    from oracle.physics_ref import RefArm
    xml = MARGIN_REF_GAP.replace('<default><geom contype="1" conaffinity="1" condim="3" friction="0.8 0.005 0.0001"/></default>',
                                 '<default><geom contype="1" conaffinity="1" condim="3" friction="0.8 0.005 0.0001"/>'
                                 '<joint solreflimit="-900 -40"/></default>')
    xml = xml.replace('<geom name="floor" type="plane" size="2 2 0.1" margin="0.002" gap="0.0005"/>',
                      '<geom name="floor" type="plane" size="2 2 0.1" margin="0.002" gap="0.0005" solref="-20000 -250"/>')
    xml = xml.replace('<geom name="tip" type="sphere" pos="0.16 0 0" size="0.025" margin="0.006" gap="0.004"/>',
                      '<geom name="tip" type="sphere" pos="0.16 0 0" size="0.025" margin="0.006" gap="0.004" solref="-6000 -400"/>')
    assert xml.count("solref") == 3
    (tmp_path / "ds.xml").write_text(xml)
    raw = load_mjcf(str(tmp_path / "ds.xml"), self_collision=False)
    eng = TreeRolloutEngine(raw, dtype="f64")
    ref = RefArm(raw.to_flat())
    rs = np.random.RandomState(6)
    tgt = np.asarray(raw.target_pos, float)
    worst, rows = 0.0, 0
    for k in range(64):
        q = raw.qpos0 + rs.uniform(-1, 1, 3) * [0.22, 1.1, 1.3]
        v = rs.standard_normal(3) * [0.5, 3.0, 3.0] * rs.choice([0.0, 1.0])
        u = rs.uniform(-1.2, 1.2, 3) * [0.3, 1.0, 1.0]
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        rows += ref.step(q, v, u)[3][0] > 0
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("direct solref: one env step from 64 random states (%d with rows), worst relative error %.2e" % (rows, worst))
    assert worst < 1e-9 and rows > 20 and eng.solver_failures() == 0

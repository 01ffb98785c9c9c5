"""GPU parity of the tree kernel on the reference's vendored locomotion models (mjmpc/envs/assets/xml/swimmer.xml,
half_cheetah.xml; restated in mjmpc_amd/models/) against the FP64 C oracle, through the C ABI: slide joints and
floating roots, joint springs, motors on a subset of the joints, the inertia-box fluid model (swimmer), capsule/plane
contacts with pyramidal friction cones (cheetah), the forward-progress reward and its observation layout.

Tolerances.  f64: 1e-9 (SURVEY 8d's gate).  The swimmer holds it over whole rollouts.  The cheetah's contact dynamics
amplify a rounding-level difference by about 1.4x per env step (measured: 2e-15 after one step, 4e-11 after eight,
1e-6 after thirty-two - in f64 and f32 alike, scaled by the unit roundoff), so its 1e-9 comparisons are one-step checks
from many states plus a horizon-8 rollout, and longer horizons are compared at the tolerance that growth implies.
f32 is compared over one env step (abs 2e-3) and through the statistics MPC consumes."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _models():
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.swimmer import swimmer_raw
    return dict(swimmer=swimmer_raw, cheetah=half_cheetah_raw)


@pytest.fixture(scope="module", params=["swimmer", "cheetah"])
def loco(request):
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _models()[request.param]()
    return request.param, raw, TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())


def _case(name, nv, nu, seed, P, H):
    rs = np.random.RandomState(seed)
    q0, v0 = 0.15 * rs.standard_normal(nv), 0.5 * rs.standard_normal(nv)
    if name == "cheetah":
        q0[1] = rs.uniform(-0.12, 0.05)                # from resting on the ground to just above it
    return q0, v0, 0.3 * rs.standard_normal((H, nu)), 0.7 * rs.standard_normal((P, H, nu))


def test_f64_rollout_matches_oracle(loco):
    name, raw, eng, ref = loco
    H = 24 if name == "swimmer" else 8
    q0, v0, mean, noise = _case(name, ref.nv, eng.d_action, 1, 101, H)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    obs, rew, act, done, info, nobs = eng.rollout(101, H, mean, noise)
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(q0, v0, np.zeros(3), mean, noise)
    assert obs.shape == (101, H, eng.d_obs) and eng.d_obs == 2 * ref.nv - raw.obs_skip
    assert np.array_equal(act, o_act)                   # the action as given (unclipped), though the motors clip it
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs[:, 0], np.broadcast_to(np.concatenate([q0[raw.obs_skip:], v0]), obs[:, 0].shape), atol=1e-15)
    assert eng.solver_failures() == 0


def test_f64_single_steps_from_many_states(loco):
    """One env step (4 / 5 substeps) from 24 random states, 32 actions each, at 1e-10."""
    name, raw, eng, ref = loco
    worst = 0.0
    for seed in range(24):
        q0, v0, mean, noise = _case(name, ref.nv, eng.d_action, 100 + seed, 32, 1)
        eng.set_env_state(dict(qpos=q0, qvel=v0))
        obs, rew, act, done, info, nobs = eng.rollout(32, 1, mean, 3.0 * noise)      # well past the control limits
        o = ref.rollout(q0, v0, np.zeros(3), mean, 3.0 * noise)
        worst = max(worst, np.abs(nobs - o[4]).max(), np.abs(rew - o[1]).max())
    assert worst < 1e-10 and eng.solver_failures() == 0
    if name == "cheetah":
        assert ref.newton_stats()["iters"] > 0          # the states did touch the ground


def test_cheetah_long_horizon_error_growth():
    """Horizon 32: the f64 kernel stays within the rounding-amplification envelope of the oracle (median 1e-9, max 1e-4)."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _models()["cheetah"]()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    q0, v0, mean, noise = _case("cheetah", 9, 6, 7, 256, 32)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    obs, rew, act, done, info, nobs = eng.rollout(256, 32, mean, noise)
    o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
    e = np.abs(nobs - o[4]).max(axis=2)
    assert np.median(e[:, -1]) < 1e-9 and e.max() < 1e-4 and eng.solver_failures() == 0


@pytest.mark.parametrize("name", ["swimmer", "cheetah"])
def test_f32_one_step_and_rollout_statistics(name):
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _models()[name]()
    eng, ref = TreeRolloutEngine(raw, dtype="f32"), RefArm(raw.to_flat())
    q0, v0, mean, noise = _case(name, ref.nv, eng.d_action, 3, 256, 1)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    obs, rew, act, done, info, nobs = eng.rollout(256, 1, mean, noise)
    o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
    assert np.abs(nobs - o[4]).max() < 2e-3 and np.abs(rew - o[1]).max() < 2e-3        # measured 1e-4 / 2e-4
    # what MPC consumes: the cost-to-go of every particle over a horizon; particles diverge one by one on the cheetah
    # (chaotic contacts), their distribution does not
    H = 16
    q0, v0, mean, noise = _case(name, ref.nv, eng.d_action, 4, 1024, H)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    _, rew, _, _, _, _ = eng.rollout(1024, H, mean, noise)
    o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
    ret, oret = rew.sum(axis=1), o[1].sum(axis=1)
    assert abs(ret.mean() - oret.mean()) < 0.02 * oret.std() + 1e-3
    assert abs(ret.std() - oret.std()) < 0.05 * oret.std()
    assert np.corrcoef(ret, oret)[0, 1] > (0.999 if name == "swimmer" else 0.97)
    assert eng.solver_failures() <= 2


@pytest.mark.parametrize("name", ["swimmer", "cheetah"])
def test_f32_launches_above_4096_particles_run_16_lanes_per_particle(name):
    """f32 launches of more than 4096 particles take the 16-lanes-per-particle instantiation (smaller ones keep 32 lanes):
    6001 particles (a ragged last wavefront), two env steps, every particle against the oracle."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw = _models()[name]()
    eng, ref = TreeRolloutEngine(raw, dtype="f32"), RefArm(raw.to_flat())
    q0, v0, mean, noise = _case(name, ref.nv, eng.d_action, 9, 6001, 2)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    obs, rew, act, done, info, nobs = eng.rollout(6001, 2, mean, noise)
    o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
    e = np.abs(nobs - o[4]).max(axis=(1, 2))
    assert np.median(e) < 2e-4 and np.quantile(e, 0.99) < 2e-3 and e.max() < 5e-2       # (contacts: a few particles switch a substep apart)
    assert np.abs(rew - o[1]).max() < 5e-2 and eng.solver_failures() <= 2


def test_env_classes_step_like_the_oracle():
    """SwimmerEnv / HalfCheetahEnv (the reference's env classes on the tree engine): step, observation, state round trip."""
    from mjmpc_amd.envs.locomotion_env import HalfCheetahEnv, SwimmerEnv
    from oracle.physics_ref import RefArm
    for cls, key in ((SwimmerEnv, "reward_fwd"), (HalfCheetahEnv, "reward_run")):
        env = cls()
        ref = RefArm(env.raw.to_flat())
        ob = env.reset(seed=5)
        st = env.get_env_state()
        assert set(st) == {"qpos", "qvel"} and ob.shape == (env.d_obs,)
        q, v = st["qpos"].copy(), st["qvel"].copy()
        rs = np.random.RandomState(0)
        for _ in range(5):
            a = rs.uniform(-1.5, 1.5, env.d_action)
            ob, r, done, info = env.step(a)
            q, v, ro, oo = ref.env_step(q, v, a, np.zeros(3))
            np.testing.assert_allclose(ob, oo, atol=1e-9)
            assert abs(r - ro) < 1e-9 and done is False
            assert abs(info[key] + info["reward_ctrl"] - r) < 1e-9
        env.set_env_state(st)
        np.testing.assert_array_equal(env.get_env_state()["qpos"], st["qpos"])


def test_mppi_makes_the_cheetah_run():
    """Closed loop through the unchanged controller classes: MPPI over the tree engine moves the cheetah forward,
    a zero policy does not."""
    from mjmpc_amd.control.mppi import MPPI
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    from mjmpc_amd.envs.locomotion_env import HalfCheetahEnv
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    env = HalfCheetahEnv(dtype="f32")
    sim = TreeRolloutEngine(env.raw, dtype="f32")
    ctrl = MPPI(d_state=env.d_state, d_obs=env.d_obs, d_action=env.d_action, action_lows=env.action_lows,
                action_highs=env.action_highs, horizon=16, init_cov=0.3, base_action="null", lam=0.2, num_particles=1024,
                step_size=1.0, alpha=1, gamma=1.0, n_iters=1, filter_coeffs=[0.25, 0.8, 0.0], seed=0,
                noise_mode="device", noise_dtype="f32")
    ctrl.set_sim_state_fn = sim.set_env_state
    ctrl.rollout_fn = make_device_rollout_fn(sim)
    env.reset(seed=0)
    for _ in range(60):
        a, _ = ctrl.optimize(env.get_env_state())
        env.step(a)
    assert env.get_env_state()["qpos"][0] > 1.0         # metres in 3 s of simulated time


@pytest.mark.parametrize("cfg,controller,extra", [
    ("half_cheetah_gpu.yml", "mppi", []), ("swimmer_gpu.yml", "cem", []),
    ("half_cheetah_gpu.yml", "mppi", ["--dyn_randomize_config", os.path.join(ROOT, "examples", "configs", "half_cheetah_gpu_dyn_randomize.yml")])])
def test_example_driver_on_locomotion_configs(tmp_path, cfg, controller, extra):
    with open(os.path.join(ROOT, "examples", "configs", cfg)) as f:
        exp = yaml.safe_load(f)
    exp["n_episodes"], exp["max_ep_length"] = 1, 5
    for block in exp.values():
        if isinstance(block, dict) and "particles_per_cpu" in block:
            block["particles_per_cpu"] = 128
            if extra:
                block["num_cpu"], block["particles_per_cpu"] = 4, 32       # four model shards
    p = tmp_path / "loco.yml"
    p.write_text(yaml.safe_dump(exp))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "example_mpc.py"), "--config", str(p),
                          "--controller", controller, "--noise_mode", "device"] + extra, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "forward progress" in out.stdout and "solver failures 0" in out.stdout
    assert not extra or "randomized params" in out.stdout


def test_bench_line_for_a_tree_workload():
    """`bench.py --workload swimmer`: the bench's JSON contract (roofline incl. the counted-FLOP block, cpu_baseline)
    for a tree-engine model."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "swimmer", "--particles", "256",
                          "--horizon", "8", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["unit"] == "particle-steps/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["roofline"]["kernel"].startswith("tree_rollout_kernel") and d["roofline"]["kernel_ms"] > 0
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["valu"]["flops_per_particle_step"] > 1e4
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["solver_failures"] == 0


def test_dynamics_randomization_per_shard_on_the_tree_engine():
    """``SubprocVecEnv.randomize_dynamics`` on the tree engine (HalfCheetah, 4 shards): every shard simulates its own
    model block - masses, inertias, damping, contact radii / capsule lengths, friction - and agrees with the oracle edited
    through its own setters (run-time-edit semantics: invweight0 kept)."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.compile import principal_inertia
    from oracle.physics_ref import RefArm
    raw = _models()["cheetah"]()
    eng = TreeRolloutEngine(raw, dtype="f64", num_shards=4)
    cfg = {"body_mass": {"torso": [0.3, 0.1], "ffoot": [0.5, 0.0]}, "body_inertia": {"bthigh": [0.3, 0.0]},
           "dof_damping": {"bshin": [0.4, 0.2]}, "geom_size": {"ffoot": [0.2, 0.0], "bfoot": [0.1, 0.0]},
           "geom_friction": {"bfoot": [0.5, 0.5]}, "dof_frictionloss": {"fshin": [0.5, 0.0]},
           "sensor_noise": {"torso_gyro": [0.5, 0.0]}}      # (gym_env_wrapper.py:396-398: a draw, and no effect on the dynamics)
    with pytest.raises(ValueError):
        eng.randomize_dynamics({"sensor_noise": {"torso_gyro": [0.5, 0.0]}}, base_seed=321)       # (no such sensor yet)
    eng = TreeRolloutEngine(raw, dtype="f64", num_shards=4)
    raw.sensors["torso_gyro"] = 0.02
    defaults, rand = eng.randomize_dynamics(cfg, base_seed=321)
    assert 0.01 <= rand[0]["sensor_noise"]["torso_gyro"] <= 0.03 and defaults[0]["sensor_noise"]["torso_gyro"] == 0.02
    assert len(rand) == 4 and rand[0]["body_mass"]["torso"] != rand[1]["body_mass"]["torso"]
    names = [b.name for b in raw.bodies]
    joints = [b.joint.name for b in raw.bodies if b.joint is not None]
    geoms = [g for b in raw.bodies for g in b.geoms if g.collide]          # two contact points each, "to" end first
    P, H = 64, 4
    q0, v0, mean, noise = _case("cheetah", 9, 6, 17, P, H)
    q0[1] = -0.11
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, mean, noise)
    blocks = []
    for i in range(4):
        ref = RefArm(raw.to_flat())
        r = rand[i]
        for n, m in r["body_mass"].items():
            ref.set_body_mass(names.index(n) + 1, m)
        for n, mom in r["body_inertia"].items():
            _, V = principal_inertia(eng.model.body_inertia[names.index(n)])
            ref.set_body_inertia(names.index(n) + 1, V @ np.diag(mom) @ V.T)
        for n, d in r["dof_damping"].items():
            ref.set_dof_damping(joints.index(n), d)
        for n, size in r["geom_size"].items():
            k = [g.name for g in geoms].index(n)
            a, b = np.asarray(geoms[k].a, float), np.asarray(geoms[k].b, float)
            c, u = 0.5 * (a + b), (b - a) / np.linalg.norm(b - a)
            for e, sgn in ((0, 1.0), (1, -1.0)):
                ref.set_sphere_radius(2 * k + e, size[0])
                ref.set_sphere_pos(2 * k + e, c + sgn * size[1] * u)
        for n, fr in r["geom_friction"].items():
            k = [g.name for g in geoms].index(n)
            for e in (0, 1):
                ref.set_sphere_mu(2 * k + e, max(fr[0], raw.plane.friction))
        sl = slice(i * P // 4, (i + 1) * P // 4)
        o = ref.rollout(q0, v0, np.zeros(3), mean, noise[sl])
        np.testing.assert_allclose(rew[sl], o[1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(nobs[sl], o[4], rtol=0, atol=1e-9)
        blocks.append(o[1])
    assert np.abs(blocks[0] - blocks[1]).max() > 1e-3          # the shards really differ
    assert eng.solver_failures() == 0
    with pytest.raises(Exception):
        eng.rollout(62, H, mean, noise[:62])                    # particles must divide into the shards


@pytest.mark.parametrize("name", ["swimmer", "cheetah"])
def test_device_resident_env_step_and_graph_replay(name):
    """``TreeRolloutEngine.step_state`` (the real env kept on the device) equals the env class's host-side step, and an
    MPPI closed loop replayed as a hipGraph (iteration + env step captured) equals the eager one."""
    from mjmpc_amd.control.mppi import MPPI
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    from mjmpc_amd.envs.locomotion_env import HalfCheetahEnv, SwimmerEnv
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    env = dict(swimmer=SwimmerEnv, cheetah=HalfCheetahEnv)[name]()
    env.reset(seed=3)
    eng = TreeRolloutEngine(env.raw, dtype="f64")
    eng.set_env_state(env.get_env_state())
    rs = np.random.RandomState(1)
    for _ in range(6):
        a = rs.uniform(-1.2, 1.2, env.d_action)
        ob, r, _, _ = env.step(a)
        cost, nobs = eng.step_state(a)
        assert abs(float(cost.item()) + r) < 1e-12
        np.testing.assert_allclose(nobs.cpu().numpy(), ob, rtol=0, atol=1e-12)
    st = eng.get_state_device()
    np.testing.assert_allclose(st["qpos"], env.get_env_state()["qpos"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(st["qvel"], env.get_env_state()["qvel"], rtol=0, atol=1e-12)

    def closed_loop(graph):
        e = TreeRolloutEngine(env.raw, dtype="f64")
        c = MPPI(d_state=e.d_state, d_obs=e.d_obs, d_action=e.d_action, action_lows=e.action_lows, action_highs=e.action_highs,
                 horizon=8, init_cov=0.3, base_action="null", lam=0.5, num_particles=256, step_size=1.0, alpha=1, gamma=1.0,
                 n_iters=1, filter_coeffs=[0.25, 0.8, 0.0], seed=5, noise_mode="device", noise_dtype="f64")
        c.rollout_fn = make_device_rollout_fn(e)
        c.set_sim_state_fn = lambda s: None
        e.set_env_state(dict(qpos=0.05 * np.arange(e.model.nv), qvel=np.zeros(e.model.nv)))
        if graph:
            assert c._graph_capable()
            c.enable_graph(post_step=e.step_state)
        acts = []
        for _ in range(8):
            a, _ = c.optimize({"resident": True})
            acts.append(np.array(a))
            if not graph:
                e.step_state(a)
        assert e.solver_failures() == 0
        return np.array(acts), e.get_state_device()

    a_e, s_e = closed_loop(False)
    a_g, s_g = closed_loop(True)
    np.testing.assert_allclose(a_g, a_e, rtol=0, atol=1e-9)
    np.testing.assert_allclose(s_g["qpos"], s_e["qpos"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("lam", [0.2, 10.0])
def test_cheetah_mppi_step_at_the_bench_shape(lam):
    """What MPC consumes on the cheetah at the bench shape: one MPPI step, 4096 x 32, identical
    host noise - ``optimize()`` on the HIP engine (f64) against oracle rollouts + ``mppi_update`` - at the bench's
    lam = 0.2 (the returns of 4096 rollouts span far more than 0.2: the best particle carries 0.996 of the weight) and
    at lam = 10 (a genuinely weighted mean: largest weight < 0.1).
    Tolerance: single costs agree to 1e-10 after one env step and drift apart by ~1.4x per env step afterwards (both sides:
    the contact dynamics amplify rounding), i.e. up to ~1e-6 on a few particles at step 32; a cost-to-go error e moves a
    softmax weight by e / lam relatively, so the weighted mean may move by ~1e-5 |a| at the very most.  Measured (printed):
    2e-14 at lam = 0.2, 1e-14 at lam = 10; asserted 1e-6."""
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import make_rollout_fn
    from mjmpc_amd.envs.locomotion_env import HalfCheetahEnv
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle import controllers_ref as cr
    from oracle.physics_ref import RefArm
    raw = _models()["cheetah"]()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    env = HalfCheetahEnv(dtype="f64")
    env.reset(seed=123)
    st = env.get_env_state()
    P, H, A, cov, filt = 4096, 32, 6, 0.3, [0.25, 0.8, 0.0]
    ctrl = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=A, horizon=H, init_cov=cov, base_action="null", lam=lam,
                num_particles=P, step_size=1.0, alpha=1, gamma=1.0, n_iters=1, action_lows=eng.action_lows,
                action_highs=eng.action_highs, filter_coeffs=filt, seed=123)
    ctrl.set_sim_state_fn = eng.set_env_state
    ctrl.rollout_fn = make_rollout_fn(eng)
    action, _ = ctrl.optimize(st, hotstart=False)
    noise = cr.generate_noise(cov * np.eye(A), filt, (P, H), 123)
    _, rew, act, _, _ = ref.rollout(st["qpos"], st["qvel"], np.zeros(3), np.zeros((H, A)), noise, want_obs=False)
    gseq = cr.gamma_seq(1.0, H)
    mean = cr.mppi_update(-rew, act, np.zeros((H, A)), cov * np.eye(A), gseq, lam, 1, 1.0)
    w = cr.softmax0((-1.0 / lam) * cr.cost_to_go(-rew, gseq)[:, 0])
    err = np.abs(ctrl.mean_action - mean).max()
    print("cheetah 4096x32 lam=%g: max |mean_hip - mean_oracle| = %.3e, |action error| %.3e, largest softmax weight %.3f"
          % (lam, err, np.abs(action - mean[0]).max(), w.max()))
    assert err < 1e-6 and (lam < 1 or w.max() < 0.5)
    assert eng.solver_failures() == 0


def test_swimmer_self_contact_matches_oracle():
    """swimmer.xml's segments collide with each other (capsule-capsule, pyramidal friction): the chain curled into a loop -
    up to three pairs in contact, rows between links of ONE kinematic path - rollouts at 1e-9, f64, and one env step from
    32 curled states; and the launches that never curl are unchanged by the pairs (bit-identical costs)."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.swimmer import swimmer_raw
    from oracle.physics_ref import RefArm
    raw = swimmer_raw()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    rs = np.random.RandomState(5)
    worst, touched = 0.0, 0
    for k in range(32):
        q0 = np.zeros(7)
        q0[:3] = rs.uniform(-1, 1, 3)
        q0[3:] = rs.choice([-1.0, 1.0]) * rs.uniform(1.25, 1.5, 4)
        v0 = 0.5 * rs.standard_normal(7)
        mean, noise = np.zeros((1, 4)), rs.standard_normal((16, 1, 4))
        eng.set_env_state(dict(qpos=q0, qvel=v0))
        obs, rew, act, done, info, nobs = eng.rollout(16, 1, mean, noise)
        before = ref.newton_stats()["iters"]
        o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
        touched += ref.newton_stats()["iters"] > before
        worst = max(worst, np.abs(nobs - o[4]).max(), np.abs(rew - o[1]).max())
    assert touched >= 24 and worst < 1e-9 and eng.solver_failures() == 0, (touched, worst)
    # a rollout that starts curled and is driven further in
    q0 = np.array([0.0, 0.0, 0.3, -1.4, -1.45, -1.4, -1.35])
    mean, noise = -0.5 * np.ones((12, 4)), 0.5 * rs.standard_normal((64, 12, 4))
    eng.set_env_state(dict(qpos=q0, qvel=np.zeros(7)))
    obs, rew, act, done, info, nobs = eng.rollout(64, 12, mean, noise)
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(q0, np.zeros(7), np.zeros(3), mean, noise)
    np.testing.assert_allclose(rew, o_rew, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o_nobs, rtol=0, atol=1e-9)
    # straight-ish postures: the pairs are never in contact and change nothing
    plain = TreeRolloutEngine(swimmer_raw(self_collision=False), dtype="f64")
    q0, v0, mean, noise = _case("swimmer", 7, 4, 3, 64, 16)
    for e in (eng, plain):
        e.set_env_state(dict(qpos=q0, qvel=v0))
    a, b = eng.rollout(64, 16, mean, noise)[1], plain.rollout(64, 16, mean, noise)[1]
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", ["swimmer", "cheetah"])
def test_results_do_not_depend_on_the_batch_size(name, dtype):
    """The instantiation is chosen from model and dtype alone, and (round 6) every decision inside the solver is taken per
    PARTICLE - a converged particle is frozen while the wavefront iterates on for its mates, a re-iteration is a rank-one
    correction or a refactorisation by the particle's own changes, the sine / cosine update is chosen per lane: a particle's
    trajectory is the same BITS with 4095 others, with 8191 others, or rolled out alone (P = 1: the device-resident real
    env).  Until round 5 the last was 2e-4 in f32."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    raw = _models()[name]()
    eng = TreeRolloutEngine(raw, dtype=dtype)
    nv = eng.model.nv
    q0, v0, mean, noise = _case(name, nv, eng.d_action, 9, 8192, 6)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    npdt = np.float32 if dtype == "f32" else np.float64
    got = {}
    for P in (1, 3, 4096, 8192):
        out = eng.rollout_device(P, 6, mean, noise[:P].astype(npdt), want_obs=True)
        got[P] = (out[0].cpu().numpy().copy(), out[3].cpu().numpy().copy())
    for P in (1, 3, 4096):
        assert np.array_equal(got[P][0], got[8192][0][:P]) and np.array_equal(got[P][1], got[8192][1][:P]), P
    # ... and whoever a particle's wave-mates are: the same 64 particles in another order
    perm = np.random.RandomState(5).permutation(64)
    out = eng.rollout_device(64, 6, mean, noise[:64][perm].astype(npdt), want_obs=True)
    assert np.array_equal(out[0].cpu().numpy(), got[8192][0][:64][perm])
    assert np.array_equal(out[3].cpu().numpy(), got[8192][1][:64][perm])


def test_diverged_rollouts_are_counted_apart_from_solver_failures():
    """A particle whose state or acceleration leaves MuJoCo's bounds (a NaN, or an entry beyond mjMAXVAL = 1e10) is reset as
    MuJoCo resets it (mj_checkPos / mj_checkVel / mj_checkAcc -> mj_resetData [EXT]; tests/test_reset_gpu.py holds the
    resulting rollouts to the oracle), counted by diverged_substeps() - mjmpc_tree_diverged - and not by solver_failures();
    its costs stay finite.  (Until round 4 such particles carried a +inf return, which the updates still read as "no
    weight": test_controllers_gpu.py::test_diverged_rollouts_do_not_poison_the_update.)"""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    raw = _models()["cheetah"]()
    eng = TreeRolloutEngine(raw, dtype="f64")
    q0, v0, mean, noise = _case("cheetah", eng.model.nv, eng.d_action, 3, 64, 4)
    eng.set_env_state(dict(qpos=q0, qvel=v0))
    eng.rollout(64, 4, mean, noise)
    assert eng.diverged_substeps() == 0 and eng.solver_failures() == 0
    eng.set_env_state(dict(qpos=q0, qvel=np.full(eng.model.nv, 1e200)))
    rew = eng.rollout(64, 4, mean, noise)[1]
    assert np.isfinite(rew).all()
    assert eng.diverged_substeps() == 64 and eng.solver_failures() == 0      # (one reset per particle: the velocity check of its first substep)


@pytest.mark.parametrize("model,dtype", [("cheetah", "f64"), ("cheetah", "f32"), ("cartpole", "f64")])
def test_tree_rollout_fused_equals_filter_rollout_and_cost_to_go(model, dtype):
    """``mjmpc_tree_rollout_fused`` (round 4): the recursive noise filter (control_utils.py:32-33) and the discounted
    cost-to-go at t = 0 (control_utils.py:37-46) inside the rollout launch = filter pass + plain rollout + numpy."""
    import torch
    from mjmpc_amd.control._device import DeviceUpdater
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    if model == "cheetah":
        from mjmpc_amd.models.half_cheetah import half_cheetah_raw
        raw, start = half_cheetah_raw(), None
    else:
        from mjmpc_amd.models.synthetic import start_state, synthetic_raw
        raw = synthetic_raw(model)
        start = start_state(model, raw)
    eng = TreeRolloutEngine(raw, dtype=dtype)
    if start is not None:
        eng.set_env_state(start)
    P, H, A = 512, 12, eng.d_action
    assert hasattr(eng, "rollout_fused")
    filt, gamma = [0.25, 0.8, 0.1], 0.97
    dev = DeviceUpdater(H, A, gamma ** np.arange(H))
    rs = np.random.RandomState(3)
    mean = torch.from_numpy(0.2 * rs.standard_normal((H, A))).cuda()
    raw_noise = dev.sample_noise(P, 0.3 * np.eye(A), filt, 11, 0, dtype=dtype, filtered=False).clone()
    filtered = dev.sample_noise(P, 0.3 * np.eye(A), filt, 11, 0, dtype=dtype, filtered=True).clone()
    coeffs = torch.tensor(filt, dtype=torch.float64, device="cuda")
    c1, a1, q1 = (x.clone() for x in eng.rollout_fused(P, H, mean, raw_noise, coeffs, dev.gseq))
    c0, a0, _, _ = eng.rollout_device(P, H, mean, filtered)
    torch.cuda.synchronize()
    tol = 1e-12 if dtype == "f64" else 2e-5
    np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=0, atol=tol)
    ctol = 1e-9 if dtype == "f64" else 5e-3
    np.testing.assert_allclose(c1.cpu().numpy(), c0.cpu().numpy(), rtol=ctol, atol=ctol)
    want = (c1.cpu().numpy().astype(np.float64) * (gamma ** np.arange(H))[None]).sum(1)
    np.testing.assert_allclose(q1.cpu().numpy(), want, rtol=1e-12 if dtype == "f64" else 1e-6, atol=1e-12)
    assert eng.solver_failures() == 0


def test_cem_on_a_tree_model_takes_the_fused_step_and_matches_the_eager_loop():
    """With ``rollout_fused`` the tree engine gives CEM its q0 path: a captured iteration is the fused CEM step (A <= 8) and
    walks through the same closed loop as the eager, launch-by-launch controller."""
    import torch
    from mjmpc_amd.control import CEM
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw

    def loop(graph):
        e = TreeRolloutEngine(half_cheetah_raw(), dtype="f64")
        c = CEM(d_state=e.d_state, d_obs=e.d_obs, d_action=e.d_action, action_lows=e.action_lows, action_highs=e.action_highs,
                horizon=8, init_cov=0.3, base_action="null", elite_frac=0.1, beta=0.05, cov_type="full", num_particles=512,
                step_size=0.8, gamma=1.0, n_iters=1, filter_coeffs=[0.25, 0.8, 0.0], seed=5, noise_mode="device",
                noise_dtype="f64")
        c.rollout_fn = make_device_rollout_fn(e)
        c.set_sim_state_fn = lambda s: None
        e.set_env_state(dict(qpos=0.05 * np.arange(e.model.nv), qvel=np.zeros(e.model.nv)))
        if graph:
            c.enable_graph(post_step=e.step_state)
        acts = []
        for _ in range(6):
            a, _ = c.optimize({"resident": True})
            acts.append(np.array(a))
            if not graph:
                e.step_state(a)
        torch.cuda.synchronize()
        assert e.solver_failures() == 0
        if graph:
            assert c._cem_fused() and c.launch_mode.startswith("launch tape")
        return np.array(acts), c.cov_action.copy()

    a_e, c_e = loop(False)
    a_g, c_g = loop(True)
    np.testing.assert_allclose(a_g, a_e, rtol=0, atol=1e-8)
    np.testing.assert_allclose(c_g, c_e, rtol=1e-7, atol=1e-10)

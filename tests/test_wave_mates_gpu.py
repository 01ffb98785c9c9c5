"""A particle's trajectory does not depend on who shares its wavefront (round 6): every
decision inside the tree kernel's solver is taken per particle - a converged particle is frozen while its wave-mates
iterate on, rank-one correction or refactorisation follows the particle's own changes, the sine / cosine update is chosen
per lane.  Checked bitwise on every tree workload, f64 and f32: the same particles alone (P = 1, the device-resident
real env's launch shape), in another order, and among many."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload(name):
    import dataclasses
    if name in ("hand", "handf"):
        from mjmpc_amd.models.hand24 import hand24_raw
        raw = hand24_raw()
        if name == "handf":         # friction cones: the full tree-sparse instantiation with its merged Euler factor
            for b in raw.bodies:
                for g_ in b.geoms:
                    if g_.collide:
                        g_.friction, g_.condim = 0.8, 3
            raw.plane = dataclasses.replace(raw.plane, friction=0.5, condim=3)
        return raw, None, 0.5
    if name == "pen":
        from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
        raw = pen_hand_raw()
        st = holding_state()
        return raw, dict(qpos=st["qp"], qvel=st["qv"], target_pos=np.asarray(raw.target_pos, float)), 0.1
    from mjmpc_amd.models.synthetic import start_state, synthetic_raw
    raw = synthetic_raw(name)
    return raw, start_state(name, raw), 0.05 if name == "gripper" else (0.1 if name == "tray" else 0.5)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", ["hand", "handf", "pen", "cartpole", "tray", "door", "gripper"])
def test_a_particle_does_not_see_its_wave_mates(name, dtype):
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    raw, start, scale = _workload(name)
    eng = TreeRolloutEngine(raw, dtype=dtype)
    if start is not None:
        eng.set_env_state(start)
    A, H, P = eng.d_action, 8, 256
    rs = np.random.RandomState(11)
    npdt = np.float32 if dtype == "f32" else np.float64
    noise = (scale * rs.standard_normal((P, H, A))).astype(npdt)
    mean = np.zeros((H, A))
    if name == "gripper":
        mean[:, 1:] = 0.2
    if name == "pen":
        mean += start["qpos"][6:]

    def run(nz):
        out = eng.rollout_device(nz.shape[0], H, mean, nz, want_obs=True)
        return out[0].cpu().numpy().copy(), out[3].cpu().numpy().copy()

    c_all, o_all = run(noise)
    assert np.isfinite(c_all).all()
    perm = rs.permutation(P)
    c_p, o_p = run(noise[perm])
    assert np.array_equal(c_p, c_all[perm]) and np.array_equal(o_p, o_all[perm])
    for k in (0, 1, 7, 130):            # alone: its wave-mates are the launch's idle lanes
        c_1, o_1 = run(noise[k:k + 1])
        assert np.array_equal(c_1[0], c_all[k]) and np.array_equal(o_1[0], o_all[k]), k


_WM_SEEDS = (range(*[int(x) for x in os.environ["MJMPC_WM_SEEDS"].split(":")])
             if os.environ.get("MJMPC_WM_SEEDS") else range(0, 48, 2))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("seed", _WM_SEEDS)
def test_random_models_particles_do_not_see_their_wave_mates(seed, dtype):
    """The same property over the random-model generator of tests/test_random_models_gpu.py - every instantiation the
    generator reaches (lean / full / general levels 1 - 3, 16 or 32 lanes, tree-sparse or dense, ball and free joints,
    equalities, tendons, friction loss, pyramidal and elliptic cones): 32 particles from a random state, the same particles
    in another order, four of them alone.  (MJMPC_WM_SEEDS=a:b runs another range of seeds.)"""
    from test_random_models_gpu import MAX_REDRAWS, random_model, random_state
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    raw, eng, tries = None, None, 0
    while eng is None:
        raw = random_model(1000 * tries + seed)
        try:
            eng = TreeRolloutEngine(raw, dtype=dtype)
        except (ValueError, NotImplementedError, AssertionError):
            eng, tries = None, tries + 1
            assert tries <= MAX_REDRAWS
    rs = np.random.RandomState(seed + 1234)
    q, v = random_state(raw, rs)
    eng.set_env_state(dict(qp=q, qv=v, target_pos=np.asarray(raw.target_pos, float)))
    A, H, P = eng.d_action, 6, 32
    npdt = np.float32 if dtype == "f32" else np.float64
    noise = rs.uniform(-1.5, 1.5, (P, H, A)).astype(npdt)
    mean = np.zeros((H, A))

    def run(nz):
        out = eng.rollout_device(nz.shape[0], H, mean, nz, want_obs=True)
        return out[0].cpu().numpy().copy(), out[3].cpu().numpy().copy()

    c_all, o_all = run(noise)
    perm = rs.permutation(P)
    c_p, o_p = run(noise[perm])
    # (NaN-safe: a particle that blows up does so identically)
    assert np.array_equal(c_p, c_all[perm], equal_nan=True) and np.array_equal(o_p, o_all[perm], equal_nan=True)
    for k in (0, 3, 17, 31):
        c_1, o_1 = run(noise[k:k + 1])
        assert np.array_equal(c_1[0], c_all[k], equal_nan=True) and np.array_equal(o_1[0], o_all[k], equal_nan=True), k

"""GPU: randomized one-step stress of the tree kernel's full instantiation on the locomotion models - many start states
far from the nominal pose (tumbling, fast, folded), a few actions each, every particle against the FP64 oracle.  One env
step keeps the comparison free of the chaotic amplification longer rollouts show (tests/test_locomotion_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n_states", [("swimmer", 150), ("cheetah", 300)])
def test_one_step_from_wild_states(name, n_states):
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.swimmer import swimmer_raw
    from oracle.physics_ref import RefArm
    raw = dict(swimmer=swimmer_raw, cheetah=half_cheetah_raw)[name]()
    eng, ref = TreeRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    nv, nu = ref.nv, eng.d_action
    rs = np.random.RandomState(2024)
    worst, touched = 0.0, 0
    for k in range(n_states):
        scale = [0.1, 0.5, 1.5][k % 3]
        q0 = scale * rs.standard_normal(nv)
        v0 = 4.0 * scale * rs.standard_normal(nv)
        if name == "cheetah":
            q0[1] = rs.uniform(-0.6, 0.3)               # from deep in the ground (every capsule in contact) to airborne
            q0[2] = rs.uniform(-np.pi, np.pi)           # any pitch: on its feet, its head, its back
        mean, noise = rs.uniform(-1, 1, (1, nu)), 2.0 * rs.standard_normal((8, 1, nu))
        eng.set_env_state(dict(qpos=q0, qvel=v0))
        obs, rew, act, done, info, nobs = eng.rollout(8, 1, mean, noise)
        before = ref.newton_stats()["iters"]
        o = ref.rollout(q0, v0, np.zeros(3), mean, noise)
        touched += ref.newton_stats()["iters"] > before
        sc = 1.0 + np.abs(o[4]).max()
        worst = max(worst, np.abs(nobs - o[4]).max() / sc, np.abs(rew - o[1]).max() / (1.0 + np.abs(o[1]).max()))
    assert worst < 1e-9, worst                           # relative to the size of the state (velocities reach 1e2)
    assert eng.solver_failures() == 0 and ref.newton_stats()["fails"] == 0
    assert touched > (n_states // 2 if name == "cheetah" else n_states // 5)    # states with active constraint rows

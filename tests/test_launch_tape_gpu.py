"""GPU: a captured control iteration replayed as the list of its library calls (``Controller._LaunchTape``,
``enable_graph(tape=True)``, the default on one GPU) walks through exactly the closed loop of the hipGraph replay of the same
capture - same launches, same arguments, same stream order: bit for bit - and the controller only trusts a tape whose own
capture has as many kernel nodes as the captured iteration (``mjmpc_graph_kernel_nodes``)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
START = dict(qp=np.array([0.1, 0.2, 0.0, -0.5, 0.0, -0.3, 0.0]), qv=np.zeros(7), target_pos=np.array([0.2, -0.1, 0.2]))


def _arm_loop(kind, tape, steps=6):
    import torch
    from mjmpc_amd.control import CEM, DMDMPC, MPPI, RandomShooting
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    kw = dict(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=12, num_particles=1024, n_iters=1,
              action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0], seed=9,
              noise_mode="device", gamma=0.99, base_action="null")
    if kind == "cem":
        c = CEM(init_cov=0.6, elite_frac=0.1, step_size=0.7, beta=0.05, cov_type="full", **kw)
    elif kind == "cem_separate":
        c = CEM(init_cov=0.6, elite_frac=0.1, step_size=0.7, beta=0.05, cov_type="diagonal", **kw)
        c._want_cem_fused = False
    elif kind == "dmd_cov":
        c = DMDMPC(init_cov=0.6, beta=0.05, lam=0.2, step_size=0.7, update_cov=True, cov_type="full", **kw)
    elif kind == "rs":
        c = RandomShooting(init_cov=0.6, step_size=0.7, **kw)
    else:
        c = MPPI(init_cov=0.6, lam=0.1, step_size=0.8, alpha=1, **kw)
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    eng.set_env_state(START)
    c.enable_graph(post_step=eng.step_state, mono=False, tape=tape)
    acts = np.array([c.optimize({})[0] for _ in range(steps)])
    torch.cuda.synchronize()
    assert not getattr(c, "graph_fallback", False) and eng.solver_failures() == 0
    return acts, c.mean_action.copy(), c.cov_action.copy(), c.launch_mode


@pytest.mark.parametrize("kind", ["mppi", "cem", "cem_separate", "dmd_cov", "rs"])
def test_tape_equals_graph_replay_on_the_arm(kind):
    a_t, m_t, c_t, mode_t = _arm_loop(kind, True)
    a_g, m_g, c_g, mode_g = _arm_loop(kind, False)
    assert mode_t.startswith("launch tape") and mode_g == "hipGraph replay", (mode_t, mode_g)
    np.testing.assert_array_equal(a_t, a_g)
    np.testing.assert_array_equal(m_t, m_g)
    np.testing.assert_array_equal(c_t, c_g)


def test_tape_equals_graph_replay_on_a_tree_model():
    import torch
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.models.synthetic import synthetic_raw
    from mjmpc_amd.envs.arm_engine import make_device_rollout_fn
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine

    def loop(tape):
        raw = synthetic_raw("cartpole")
        e = TreeRolloutEngine(raw, dtype="f64")
        c = MPPI(d_state=e.d_state, d_obs=e.d_obs, d_action=e.d_action, action_lows=e.action_lows, action_highs=e.action_highs,
                 horizon=8, init_cov=0.3, base_action="null", lam=0.5, num_particles=256, step_size=1.0, alpha=1, gamma=1.0,
                 n_iters=1, filter_coeffs=[0.25, 0.8, 0.0], seed=5, noise_mode="device", noise_dtype="f64")
        c.rollout_fn = make_device_rollout_fn(e)
        c.set_sim_state_fn = lambda s: None
        from mjmpc_amd.models.synthetic import start_state
        e.set_env_state(start_state("cartpole", raw))
        c.enable_graph(post_step=e.step_state, tape=tape)
        acts = np.array([np.array(c.optimize({"resident": True})[0]) for _ in range(8)])
        torch.cuda.synchronize()
        assert e.solver_failures() == 0
        return acts, e.get_state_device(), c.launch_mode

    a_t, s_t, mode_t = loop(True)
    a_g, s_g, mode_g = loop(False)
    assert mode_t.startswith("launch tape") and mode_g == "hipGraph replay", (mode_t, mode_g)
    np.testing.assert_array_equal(a_t, a_g)
    for k in ("qpos", "qp"):
        if k in s_t:
            np.testing.assert_array_equal(s_t[k], s_g[k])


def test_a_tape_that_misses_a_launch_is_refused():
    """The guard: drop one recorded call and the tape's own capture has fewer kernel nodes than the iteration's - the
    controller keeps the hipGraph."""
    import torch
    from mjmpc_amd.control import controller as ctl
    from mjmpc_amd.control import MPPI
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    c = MPPI(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=12, num_particles=512, n_iters=1, init_cov=0.6, lam=0.1,
             step_size=0.8, alpha=1, action_lows=eng.action_lows, action_highs=eng.action_highs, filter_coeffs=[0.25, 0.8, 0.0],
             seed=9, noise_mode="device", gamma=0.99, base_action="null")
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    eng.set_env_state(START)
    c.enable_graph(post_step=eng.step_state, mono=False)
    real = ctl._LaunchTape

    class Lossy(real):
        def __init__(self, calls, *a):
            super().__init__(calls, *a)
            self.calls = self.calls[:-1]

    ctl._LaunchTape = Lossy
    try:
        a0, _ = c.optimize({})
    finally:
        ctl._LaunchTape = real
    torch.cuda.synchronize()
    assert c.launch_mode == "hipGraph replay"
    assert np.all(np.isfinite(a0))


def test_tape_is_recorded_again_after_a_reset():
    """``reset()`` drops the captured iteration; the next ``optimize()`` captures and records afresh - the episode after the
    reset equals the first one (same seed, same start state), under the tape as under the hipGraph."""
    import torch
    from mjmpc_amd.control import CEM
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    c = CEM(d_state=eng.d_state, d_obs=eng.d_obs, d_action=7, horizon=12, num_particles=1024, n_iters=1, init_cov=0.6,
            elite_frac=0.1, step_size=0.7, beta=0.05, cov_type="full", action_lows=eng.action_lows, action_highs=eng.action_highs,
            filter_coeffs=[0.25, 0.8, 0.0], seed=9, noise_mode="device", gamma=0.99, base_action="null")
    c.rollout_fn = make_device_rollout_fn(eng)
    c.set_sim_state_fn = lambda s: None
    c.enable_graph(post_step=eng.step_state)
    episodes = []
    for _ in range(2):
        c.reset()
        eng.set_env_state(START)
        episodes.append(np.array([c.optimize({})[0] for _ in range(4)]))
        torch.cuda.synchronize()
        assert c.launch_mode.startswith("launch tape")
    np.testing.assert_array_equal(episodes[0], episodes[1])


def test_graph_signature_tells_launch_shapes_apart():
    """``mjmpc_graph_signature``: two captures of the same rollout have the same signature, a capture with another particle
    count (as many kernel nodes, another grid) a different one - what the tape's acceptance compares besides node counts."""
    import ctypes
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    if not hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph"):
        pytest.skip("this torch has no raw graph access")
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype="f64")
    eng.set_env_state(START)
    lib = _lib.load()

    def signature(P):
        mean = torch.zeros(8, 7, dtype=torch.float64, device="cuda")
        noise = torch.zeros(P, 8, 7, dtype=torch.float64, device="cuda")
        eng.rollout_device(P, 8, mean, noise)               # (buffers exist before the capture)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            eng.rollout_device(P, 8, mean, noise)
        n, s = (ctypes.c_int64 * 2)(), (ctypes.c_uint64 * 2)()
        _lib.check(lib.mjmpc_graph_kernel_nodes(ctypes.c_void_p(int(g.raw_cuda_graph())), n))
        _lib.check(lib.mjmpc_graph_signature(ctypes.c_void_p(int(g.raw_cuda_graph())), s))
        return (n[0], n[1]), (s[0], s[1])

    n_a, s_a = signature(1024)
    n_b, s_b = signature(1024)
    n_c, s_c = signature(4096)
    assert n_a == n_b == n_c and n_a[0] >= 1
    assert s_a == s_b and s_c[0] != s_a[0] and s_c[1] == s_a[1]

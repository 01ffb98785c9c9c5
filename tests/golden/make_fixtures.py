#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where the reference tree is mounted read-only at
/root/reference.  It imports the reference's own modules by path (a ~20-line in-memory ``gym``
stub replaces the uninstalled dependency; ``mjmpc/__init__.py`` is never executed because it
imports mujoco_py) and records inputs + outputs as small .npz files.  No reference source is
copied: the fixtures are data only.  On the GPU box this script is a no-op (no reference there).

    python tests/golden/make_fixtures.py

Pinned here:
  noise.npz        control_utils.generate_noise            (mjmpc/utils/control_utils.py:24-34)
  cost_to_go.npz   control_utils.cost_to_go                (mjmpc/utils/control_utils.py:37-46)
  update_*.npz     _update_distribution / _shift / _calc_val of MPPI, CEM, DMDMPC,
                   RandomShooting, PFMPC                   (mjmpc/control/*.py)
  closed_loop.npz  GymEnvWrapper.rollout(mode='closed_loop_linear') over PendulumEnv / LQREnv
  mppiq.npz        MPPIQ.calculate_returns / _update_distribution / _calc_val   (mjmpc/control/mppiq.py)
  e2e_*.npz        Controller.optimize() x k steps through the real GymEnvWrapper.rollout over
                   PendulumEnv / LQREnv                    (mjmpc/envs/gym_env_wrapper.py:89-156)
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True      # the reference tree is read-only: importing it must not drop __pycache__ there

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ reference import recipe
def _gym_stub():
    gym = types.ModuleType("gym")

    class Env(object):
        metadata = {}

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is not None:
                low = np.full(shape, low, dtype=np.float64)
                high = np.full(shape, high, dtype=np.float64)
            self.low, self.high = np.asarray(low, np.float64), np.asarray(high, np.float64)
            self.shape = self.low.shape

    class Dict(dict):
        pass

    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Dict = Box, Dict
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")

    def np_random(seed=None):
        return np.random.RandomState(seed), seed

    seeding.np_random = np_random
    utils.seeding = seeding
    gym.Env, gym.spaces, gym.utils = Env, spaces, utils
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils,
                        "gym.utils.seeding": seeding})


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    _gym_stub()
    for pkg in ("mjmpc", "mjmpc.utils", "mjmpc.control", "mjmpc.envs", "mjmpc.envs.basic"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules["mjmpc.utils.helpers"] = types.ModuleType("mjmpc.utils.helpers")
    sys.modules["mjmpc.utils"].helpers = sys.modules["mjmpc.utils.helpers"]
    ns = types.SimpleNamespace()
    ns.control_utils = _load("mjmpc.utils.control_utils", "mjmpc/utils/control_utils.py")
    _load("mjmpc.control.controller", "mjmpc/control/controller.py")
    _load("mjmpc.control.olgaussian_mpc", "mjmpc/control/olgaussian_mpc.py")
    ns.MPPI = _load("mjmpc.control.mppi", "mjmpc/control/mppi.py").MPPI
    ns.CEM = _load("mjmpc.control.cem", "mjmpc/control/cem.py").CEM
    ns.DMDMPC = _load("mjmpc.control.gaussian_dmd", "mjmpc/control/gaussian_dmd.py").DMDMPC
    ns.RandomShooting = _load("mjmpc.control.random_shooting", "mjmpc/control/random_shooting.py").RandomShooting
    ns.PFMPC = _load("mjmpc.control.particle_filter_controller",
                     "mjmpc/control/particle_filter_controller.py").PFMPC
    ns.PendulumEnv = _load("mjmpc.envs.basic.pendulum", "mjmpc/envs/basic/pendulum.py").PendulumEnv
    ns.LQREnv = _load("mjmpc.envs.basic.lqr", "mjmpc/envs/basic/lqr.py").LQREnv
    ns.GymEnvWrapper = _load("mjmpc.envs.gym_env_wrapper", "mjmpc/envs/gym_env_wrapper.py").GymEnvWrapper
    return ns


# ------------------------------------------------------------------ fixtures
def fx_noise(ref):
    out = {}
    cases = [("a", 1.0, [1.0, 0.0, 0.0], (32, 16), 123, 7),
             ("b", 1.0, [0.25, 0.8, 0.0], (32, 16), 124, 7),
             ("c", 3.5, [0.25, 0.8, 0.0], (64, 32), 123, 7),
             ("d", 0.1, [0.5, 0.3, 0.2], (16, 8), 7, 3),
             ("e", 1.0, [0.25, 0.8, 0.0], (1, 1), 123 + 123 * 5, 7)]
    for tag, c, coeffs, shape, seed, A in cases:
        eps = ref.control_utils.generate_noise(c * np.eye(A), coeffs, shape, seed)
        out["%s_cov" % tag] = c * np.eye(A)
        out["%s_coeffs" % tag] = np.array(coeffs)
        out["%s_shape" % tag] = np.array(shape)
        out["%s_seed" % tag] = np.array(seed)
        out["%s_eps" % tag] = eps
    # a general (non-isotropic) covariance: SVD path is LAPACK dependent, stored for tolerance tests
    rs = np.random.RandomState(5)
    B = rs.randn(4, 4)
    cov = B @ B.T + 0.5 * np.eye(4)
    out["g_cov"] = cov
    out["g_coeffs"] = np.array([0.25, 0.8, 0.0])
    out["g_shape"] = np.array((48, 12))
    out["g_seed"] = np.array(99)
    out["g_eps"] = ref.control_utils.generate_noise(cov, [0.25, 0.8, 0.0], (48, 12), 99)
    np.savez_compressed(os.path.join(OUT, "noise.npz"), **out)


def fx_cost_to_go(ref):
    rs = np.random.RandomState(11)
    out = {}
    for tag, gamma, H in (("g1", 1.0, 16), ("g99", 0.99, 32), ("g0", 0.0, 8)):
        costs = rs.rand(24, H) * 5
        gseq = np.cumprod([1.0] + [gamma] * (H - 1)).reshape(1, H)
        out["%s_costs" % tag] = costs
        out["%s_gamma" % tag] = np.array(gamma)
        out["%s_out" % tag] = ref.control_utils.cost_to_go(costs.copy(), gseq)
    np.savez_compressed(os.path.join(OUT, "cost_to_go.npz"), **out)


def _traj(rs, P, H, A, mean):
    delta = rs.randn(P, H, A) * 0.7
    costs = rs.rand(P, H) * 3.0 + 0.2 * np.abs(delta).sum(-1)
    return dict(costs=costs, actions=mean[None] + delta)


def _base(H, A, P):
    return dict(d_state=5, d_obs=6, d_action=A, horizon=H, num_particles=P, n_iters=1,
                action_lows=-np.ones(A), action_highs=np.ones(A), seed=123)


def fx_updates(ref):
    """One _update_distribution + _shift (+ _calc_val) per controller configuration."""
    rs = np.random.RandomState(2024)
    H, A, P = 10, 4, 64
    out = {}

    def record(tag, ctrl, traj, has_cov=True):
        out[tag + "_costs"], out[tag + "_actions"] = traj["costs"], traj["actions"]
        out[tag + "_mean0"] = ctrl.mean_action.copy()
        if has_cov:
            out[tag + "_cov0"] = ctrl.cov_action.copy()
        try:
            out[tag + "_val"] = np.array(ctrl._calc_val(traj))
        except (NotImplementedError, ValueError):
            pass        # reference quirk: MPPI._calc_val breaks with time_based_weights (mppi.py:120)
        ctrl._update_distribution(traj)
        out[tag + "_mean1"] = ctrl.mean_action.copy()
        if has_cov:
            out[tag + "_cov1"] = ctrl.cov_action.copy()
        ctrl.num_steps += 1
        ctrl._shift()
        out[tag + "_mean2"] = ctrl.mean_action.copy()
        if has_cov:
            out[tag + "_cov2"] = ctrl.cov_action.copy()

    def warm(ctrl):
        ctrl.mean_action = rs.randn(H, A) * 0.3
        return ctrl

    i = 0
    for lam in (0.01, 0.2):
        for alpha in (1, 0):
            for tbw in (False, True):
                for gamma, step, base in ((1.0, 1.0, "null"), (0.97, 0.55, "repeat")):
                    c = warm(ref.MPPI(init_cov=1.3, base_action=base, lam=lam, step_size=step, alpha=alpha,
                                      gamma=gamma, time_based_weights=tbw, filter_coeffs=[0.25, 0.8, 0.0],
                                      **_base(H, A, P)))
                    tag = "mppi%d" % i
                    out[tag + "_cfg"] = np.array([lam, alpha, float(tbw), gamma, step, 1.3])
                    out[tag + "_base"] = np.array(base)
                    record(tag, c, _traj(rs, P, H, A, c.mean_action))
                    i += 1
    out["mppi_n"] = np.array(i)

    i = 0
    for cov_type in ("diagonal", "full"):
        for elite, beta, gamma, step in ((0.1, 0.0, 1.0, 1.0), (0.2, 0.3, 0.98, 0.6)):
            c = warm(ref.CEM(init_cov=0.9, base_action="null", elite_frac=elite, step_size=step, gamma=gamma,
                             beta=beta, cov_type=cov_type, filter_coeffs=[1.0, 0.0, 0.0], **_base(H, A, P)))
            tag = "cem%d" % i
            out[tag + "_cfg"] = np.array([elite, beta, gamma, step, 0.9])
            out[tag + "_covtype"] = np.array(cov_type)
            record(tag, c, _traj(rs, P, H, A, c.mean_action))
            i += 1
    out["cem_n"] = np.array(i)

    i = 0
    for update_cov in (False, True):
        for cov_type in ("diagonal", "full"):
            for lam, beta, gamma, step in ((0.2, 0.3, 1.0, 1.0), (0.05, 0.1, 0.95, 0.5)):
                c = warm(ref.DMDMPC(init_cov=1.1, beta=beta, base_action="repeat", lam=lam, step_size=step,
                                    gamma=gamma, update_cov=update_cov, cov_type=cov_type,
                                    filter_coeffs=[1.0, 0.0, 0.0], **_base(H, A, P)))
                tag = "dmd%d" % i
                out[tag + "_cfg"] = np.array([lam, beta, gamma, step, 1.1, float(update_cov)])
                out[tag + "_covtype"] = np.array(cov_type)
                record(tag, c, _traj(rs, P, H, A, c.mean_action))
                i += 1
    out["dmd_n"] = np.array(i)

    i = 0
    for gamma, step in ((1.0, 1.0), (0.9, 0.4)):
        c = warm(ref.RandomShooting(init_cov=0.5, base_action="null", step_size=step, gamma=gamma,
                                    filter_coeffs=[1.0, 0.0, 0.0], **_base(H, A, P)))
        tag = "rs%d" % i
        out[tag + "_cfg"] = np.array([gamma, step, 0.5])
        record(tag, c, _traj(rs, P, H, A, c.mean_action))
        i += 1
    out["rs_n"] = np.array(i)

    # PFMPC: resampling + noisy shift
    i = 0
    for lam, gamma, base in ((0.6, 1.0, "null"), (0.1, 0.97, "repeat")):
        kw = _base(H, A, P)
        c = ref.PFMPC(cov_shift=0.1, cov_resample=1.0, base_action=base, lam=lam, gamma=gamma,
                      filter_coeffs=[0.25, 0.8, 0.0], **kw)
        tag = "pf%d" % i
        out[tag + "_cfg"] = np.array([lam, gamma, 0.1, 1.0])
        out[tag + "_base"] = np.array(base)
        out[tag + "_samples0"] = c.action_samples.copy()
        traj = dict(costs=rs.rand(P, H) * 2.0, actions=c.action_samples.copy())
        out[tag + "_costs"] = traj["costs"]
        c._update_distribution(traj)
        out[tag + "_samples1"] = c.action_samples.copy()
        out[tag + "_mean1"] = c.mean_action.copy()
        c.num_steps += 1
        c._shift()
        out[tag + "_samples2"] = c.action_samples.copy()
        i += 1
    out["pf_n"] = np.array(i)
    np.savez_compressed(os.path.join(OUT, "updates.npz"), **out)


def _e2e(ref, tag, wrapper, state0, make_ctrl, steps, out):
    """optimize() x steps through the reference's own GymEnvWrapper.rollout."""
    real = wrapper
    traj_log = {}

    def rollout_fn(num_particles, horizon, mean, noise, mode):
        obs, rew, act, done, info, nobs = wrapper.rollout(num_particles, horizon, mean.copy(), noise, mode)
        d = dict(observations=obs.copy(), actions=act.copy(), costs=-1.0 * rew.copy(), dones=done.copy(),
                 next_observations=nobs.copy())
        traj_log["last"] = d
        traj_log.setdefault("first", d)
        traj_log["noise_first"] = traj_log.get("noise_first", None if noise is None else noise.copy())
        return d

    ctrl = make_ctrl()
    ctrl.set_sim_state_fn = wrapper.set_env_state
    ctrl.rollout_fn = rollout_fn
    state = state0
    actions, states = [], []
    for _ in range(steps):
        states.append(np.asarray(state["state"], float).reshape(-1).copy())
        a, _ = ctrl.optimize(state, calc_val=False, hotstart=True)
        actions.append(a.copy())
        real.set_env_state(state)
        real.step(a)
        state = {"state": np.array(real.get_env_state()["state"], float).copy()}
    out[tag + "_actions"] = np.array(actions)
    out[tag + "_states"] = np.array(states)
    out[tag + "_final_mean"] = ctrl.mean_action.copy()
    for k in ("observations", "actions", "costs", "dones", "next_observations"):
        out["%s_first_%s" % (tag, k)] = traj_log["first"][k]
    out[tag + "_first_noise"] = traj_log["noise_first"]


def fx_e2e(ref):
    out = {}
    # ---------------- Pendulum (d_obs 3, d_state 2, d_action 1)
    env = ref.PendulumEnv()
    env._max_episode_steps = 200
    w = ref.GymEnvWrapper(env)
    H, P = 10, 48
    kw = dict(d_state=w.d_state, d_obs=w.d_obs, d_action=w.d_action, horizon=H, num_particles=P, n_iters=1,
              action_lows=env.action_space.low, action_highs=env.action_space.high, seed=123)
    s0 = {"state": np.array([2.5, -0.4])}
    _e2e(ref, "pend_mppi", w, s0, lambda: ref.MPPI(init_cov=0.8, base_action="null", lam=0.1, step_size=0.9,
         alpha=1, gamma=0.99, filter_coeffs=[0.25, 0.8, 0.0], **kw), 6, out)
    _e2e(ref, "pend_rs", w, s0, lambda: ref.RandomShooting(init_cov=0.8, base_action="null", step_size=1.0,
         gamma=1.0, filter_coeffs=[1.0, 0.0, 0.0], **kw), 4, out)
    # the three olgaussian_mpc.py branches no other fixture takes:
    #   use_zero_control_seq (:110-111: the last particle's perturbation is -mean, i.e. it rolls out zero controls),
    #   base_action='random' (:122-123: the new last row comes from the GLOBAL numpy stream, which generate_noise
    #   left just behind the (P, H) draw of this step), sample_mode='sample' (:72-75: action = mean[0] + one draw
    #   seeded seed + 123 * num_steps)
    _e2e(ref, "pend_zero", w, s0, lambda: ref.MPPI(init_cov=0.8, base_action="null", lam=0.1, step_size=0.9,
         alpha=1, gamma=0.99, filter_coeffs=[0.25, 0.8, 0.0], use_zero_control_seq=True, **kw), 5, out)
    _e2e(ref, "pend_random", w, s0, lambda: ref.MPPI(init_cov=0.8, base_action="random", lam=0.1, step_size=0.9,
         alpha=1, gamma=0.99, filter_coeffs=[0.25, 0.8, 0.0], **kw), 5, out)
    _e2e(ref, "pend_sample", w, s0, lambda: ref.MPPI(init_cov=0.8, base_action="null", lam=0.1, step_size=0.9,
         alpha=1, gamma=0.99, filter_coeffs=[0.25, 0.8, 0.0], sample_mode="sample", **kw), 5, out)
    # ---------------- LQR  (d_state 3 (column vector), d_action 2)
    rs = np.random.RandomState(3)
    Amat = np.eye(3) + 0.05 * rs.randn(3, 3)
    Bmat = 0.3 * rs.randn(3, 2)
    Q = np.diag([1.0, 0.5, 2.0])
    R = 0.1 * np.eye(2)
    out["lqr_A"], out["lqr_B"], out["lqr_Q"], out["lqr_R"] = Amat, Bmat, Q, R
    class LQRWithGetObs(ref.LQREnv):
        # reference quirk: LQREnv only defines _get_obs (lqr.py:43) while GymEnvWrapper.get_obs
        # calls env.get_obs() (gym_env_wrapper.py:74-81); alias it so the wrapper can run
        def get_obs(self):
            return self._get_obs()

        # reference quirk: reset() builds a (d_state, 1) column state (lqr.py:41) which
        # A.dot(state) + B.dot(u) broadcasts to (d_state, d_state) for the 1-D actions the
        # wrapper sends; keep the state 1-D so step() (lqr.py:31-35, unchanged) is well defined
        def reset(self, seed=None):
            super().reset(seed)
            self.state = self.state.reshape(-1)
            return self.state.copy()

    env = LQRWithGetObs(Amat, Bmat, Q, R)
    env._max_episode_steps = 200
    w = ref.GymEnvWrapper(env)
    kw = dict(d_state=w.d_state, d_obs=w.d_obs, d_action=w.d_action, horizon=8, num_particles=40, n_iters=2,
              action_lows=-np.ones(2) * 5, action_highs=np.ones(2) * 5, seed=77)
    s0 = {"state": np.array([1.0, -2.0, 0.5])}
    _e2e(ref, "lqr_cem", w, s0, lambda: ref.CEM(init_cov=1.0, base_action="null", elite_frac=0.2, step_size=0.8,
         gamma=1.0, beta=0.1, cov_type="full", filter_coeffs=[1.0, 0.0, 0.0], **kw), 4, out)
    _e2e(ref, "lqr_dmd", w, s0, lambda: ref.DMDMPC(init_cov=1.0, beta=0.1, base_action="null", lam=0.5,
         step_size=0.7, gamma=1.0, update_cov=True, cov_type="diagonal", filter_coeffs=[1.0, 0.0, 0.0], **kw),
         4, out)
    # mean-only rollout (noise=None) through the wrapper
    w.set_env_state(s0)
    obs, rew, act, done, info, nobs = w.rollout(1, 8, 0.1 * np.ones((8, 2)), None, "open_loop")
    out["lqr_meanonly_obs"], out["lqr_meanonly_rew"], out["lqr_meanonly_act"] = obs, rew, act
    np.savez_compressed(os.path.join(OUT, "e2e.npz"), **out)


def _lqr_env(ref, Amat, Bmat, Q, R):
    class LQRWithGetObs(ref.LQREnv):       # same two reference quirks as in fx_e2e
        def get_obs(self):
            return self._get_obs()

        def reset(self, seed=None):
            super().reset(seed)
            self.state = self.state.reshape(-1)
            return self.state.copy()

    env = LQRWithGetObs(Amat, Bmat, Q, R)
    env._max_episode_steps = 200
    return env


def fx_closed_loop(ref):
    """mode='closed_loop_linear' (gym_env_wrapper.py:135-136) through the reference's wrapper."""
    out = {}
    rs = np.random.RandomState(41)
    env = ref.PendulumEnv()
    env._max_episode_steps = 200
    w = ref.GymEnvWrapper(env)
    P, H = 12, 9
    W = 0.6 * rs.randn(w.d_obs + 1, 1)
    noise = 0.5 * rs.randn(P, H, 1)
    s0 = {"state": np.array([1.3, 0.7])}
    w.set_env_state(s0)
    obs, rew, act, done, info, nobs = w.rollout(P, H, W.copy(), noise.copy(), "closed_loop_linear")
    out.update(pend_state=s0["state"], pend_W=W, pend_noise=noise, pend_obs=obs, pend_rew=rew, pend_act=act,
               pend_nobs=nobs)
    w.set_env_state(s0)
    obs, rew, act, done, info, nobs = w.rollout(1, H, W.copy(), None, "closed_loop_linear")
    out.update(pend_mean_obs=obs, pend_mean_rew=rew, pend_mean_act=act)

    Amat = np.eye(3) + 0.05 * rs.randn(3, 3)
    Bmat = 0.3 * rs.randn(3, 2)
    Q, R = np.diag([1.0, 0.5, 2.0]), 0.1 * np.eye(2)
    w = ref.GymEnvWrapper(_lqr_env(ref, Amat, Bmat, Q, R))
    W = 0.4 * rs.randn(w.d_obs + 1, 2)
    P, H = 10, 7
    noise = 0.3 * rs.randn(P, H, 2)
    s0 = {"state": np.array([0.5, -1.0, 0.25])}
    w.set_env_state(s0)
    obs, rew, act, done, info, nobs = w.rollout(P, H, W.copy(), noise.copy(), "closed_loop_linear")
    out.update(lqr_A=Amat, lqr_B=Bmat, lqr_Q=Q, lqr_R=R, lqr_state=s0["state"], lqr_W=W, lqr_noise=noise,
               lqr_obs=obs, lqr_rew=rew, lqr_act=act, lqr_nobs=nobs)
    np.savez_compressed(os.path.join(OUT, "closed_loop.npz"), **out)


def fx_mppiq(ref):
    """MPPIQ (mjmpc/control/mppiq.py): TD(lambda) returns over optional terminal Q estimates, softmax update."""
    MPPIQ = _load("mjmpc.control.mppiq", "mjmpc/control/mppiq.py").MPPIQ
    rs = np.random.RandomState(909)
    H, A, P = 9, 3, 48
    out = {}
    i = 0
    for beta, alpha, tbw, gamma, td_lam, step, with_q in (
            (0.3, 1, True, 0.98, 0.9, 1.0, True), (0.3, 1, False, 0.98, 0.9, 0.7, True),
            (0.05, 1, True, 1.0, 1.0, 1.0, False), (0.2, 0, True, 0.95, 0.8, 0.6, True),
            (0.2, 0, False, 1.0, 0.5, 1.0, False), (0.5, 1, True, 0.9, 0.0, 1.0, True)):
        c = MPPIQ(init_cov=1.2, base_action="null", beta=beta, step_size=step, alpha=alpha, gamma=gamma, n_iters=1,
                  td_lam=td_lam, time_based_weights=tbw, filter_coeffs=[1.0, 0.0, 0.0], d_state=5, d_obs=6,
                  d_action=A, horizon=H, num_particles=P, action_lows=-np.ones(A), action_highs=np.ones(A), seed=3)
        c.mean_action = rs.randn(H, A) * 0.3
        traj = _traj(rs, P, H, A, c.mean_action)
        if with_q:
            traj["qvals"] = rs.rand(P, H) * 4.0
        tag = "q%d" % i
        out[tag + "_cfg"] = np.array([beta, alpha, float(tbw), gamma, td_lam, step, 1.2, float(with_q)])
        out[tag + "_costs"], out[tag + "_actions"] = traj["costs"], traj["actions"]
        if with_q:
            out[tag + "_qvals"] = traj["qvals"]
        out[tag + "_mean0"] = c.mean_action.copy()
        delta = traj["actions"] - c.mean_action[None]
        out[tag + "_returns"] = c.calculate_returns(traj["costs"] + beta * c._control_costs(delta),
                                                    traj.get("qvals"), gamma, td_lam)
        out[tag + "_val"] = np.array(c._calc_val(traj))
        c._update_distribution(traj)
        out[tag + "_mean1"] = c.mean_action.copy()
        i += 1
    out["n"] = np.array(i)
    np.savez_compressed(os.path.join(OUT, "mppiq.npz"), **out)


def main():
    if not os.path.isdir(REF):
        print("reference tree not present; fixtures are already committed - nothing to do")
        return 0
    ref = load_reference()
    fx_noise(ref)
    fx_cost_to_go(ref)
    fx_updates(ref)
    fx_e2e(ref)
    fx_closed_loop(ref)
    fx_mppiq(ref)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("%-20s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))
    return 0


if __name__ == "__main__":
    sys.exit(main())

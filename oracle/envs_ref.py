"""numpy restatement of the rollout loop over the two analytic envs  --  TEST ORACLE, NOT PRODUCT.

Pinned against tests/golden/e2e.npz and closed_loop.npz, which were produced by the reference's own
``GymEnvWrapper.rollout`` (mjmpc/envs/gym_env_wrapper.py:89-156) driving ``PendulumEnv``
(mjmpc/envs/basic/pendulum.py:33-50,62-64) and ``LQREnv`` (mjmpc/envs/basic/lqr.py:31-35).
Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
import numpy as np


class PendulumRef:
    d_obs, d_state, d_action = 3, 2, 1
    max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0

    def obs(self, s):
        return np.array([np.cos(s[0]), np.sin(s[0]), s[1]])

    def step(self, s, u):
        th, thdot = s
        u = np.clip(u, -self.max_torque, self.max_torque)[0]
        ang = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = ang ** 2 + .1 * thdot ** 2 + .001 * (u ** 2)
        newthdot = thdot + (-3 * self.g / (2 * self.l) * np.sin(th + np.pi) + 3. / (self.m * self.l ** 2) * u) * self.dt
        newth = th + newthdot * self.dt
        newthdot = np.clip(newthdot, -self.max_speed, self.max_speed)
        return np.array([newth, newthdot]), -cost


class LQRRef:
    def __init__(self, A, B, Q, R):
        self.A, self.B, self.Q, self.R = A, B, Q, R
        self.d_state = self.d_obs = A.shape[0]
        self.d_action = B.shape[1]

    def obs(self, s):
        return s.copy()

    def step(self, s, u):
        cost = s.T.dot(self.Q).dot(s) + u.T.dot(self.R).dot(u)
        return self.A.dot(s) + self.B.dot(u), -cost


def rollout(env, state0, num_particles, horizon, mean, noise, mode="open_loop"):
    """GymEnvWrapper.rollout -> (obs, rew, act, done, next_obs); mode 'closed_loop_linear': ``mean`` is the
    (d_obs+1, d_action) weight matrix and the action is W^T [obs; 1] (+ noise) (gym_env_wrapper.py:133-136)."""
    P, H = num_particles, horizon
    obs = np.zeros((P, H, env.d_obs))
    nobs = np.zeros((P, H, env.d_obs))
    rew = np.zeros((P, H))
    act = np.zeros((P, H, env.d_action))
    done = np.zeros((P, H))
    for b in range(P):
        s = np.asarray(state0, float).copy()
        cur = env.obs(s)
        for t in range(H):
            mean_act = mean[t] if mode == "open_loop" else mean.T @ np.append(cur, 1.0)
            u = mean_act + (noise[b, t] if noise is not None else 0.0)
            s, r = env.step(s, u)
            nxt = env.obs(s)
            obs[b, t], nobs[b, t], rew[b, t], act[b, t] = cur, nxt, r, u
            cur = nxt
    return obs, rew, act, done, nobs

#!/usr/bin/env python
"""Closed-loop MPC episodes on the MI355X engine - the counterpart of mjmpc's examples/example_mpc.py.

    python examples/example_mpc.py --config examples/configs/reacher_gpu.yml --controller mppi
        [--dyn_randomize_config examples/configs/reacher_gpu_dyn_randomize.yml]
        [--noise_mode host|device|device_mt19937] [--dtype f64|f32] [--graph]

Same experiment file format, same loop: per episode a fresh MPCPolicy whose controller gets
``set_sim_state_fn`` / ``rollout_fn`` assigned, ``policy.get_action(state)`` -> ``env.step(action)``.  What changes
is who executes the rollouts: one ``ArmRolloutEngine`` (all particles in one kernel launch, `num_cpu` model shards)
instead of `num_cpu` MuJoCo worker processes.  `--noise_mode host` keeps the reference's random stream.
"""
import argparse
import os
import sys
import time
from copy import deepcopy

import numpy as np
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mjmpc_amd.envs.arm_engine import ArmRolloutEngine, make_device_rollout_fn, make_rollout_fn   # noqa: E402
from mjmpc_amd.envs.locomotion_env import HalfCheetahEnv, SwimmerEnv                             # noqa: E402
from mjmpc_amd.envs.reacher_env import ContinualReacher7DOFEnv, HandTreeEnv, Reacher7DOFEnv      # noqa: E402
from mjmpc_amd.envs.synthetic_env import CartPoleEnv, DoorEnv, GripperEnv, TrayEnv                           # noqa: E402
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine                                         # noqa: E402
from mjmpc_amd.models.half_cheetah import half_cheetah_raw                                       # noqa: E402
from mjmpc_amd.models.hand24 import hand24_raw                                                   # noqa: E402
from mjmpc_amd.models.reacher7dof import reacher7dof_raw                                         # noqa: E402
from mjmpc_amd.models.swimmer import swimmer_raw                                                 # noqa: E402
from mjmpc_amd.models.synthetic import synthetic_raw                                             # noqa: E402
from mjmpc_amd.policies import MPCPolicy                                                         # noqa: E402

# the reference's registered MuJoCo envs whose models are vendored (mjmpc/envs/__init__.py:11-31), plus the synthetic
# 24-dof tree; all but the two reachers run on the tree engine
ENVS = {"reacher_7dof-v0": Reacher7DOFEnv, "continual_reacher-v0": ContinualReacher7DOFEnv,
        "Swimmer-v0": SwimmerEnv, "HalfCheetah-v0": HalfCheetahEnv, "hand_tree-v0": HandTreeEnv,
        # round 4: synthetic MJCF models of the kinds the reference's other experiment files name (general kernel instantiation)
        "cartpole_friction-v0": CartPoleEnv, "tray_glass_synthetic-v0": TrayEnv, "door_latch_synthetic-v0": DoorEnv,
        "pen_gripper_synthetic-v0": GripperEnv}      # (round 5: capsule / box and cylinder / plane contacts, joint ref / margin)
TREE_MODELS = {"hand_tree-v0": hand24_raw, "Swimmer-v0": swimmer_raw, "HalfCheetah-v0": half_cheetah_raw,
               "cartpole_friction-v0": lambda: synthetic_raw("cartpole"), "tray_glass_synthetic-v0": lambda: synthetic_raw("tray"),
               "door_latch_synthetic-v0": lambda: synthetic_raw("door"), "pen_gripper_synthetic-v0": lambda: synthetic_raw("gripper")}


def make_sim(env_name, dtype, num_shards):
    """The rollout engine standing in for the reference's SubprocVecEnv worker pool."""
    if env_name in TREE_MODELS:
        return TreeRolloutEngine(TREE_MODELS[env_name](), dtype=dtype, num_shards=num_shards)
    return ArmRolloutEngine(reacher7dof_raw(), dtype=dtype, num_shards=num_shards)


def main():
    ap = argparse.ArgumentParser(description="Run an MPC algorithm on a given environment")
    ap.add_argument("--config", required=True, help="yaml file with experiment parameters")
    ap.add_argument("--dyn_randomize_config", help="yaml file with dynamics randomization parameters")
    ap.add_argument("--controller", default="mppi", help="controller block of the config to run")
    ap.add_argument("--noise_mode", default="host", choices=["host", "device", "device_mt19937"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--graph", action="store_true", help="replay the control iteration as a hipGraph (device noise modes)")
    ap.add_argument("--episodes", type=int, help="override n_episodes")
    args = ap.parse_args()
    with open(args.config) as f:
        exp = yaml.safe_load(f)
    if exp["env_name"] not in ENVS:
        raise SystemExit("environment %r is not built (have: %s)" % (exp["env_name"], ", ".join(ENVS)))
    if not isinstance(exp.get(args.controller), dict) or args.controller == "shared":
        raise SystemExit("the config has no %r controller block" % args.controller)
    params = dict(exp[args.controller])
    num_cpu = params.pop("num_cpu", 1)
    if "particles_per_cpu" in params:
        params["num_particles"] = num_cpu * params.pop("particles_per_cpu")
    controller_type = "mppi" if args.controller == "bench" else args.controller

    env = ENVS[exp["env_name"]](dtype=args.dtype)                    # the "real" environment
    env.real_env_step(True)
    sim = make_sim(exp["env_name"], args.dtype, num_cpu)             # the rollout engine
    if args.dyn_randomize_config:
        with open(args.dyn_randomize_config) as f:
            default_params, randomized = sim.randomize_dynamics(yaml.safe_load(f), base_seed=exp["seed"])
        print("default params   :", default_params[0])
        print("randomized params:", randomized)

    params.update(d_obs=env.d_obs, d_state=env.d_state, d_action=env.d_action, action_lows=env.action_lows,
                  action_highs=env.action_highs)
    if controller_type != "pfmpc":
        params.setdefault("base_action", exp.get("base_action", "null"))
        params.update(noise_mode=args.noise_mode, noise_dtype=args.dtype)
    else:
        params.setdefault("base_action", exp.get("base_action", "null"))
    n_episodes = args.episodes or exp["n_episodes"]
    ep_rewards = np.zeros(n_episodes)
    trajectories = []
    t_ctrl, n_ctrl = 0.0, 0
    for i in range(n_episodes):
        episode_seed = exp["seed"] + i * 12345                       # consistent episodes, as in the reference
        params["seed"] = episode_seed
        env.reset(seed=episode_seed)
        policy = MPCPolicy(controller_type=controller_type, param_dict=params, batch_size=1)
        ctrl = policy.controller
        ctrl.set_sim_state_fn = sim.set_env_state
        device_path = args.noise_mode != "host" and controller_type != "pfmpc"
        ctrl.rollout_fn = make_device_rollout_fn(sim) if device_path else make_rollout_fn(sim)
        if args.graph and device_path:
            ctrl.enable_graph()
        rewards, infos, actions, observations = [], [], [], []
        for _ in range(exp["max_ep_length"]):
            state = deepcopy(env.get_env_state())
            t0 = time.perf_counter()
            action, _ = policy.get_action(state, calc_val=False)
            t_ctrl += time.perf_counter() - t0
            n_ctrl += 1
            obs, reward, done, info = env.step(action)
            observations.append(obs)
            actions.append(action)
            rewards.append(reward)
            infos.append(info["goal_achieved"])
            ep_rewards[i] += reward
        trajectories.append(dict(observations=np.array(observations), actions=np.array(actions),
                                 rewards=np.array(rewards), env_infos=dict(goal_achieved=np.array(infos))))
        if exp["env_name"] in ("Swimmer-v0", "HalfCheetah-v0"):
            print("episode %d: reward %.3f, forward progress %.3f m" % (i, ep_rewards[i], env.get_env_state()["qpos"][0]))
        else:
            print("episode %d: reward %.3f, final distance to target %.4f (closest %.4f)"
                  % (i, ep_rewards[i], np.linalg.norm(observations[-1][-3:]), min(np.linalg.norm(o[-3:]) for o in observations)))
    failures = sim.solver_failures()
    sim.close()
    print("Avg. reward = %.4f, Std. Reward = %.4f, Success Metric = %.1f" % (
        ep_rewards.mean(), ep_rewards.std(), env.evaluate_success(trajectories)))
    print("%s: %d particles x H%d, %.3f ms per optimize() (%.0f Hz), solver failures %d" % (
        args.controller, params["num_particles"], params["horizon"], 1e3 * t_ctrl / n_ctrl, n_ctrl / t_ctrl,
        failures))


if __name__ == "__main__":
    main()

"""Rollout-kernel time of the tree engine: python tools/tree_time.py [P] [H] [dtype] [hand|handf|swimmer|cheetah|pen|penf|cartpole|tray|door]
(MJMPC_AMD_LIB selects an alternative build of the library, e.g. one compiled with -DTREE_SKIP=...)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
from mjmpc_amd.models.hand24 import hand24_raw
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt = sys.argv[3] if len(sys.argv) > 3 else "f64"
name = sys.argv[4] if len(sys.argv) > 4 else "hand"
start = None
if name in ("hand", "handf"):
    raw = hand24_raw()
    if name == "handf":         # friction cones on the fingertips and the table: the full instantiation (more than 16 dofs)
        import dataclasses
        for b in raw.bodies:
            for g_ in b.geoms:
                if g_.collide:
                    g_.friction, g_.condim = 0.8, 3
        raw.plane = dataclasses.replace(raw.plane, friction=0.5, condim=3)
elif name in ("pen", "penf"):
    from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
    raw = pen_hand_raw()
    if name == "penf":          # dry friction in the finger joints: friction-loss rows -> the GENERAL instantiation at 32 lanes
        for b in raw.bodies:
            if b.joint is not None and (b.name.endswith("_mid") or b.name.endswith("_prox") or b.name == "arm_wrist"):
                b.joint.frictionloss = 0.02
    st = holding_state()
    start = dict(qpos=st["qp"], qvel=st["qv"], target_pos=np.asarray(raw.target_pos, float))
elif name in ("cartpole", "tray", "door", "gripper"):
    from mjmpc_amd.models.synthetic import start_state, synthetic_raw
    raw = synthetic_raw(name)
    start = start_state(name, raw)
else:
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.swimmer import swimmer_raw
    raw = dict(swimmer=swimmer_raw, cheetah=half_cheetah_raw)[name]()
eng = TreeRolloutEngine(raw, dtype=dt)
A = eng.d_action
if start is not None:
    eng.set_env_state(start)
if name == "cheetah":       # resting on its feet: contacts from the first substep on
    q0 = np.array([0.0, -0.1324, 0.0521, 0.0342, 0.0679, -0.0139, -0.0589, -0.14, -0.131])
    eng.set_env_state(dict(qpos=q0, qvel=np.zeros(9)))
g = torch.Generator(device="cuda").manual_seed(0)
noise = (0.1 if name in ("pen", "penf", "tray") else (0.05 if name == "gripper" else 0.5)) * torch.randn(P, H, A, device="cuda", dtype=torch.float32 if dt == "f32" else torch.float64, generator=g)
if os.environ.get("TREE_TIME_SAME_MATES"):      # every wavefront holds copies of ONE particle (TREE_TIME_SAME_MATES = particles
    k = int(os.environ["TREE_TIME_SAME_MATES"])  # per wave: 4 at 16 lanes, 2 at 32): no Newton iteration is forced by a wave-mate -
    noise = noise[::k].repeat_interleave(k, dim=0)[:P].contiguous()     # what letting particles iterate alone could gain at most
if os.environ.get("TREE_TIME_SORT"):            # particles ordered by a similarity key before they are dealt to wavefronts (the
    mode = os.environ["TREE_TIME_SORT"]         # updates are invariant under the order): does grouping look-alikes pay?
    if mode == "first":                         # the first step's first action
        key = noise[:, 0, 0]
    elif mode == "signs":                       # the sign pattern of the first step's actions, then the first action
        bits = (noise[:, 0, :] > 0).to(torch.float64) @ (2.0 ** torch.arange(A, device="cuda", dtype=torch.float64))
        key = bits * 100.0 + noise[:, 0, 0].to(torch.float64)
    else:                                       # the mean of every action over the first four steps, projected on a fixed direction
        key = noise[:, :4, :].mean(dim=1).to(torch.float64) @ torch.linspace(1.0, 2.0, A, device="cuda", dtype=torch.float64)
    noise = noise[torch.argsort(key)].contiguous()
mean = torch.zeros(H, A, device="cuda", dtype=torch.float64)
if name == "gripper":
    mean[:, 1:] = 0.2           # (finger servos: the pose the model is drawn in)
if name in ("pen", "penf"):           # position servos: hold the start pose
    mean += torch.from_numpy(st["qp"][6:]).to(mean)
eng.rollout_device(P, H, mean, noise)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    eng.rollout_device(P, H, mean, noise)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%s %s P=%d H=%d: %.2f ms/rollout, %.2f us per particle-pair substep per SIMD, fails=%d"
      % (name, dt, P, H, ms, ms * 1e3 / (max(1.0, P / 2 / 1024) * H * raw.frame_skip), eng.solver_failures()), flush=True)

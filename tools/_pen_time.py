import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
raw = pen_hand_raw(); eng = TreeRolloutEngine(raw, dtype="f64"); st = holding_state()
eng.set_env_state(dict(qp=st["qp"], qv=st["qv"], target_pos=np.asarray(raw.target_pos, float)))
P, H = 4096, 32
g = torch.Generator(device="cuda").manual_seed(0)
noise = 0.1 * torch.randn(P, H, 24, device="cuda", dtype=torch.float64, generator=g)
mean = torch.from_numpy(np.tile(st["qp"][6:], (H, 1))).cuda()
eng.rollout_device(P, H, mean, noise); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): eng.rollout_device(P, H, mean, noise)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("MJMPC_AMD_LIB", "default"), "pen kernel 4096x32: %.2f ms, failures %d" % (e0.elapsed_time(e1) / 3, eng.solver_failures()))

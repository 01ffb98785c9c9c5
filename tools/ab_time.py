"""A/B timing of the fused rollout entry point at two states (start: qpos0, limits active; settled: arm at the target) for
the library in MJMPC_AMD_LIB.   python tools/ab_time.py [P] [dtype]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
H = 32
eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dtype)
tdt = torch.float64 if dtype == "f64" else torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
noise = torch.randn(P, H, 7, device="cuda", dtype=tdt, generator=g)
mean = torch.zeros(H, 7, device="cuda", dtype=torch.float64)
coeffs = torch.tensor([0.25, 0.8, 0.0], dtype=torch.float64, device="cuda")
gseq = torch.ones(H, dtype=torch.float64, device="cuda")
states = {"start": dict(qp=np.zeros(7), qv=np.zeros(7)),
          "settled": dict(qp=np.array([0.45, 0.35, -0.2, -0.9, 0.3, -0.4, 0.1]), qv=np.zeros(7))}
# warm the clocks
for _ in range(300):
    eng.rollout_fused(P, H, mean, noise, coeffs, gseq)
torch.cuda.synchronize()
out = []
for name, st in states.items():
    eng.set_env_state(dict(st, target_pos=np.array([0.1, 0.1, 0.1])))
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            eng.rollout_fused(P, H, mean, noise, coeffs, gseq)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    out.append("%s %.1f us" % (name, best))
print(os.path.basename(os.environ.get("MJMPC_AMD_LIB", "default")), P, dtype, " | ".join(out), "fails", eng.solver_failures())

"""Do a particle's costs in the ARM kernels depend on its wave-mates?  python tools/arm_mates_check.py
(MJMPC_AMD_LIB selects a build, e.g. one made by `tools/dev_build.sh armpp -DARM_PER_PARTICLE`).  For every launch shape
(four-wave flags up to 2048 particles, DUO up to 4096, SOLO above) the same particles are rolled out in another order."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
for dt in ("f64", "f32"):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    rs = np.random.RandomState(3)
    q0 = np.array([0.3, -1.0, 0.2, 1.9, 0.1, 0.6, 0.0])
    eng.set_env_state(dict(qp=q0, qv=0.5 * rs.standard_normal(7), target_pos=np.array([0.1, 0.1, 0.1])))
    H = 16
    npdt = np.float32 if dt == "f32" else np.float64
    for P in (256, 4096, 8192):
        noise = (2.0 * rs.standard_normal((P, H, 7))).astype(npdt)
        mean = np.zeros((H, 7))

        def run(nz):
            return eng.rollout_device(nz.shape[0], H, mean, nz)[0].cpu().numpy().copy()

        c = run(noise)
        perm = rs.permutation(P)
        cp = run(noise[perm])
        d = np.abs(cp - c[perm])
        print(dt, "P=%d" % P, "permuted: equal" if np.array_equal(cp, c[perm]) else "permuted: DIFFERENT, max |diff| %.3g in %d of %d particles"
              % (d.max(), int((d.max(axis=1) > 0).sum()), P), flush=True)

"""Phase clocks of the arm rollout kernel (developer tool).

    python tools/stamps.py [P] [dtype]      # builds an instrumented copy of the library (-DMJMPC_STAMPS) on first use

Prints, for workgroup 0, the shader cycles per substep each wavefront spent in each phase (arm_rollout.hip, Stamps).
"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "_build", "libmjmpc_amd_stamps.so")
CSRC = os.path.join(ROOT, "mjmpc_amd", "csrc")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-DMJMPC_STAMPS", "-I", CSRC] + [a for a in sys.argv[1:] if a.startswith("-D")] + sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + ["-o", LIB]
    subprocess.check_call(cmd)


NAMES = ["kinematics", "link frames", "velocities+bias (DYN)", "wait A (DYN) / tiles (QUAD)", "CRBA+tile", "rows", "factor #1",
         "wait A (SOLVE) / tau (QUAD)", "solve #1 + check", "more Newton iterations", "force + Euler solve", "records (DYN)", "wait B / X",
         "integrate", "obs records / loop | EI wait, rows (flags)", "EI wait, no rows (flags)"]

if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    os.environ["MJMPC_AMD_LIB"] = os.environ.get("STAMPS_LIB", LIB)      # (STAMPS_LIB: a tools/dev_build.sh library)
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    P = int(args[0]) if args else 4096
    dt = args[1] if len(args) > 1 else "f64"
    H = 32
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    lib = _lib.load()
    lib.mjmpc_debug_stamps.restype = ctypes.c_int
    lib.mjmpc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    g = torch.Generator(device="cuda").manual_seed(0)
    noise = torch.randn(P, H, 7, device="cuda", dtype=torch.float32 if dt == "f32" else torch.float64, generator=g)
    mean = torch.zeros(H, 7, device="cuda", dtype=torch.float64)
    out = (ctypes.c_uint64 * 64)()
    for _ in range(3):
        eng.rollout_device(P, H, mean, noise)
    lib.mjmpc_debug_stamps(eng._h, out)
    eng.rollout_device(P, H, mean, noise)
    lib.mjmpc_debug_stamps(eng._h, out)
    nsub = H * 2
    if "--timeline" in sys.argv:        # a library built with -DMJMPC_STAMPS_ABS=n: absolute clocks of substep n
        ev = [(out[16 * w + k], w, k) for w in range(4) for k in range(16) if out[16 * w + k]]
        t0 = min(e[0] for e in ev)
        for t, w, k in sorted(ev):
            print("%8d  %s wave %d  %s" % (t - t0, "      " * w, w, NAMES[k]))
        sys.exit(0)
    for w in range(4):
        tot = sum(out[16 * w + k] for k in range(16))
        if tot == 0:
            continue
        print("wave %d  (%s, P=%d): %.0f cycles per substep" % (w, dt, P, tot / nsub))
        for k in range(16):
            if out[16 * w + k]:
                print("   %-26s %8.0f  %5.1f %%" % (NAMES[k], out[16 * w + k] / nsub, 100.0 * out[16 * w + k] / tot))

"""Phase clocks and Newton statistics of the tree rollout kernel (developer tool).

    python tools/tree_stats.py --build                      # instrumented copy of the library (-DTREE_STATS)
    python tools/tree_stats.py [hand|swimmer|cheetah|pen|cartpole|tray|door] [dtype] [P] [H]

Prints, for the first particle of the launch, shader cycles per substep in each phase and the constraint statistics."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("TREE_STATS_LIB", os.path.join(ROOT, "tools", "_build", "libmjmpc_amd_treestats.so"))
CSRC = os.path.join(ROOT, "mjmpc_amd", "csrc")
NAMES = ["kinematics + contact geometry", "link quantities + bias forces (+ fluid)", "composite inertia + mass-matrix row",
         "constraint rows", "Newton iterations", "Euler factor + solve", "-", "integrate + records"]

if __name__ == "__main__":
    if "--build" in sys.argv:
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        from mjmpc_amd.build import build
        build(extra_flags=["-DTREE_STATS"], lib=LIB)         # (the product build's per-source flags + the clocks)
        sys.exit(0)
    os.environ["MJMPC_AMD_LIB"] = LIB
    import numpy as np
    import torch
    from mjmpc_amd import _lib
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.hand24 import hand24_raw
    from mjmpc_amd.models.swimmer import swimmer_raw
    args = sys.argv[1:]
    name = args[0] if args else "cheetah"
    dt = args[1] if len(args) > 1 else "f64"
    P = int(args[2]) if len(args) > 2 else 4096
    H = int(args[3]) if len(args) > 3 else 32
    from mjmpc_amd.models.pen_hand import holding_state, pen_hand_raw
    if name in ("cartpole", "tray", "door", "gripper"):
        from mjmpc_amd.models.synthetic import start_state, synthetic_raw
        raw = synthetic_raw(name)
    else:
        raw = dict(hand=hand24_raw, swimmer=swimmer_raw, cheetah=half_cheetah_raw, pen=pen_hand_raw)[name]()
    eng = TreeRolloutEngine(raw, dtype=dt)
    if name in ("cartpole", "tray", "door", "gripper"):
        eng.set_env_state(start_state(name, raw))
    if name == "pen":
        st = holding_state()
        eng.set_env_state(dict(qpos=st["qp"], qvel=st["qv"], target_pos=np.asarray(raw.target_pos, float)))
    if name == "cheetah":
        eng.set_env_state(dict(qpos=np.array([0.0, -0.1324, 0.0521, 0.0342, 0.0679, -0.0139, -0.0589, -0.14, -0.131]), qvel=np.zeros(9)))
    lib = _lib.load()
    lib.mjmpc_debug_tree_stats.restype = ctypes.c_int
    lib.mjmpc_debug_tree_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    A = eng.d_action
    g = torch.Generator(device="cuda").manual_seed(0)
    noise = (0.1 if name in ("pen", "tray") else (0.05 if name == "gripper" else 0.5)) * torch.randn(P, H, A, device="cuda", dtype=torch.float32 if dt == "f32" else torch.float64, generator=g)
    mean = torch.zeros(H, A, device="cuda", dtype=torch.float64)
    if name == "gripper":
        mean[:, 1:] = 0.2           # (finger servos: the pose the model is drawn in)
    if name == "pen":
        mean += torch.from_numpy(st["qp"][6:]).to(mean)
    out = (ctypes.c_ulonglong * 48)()
    eng.rollout_device(P, H, mean, noise)
    lib.mjmpc_debug_tree_stats(eng._h, out)
    eng.rollout_device(P, H, mean, noise)
    lib.mjmpc_debug_tree_stats(eng._h, out)
    v = list(out)
    nsub = max(v[8], 1)
    print("%s %s P=%d H=%d: %d substeps of particle 0" % (name, dt, P, H, nsub))
    tot = sum(v[:8])
    for i, nm in enumerate(NAMES):
        if nm != "-":
            print("  %-45s %8.0f cycles/substep  %5.1f %%" % (nm, v[i] / nsub, 100.0 * v[i] / tot))
    ni = max(v[11], 1)
    print("  per Newton iteration: assemble H %.0f, factor %.0f, solve %.0f, next active set %.0f, line search %.0f, rank-one correction %.0f cycles"
          % (v[12] / ni, v[13] / ni, v[14] / ni, v[15] / ni, v[22] / ni, v[23] / ni))
    if "fine" in LIB:       # a -DTREE_STATS_FINE build (TREE_STATS_LIB=tools/_build/libmjmpc_amd_treestatsfine.so)
        print("  fine, per iteration: H/rhs start %.0f, per-point assembly %.0f, factor %.0f, solve %.0f, owners' walk %.0f, sets %.0f, line search %.0f; "
              "rank-one: solve %.0f, sums %.0f, re-check %.0f; tail %.0f cycles" % tuple(v[k] / ni for k in (24, 12, 13, 14, 15, 22, 26, 27, 28, 29, 23)))
        print("  fine, per substep with rows: before the loop %.0f, after it (constraint force) %.0f cycles" % (v[31] / max(v[10], 1), v[30] / max(v[10], 1)))
    print("  iterations that found a changed set: %d - one limit row %d, one contact row %d, several rows %d, only the wave's other particle %d; friction-loss zone changes %d"
          % (v[16], v[17], v[18], v[19], v[20], v[21]))
    print("  total %.0f cycles/substep; contact points per substep %.2f; substeps with rows %.0f %%; Newton iterations per such substep %.2f"
          % (tot / nsub, v[9] / nsub, 100.0 * v[10] / nsub, v[11] / max(v[10], 1)))

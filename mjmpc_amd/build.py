"""Build the HIP library in-tree:  python -m mjmpc_amd.build [--force] [-v]

Every mjmpc_amd/csrc/*.hip is compiled for gfx950 into an object under mjmpc_amd/_build/ (only when it or a header
is newer than its object; the sources compile side by side) and the objects are linked into
mjmpc_amd/libmjmpc_amd.so.  hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun
snapshots.
"""
import glob
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libmjmpc_amd.so")
INFO = os.path.join(HERE, "build_info.json")     # which tuning alternative every source was compiled with (travels with the .so)
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-I", CSRC]
# The tree kernel's 16-lane dense instantiations (tree_rollout_dense.hip) run ONE wave per SIMD through long straight-line
# phases: LLVM's iterative schedulers interleave their independent dependency chains better than the default strategy -
# measured on MI355X: HalfCheetah 4096 x 32 f64 3.55 -> 3.12 ms, Swimmer 1.50 -> 1.33 ms, f32 32768 x 32 13.2 -> 12.4 ms
# (iterative-ilp and iterative-maxocc alike).  The 32-lane instantiations gain a per cent or two (below); the arm kernel
# (hand-placed scheduling barriers between its phases) gains a per cent.  These schedulers crash this compiler on SOME variants of the kernel
# (which ones changes with unrelated edits), so a source lists alternatives: the first that compiles is used, the plain
# flags last.
# Round 6 (after the solver's control flow changed: per-particle decisions, no rank-one correction in the 16-lane kernels):
# iterative-ilp now beats iterative-maxocc on the dense instantiations by 2 - 4 % (A/B on one box, 4096 x 32 f64 launches:
# HalfCheetah 2.005 -> 1.96 ms, Swimmer 0.965 -> 0.93, tray 2.795 -> 2.685, door 0.87 -> 0.85, pen-in-hand 9.62 -> 9.295, f32
# pen 7.95 -> 7.74; the default strategy: 2.09 / 0.99 / 2.865 / 0.885 / 10.03) and on the elliptic-cone instantiations in f64
# (gripper 3.06 -> 2.97; f32 2.68 -> 2.715); the 32-lane tree-sparse source keeps maxocc (hand 2.11 -> 2.155 with ilp):
# profiles/r06_sched_ab.txt.
PER_SOURCE_FLAGS = {"tree_rollout_dense.hip": [["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
                                               ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]],
                    # the 32-lane instantiations: iterative-maxocc takes 1.6 % (f64) / 2.5 % (f32) off the hand at
                    # 65 536 x 64 (52.7 -> 51.9 ms, 31.8 -> 31.0 ms from tools/tree_time.py's start state; two A/B pairs);
                    # max-ilp is 1-6 % slower everywhere, iterative-ilp has crashed the compiler on this source
                    "tree_rollout.hip": [["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]],
                    "tree_rollout_cone.hip": [["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
                                              ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]],
                    # the arm kernel: 1 % (f64 control step 0.2000 -> 0.1980 ms, three A/B pairs on one box; f32 2 %) with
                    # iterative-maxocc; iterative-ilp another 1 % on the fused iteration's kernel and 3 % in f32 (two A/B
                    # pairs: f64 control step 0.1957 -> 0.1935 ms, pipelined 0.1910 -> 0.1883; the plain two-wave launch
                    # +0.5 %, one-wave launches unchanged; max-ilp as maxocc, iterative-minreg / max-memory-clause slower)
                    "arm_rollout.hip": [["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
                                        ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]],
                    # (the extended-joint build of the same kernels: arm_rollout_xj.hip includes arm_rollout.hip)
                    "arm_rollout_xj.hip": [["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
                                           ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]]}


def flags_for(src, alternative=0):
    alts = PER_SOURCE_FLAGS.get(os.path.basename(src), [])
    return FLAGS + (alts[alternative] if alternative < len(alts) else [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + [os.path.join(HERE, "..", "include", "mjmpc_amd.h")]
            + [os.path.join(CSRC, "tree_rollout.hip"),         # (tree_rollout_dense.hip and tree_rollout_cone.hip include it)
               os.path.join(CSRC, "arm_rollout.hip")])         # (arm_rollout_xj.hip includes it)


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=None):
    """``extra_flags`` / ``lib``: developer builds (e.g. ``-DMJMPC_STAMPS`` into tools/_build/) - they compile every
    source afresh into their own object directory beside ``lib``."""
    lib = lib or LIB
    objdir = OBJ if lib == LIB else os.path.join(os.path.dirname(lib), "_obj_" + os.path.basename(lib))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = _headers()
    hdrs = hdrs + [os.path.abspath(__file__)]          # (the flags live in this file)
    if not force and lib == LIB and not _stale(lib, sources() + hdrs):
        return lib              # (the objects do not travel with gpurun snapshots; the library does)
    todo = [s for s in sources()
            if force or lib != LIB or _stale(os.path.join(objdir, os.path.basename(_obj(s))), [s] + hdrs)]
    if not todo and not _stale(lib, [os.path.join(objdir, os.path.basename(_obj(s))) for s in sources()]):
        return lib
    os.makedirs(objdir, exist_ok=True)

    chosen = {}

    def compile_one(src):
        out = os.path.join(objdir, os.path.basename(_obj(src)))
        alts = PER_SOURCE_FLAGS.get(os.path.basename(src), [])
        for k, special in enumerate(list(alts) + [[]]):
            cmd = [hipcc] + FLAGS + special + list(extra_flags) + ["-c", src, "-o", out]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode == 0 or not special:
                chosen[os.path.basename(src)] = {"flags": " ".join(special), "alternative": k, "of": len(alts)}
                break
            # a scheduling strategy is a tuning flag: if this compiler cannot take it for this source, try the next -
            # loudly (stdout too): the committed profiles were measured with the FIRST alternative of every source
            msg = ("mjmpc_amd.build: WARNING %s did not compile with %s; trying the next alternative (performance figures "
                   "under profiles/ were measured with the first)\n" % (os.path.basename(src), " ".join(special)))
            sys.stderr.write(msg)
            sys.stdout.write(msg)
        return src, r

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(todo)))) as ex:
        results = list(ex.map(compile_one, todo))
    for src, r in results:
        if verbose or r.returncode != 0:
            sys.stderr.write(r.stderr)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, "hipcc -c " + src)
    objs = [os.path.join(objdir, os.path.basename(_obj(s))) for s in sources()]
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", lib])
    # which alternative each source took: kept beside the library (sources not recompiled keep their earlier entry)
    info_path = INFO if lib == LIB else lib + ".build_info.json"
    info = {}
    if os.path.exists(info_path):
        try:
            with open(info_path) as f:
                info = json.load(f).get("sources", {})
        except (OSError, ValueError):
            info = {}
    info = {k: v for k, v in info.items() if k in {os.path.basename(s) for s in sources()}}
    info.update(chosen)
    for s_ in sources():
        info.setdefault(os.path.basename(s_), {"flags": "unknown (object older than build_info.json)", "alternative": -1,
                                               "of": len(PER_SOURCE_FLAGS.get(os.path.basename(s_), []))})
    with open(info_path, "w") as f:
        json.dump({"arch": ARCH, "common_flags": " ".join(FLAGS[:-2]), "extra_flags": " ".join(extra_flags),
                   "sources": info}, f, indent=1, sort_keys=True)
    return lib


def build_info(lib=None):
    """{source: {"flags", "alternative", "of"}} of the library in the tree (bench.py echoes it in `config`); a source
    compiled with anything but its first alternative is a build whose performance differs from profiles/."""
    path = INFO if lib in (None, LIB) else lib + ".build_info.json"
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {"sources": {}, "note": "build_info.json missing: library built by an older build.py"}


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv or "--force" in sys.argv)
    print("built", LIB)

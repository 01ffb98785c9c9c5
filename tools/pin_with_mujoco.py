"""Put the oracle beside MuJoCo itself, wherever MuJoCo is installed - ONE command to the pin this repository's image cannot
provide (physics parity is unpinned here; this script has never met a MuJoCo):

    pip install mujoco            # (or a machine with mujoco_py 2.0, the version the reference pins)
    python tools/pin_with_mujoco.py            # every model below, a per-feature PASS / FAIL table, exit code 0 iff all pass
    python tools/pin_with_mujoco.py reacher door random:17 ...     # a selection (random:SEED = tests/test_random_models_gpu.py)
    python tools/pin_with_mujoco.py --tol 1e-8
    python tools/pin_with_mujoco.py --self-test       # no MuJoCo needed: the oracle stands in for it (plumbing check only)

For every model: export it as MJCF (mjmpc_amd/models/export_mjcf.py), load that text into MuJoCo (the `mujoco` bindings, or
`mujoco_py` 2.0), and compare with oracle/reacher_ref.c (`RefArm.step` = or_step):
  * ONE mj_step from 32 random states (positions around qpos0, unit velocities, uniform controls) - the pin proper:
    worst |d qpos|, |d qvel| against `--tol` (default 1e-9 relative to max(1, |value|));
  * a 200-step trajectory, each side from its own previous state (reported, not judged: contacts amplify rounding);
  * the RESET case: one step from a state MuJoCo's mj_checkVel rejects (velocities of 1e11) - both sides must land on the
    same post-reset state (with mujoco_py, whose default warning callback raises instead, the case is reported as skipped).
With the `mujoco` bindings (>= 2.1.2) a capsule's volume counts its end caps fully: the model is exported and the oracle
built with capsule_cap_factor = 4/3; mujoco_py 2.0 keeps MuJoCo 2.0's pi r^2 (h + r).

The table is BY FEATURE: a feature passes when every model that exercises it passes.  Three colliders are this repository's
own closed forms, not MuJoCo's routines (capsule-box, box-box, sphere / capsule-cylinder: oracle/reacher_ref.c:1522,1986,2026) -
models that can bring them into contact are listed as EXPECTED DEVIATIONS: their result is printed but does not fail the run.
Writes nothing."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# model -> the features it exercises (what the pin of that model says something about)
MODELS = {
    # the three models the reference vendors (mjmpc/envs/assets: sawyer.xml, half_cheetah.xml, swimmer.xml)
    "reacher":      ["hinge chain (CRBA, RNE, Euler with implicit damping)", "joint limits", "frictionless contact: plane-sphere",
                     "motor actuators + ctrlrange"],
    "half_cheetah": ["planar floating base (slide, slide, hinge)", "joint limits", "springs + damping",
                     "pyramidal condim-3 contacts: plane-capsule"],
    "swimmer":      ["planar floating base (slide, slide, hinge)", "fluid forces (density, viscosity)", "capsule-capsule pairs"],
    # the five synthetic ones (mjmpc_amd/models/assets/*.xml) and the two hands
    "cartpole":     ["slide + hinge", "friction loss"],
    "door":         ["friction loss", "equality: joint coupling", "position actuators", "pyramidal condim-3 contacts: sphere-box"],
    "tray":         ["free body on a servoed platform", "position actuators", "pyramidal condim-3 contacts: sphere-box"],
    "fourbar":      ["equality: connect", "ball joint", "fixed tendon + tendon limits"],
    "gripper":      ["elliptic cones + impratio", "position actuators", "box-plane contacts"],
    "hand24":       ["24-dof tree (elimination-tree LDL)", "position actuators", "joint limits"],
    "pen_hand":     ["free body in a 24-dof hand", "pyramidal condim-3 contacts: capsule-capsule, sphere-capsule"],
}
# models whose geoms can meet through one of the own-scheme colliders: judged apart
OWN_SCHEME = {"gripper": "box-box, capsule-box, capsule-cylinder (closed forms of this repository)"}
DEFAULT = ["reacher", "half_cheetah", "swimmer", "cartpole", "door", "tray", "fourbar", "gripper", "hand24", "pen_hand"]


def _backend():
    try:
        import mujoco

        class New:
            name, cap, raises_on_reset = "mujoco %s" % mujoco.__version__, 4.0 / 3.0, False

            def __init__(self, xml):
                self.m = mujoco.MjModel.from_xml_string(xml)
                self.d = mujoco.MjData(self.m)

            def step(self, q, v, u):
                self.d.qpos[:], self.d.qvel[:], self.d.ctrl[:] = q, v, u
                self.d.qacc_warmstart[:] = 0
                mujoco.mj_step(self.m, self.d)
                return self.d.qpos.copy(), self.d.qvel.copy()
        return New
    except ImportError:
        pass
    try:
        import mujoco_py

        class Old:
            name, cap, raises_on_reset = "mujoco_py %s" % getattr(mujoco_py, "__version__", "?"), 1.0, True

            def __init__(self, xml):
                self.sim = mujoco_py.MjSim(mujoco_py.load_model_from_xml(xml))

            def step(self, q, v, u):
                st = self.sim.get_state()
                self.sim.set_state(mujoco_py.MjSimState(st.time, np.asarray(q, float), np.asarray(v, float), st.act, st.udd_state))
                self.sim.data.ctrl[:] = u
                self.sim.data.qacc_warmstart[:] = 0
                self.sim.step()
                return self.sim.data.qpos.copy(), self.sim.data.qvel.copy()
        return Old
    except ImportError:
        return None


def _model(name):
    if name.startswith("random:"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("rm", os.path.join(ROOT, "tests", "test_random_models_gpu.py"))
        rm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rm)
        return rm.random_model(int(name.split(":")[1]))
    if name in ("cartpole", "tray", "door", "fourbar", "gripper"):
        from mjmpc_amd.models.synthetic import synthetic_raw
        return synthetic_raw(name)
    from mjmpc_amd.models import half_cheetah, hand24, pen_hand, reacher7dof, swimmer
    return dict(reacher=reacher7dof.reacher7dof_raw, half_cheetah=half_cheetah.half_cheetah_raw, swimmer=swimmer.swimmer_raw,
                hand24=hand24.hand24_raw, pen_hand=pen_hand.pen_hand_raw)[name]()


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b))))) if len(a) else 0.0


def compare(B, name, tol):
    """-> dict(one_step, traj, reset) for one model; one_step is what is judged."""
    from mjmpc_amd.models.export_mjcf import to_mjcf
    from oracle.physics_ref import RefArm
    raw = _model(name)
    raw.capsule_cap_factor = B.cap
    mj, ref = B(to_mjcf(raw)), RefArm(raw.to_flat())
    rs = np.random.RandomState(0)
    nu = len(raw.actuators)
    worst = 0.0
    for _ in range(32):                                         # one step from random states
        q = raw.qpos0 + 0.2 * rs.standard_normal(raw.nq) * (np.abs(raw.qpos0) < 1e-12)
        v = rs.standard_normal(raw.nv)
        u = rs.uniform(-1, 1, nu)
        q1, v1 = mj.step(q, v, u)
        q2, v2, _, _ = ref.step(q, v, u)
        worst = max(worst, _rel(q2, q1), _rel(v2, v1))
    q, v = raw.qpos0.copy(), np.zeros(raw.nv)                   # a trajectory: both sides from their own previous state
    qo, vo = q.copy(), v.copy()
    for _ in range(200):
        u = rs.uniform(-1, 1, nu)
        q, v = mj.step(q, v, u)
        qo, vo, _, _ = ref.step(qo, vo, u)
    traj = max(_rel(qo, q), _rel(vo, v))
    reset = None                                                # MuJoCo's reset on instability (mj_checkVel -> mj_resetData)
    if not B.raises_on_reset:
        v_bad = 1e11 * np.ones(raw.nv)
        u = np.zeros(nu)
        q1, v1 = mj.step(raw.qpos0.copy(), v_bad, u)
        q2, v2, _, _, n_resets = ref.step_mj(raw.qpos0.copy(), v_bad, u)       # (or_step_mj: mj_step WITH its checks)
        assert n_resets >= 1, "the oracle did not reset from velocities of 1e11"
        reset = max(_rel(q2, q1), _rel(v2, v1))
    return dict(one_step=worst, traj=traj, reset=reset, ok=worst <= tol, reset_ok=(reset is None or reset <= tol))


def main(argv):
    tol = 1e-9
    if "--tol" in argv:
        i = argv.index("--tol")
        tol = float(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    B = _backend()
    if "--self-test" in argv:
        # plumbing check where no MuJoCo exists (this image): the ORACLE stands in for MuJoCo, so every comparison is 0 by
        # construction - it proves that the export, the state generators, the reset case and the table run, nothing else
        argv = [a for a in argv if a != "--self-test"]
        from mjmpc_amd.models.export_mjcf import to_mjcf  # noqa: F401  (the export itself is exercised in compare)

        class Self:
            name, cap, raises_on_reset = "oracle standing in for MuJoCo (--self-test: pins NOTHING)", 1.0, False

            def __init__(self, xml):
                import tempfile
                from mjmpc_amd.models.mjcf import load_mjcf
                from oracle.physics_ref import RefArm
                with tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False) as f:
                    f.write(xml)                                # (through the exported text, like MuJoCo would)
                try:
                    self.ref = RefArm(load_mjcf(f.name).to_flat())
                finally:
                    os.unlink(f.name)

            def step(self, q, v, u):
                q1, v1, _, _, _ = self.ref.step_mj(q, v, u)
                return q1, v1
        B = Self
    if B is None:
        print("neither `mujoco` nor `mujoco_py` can be imported here: nothing to compare with (the oracle stays unpinned).\n"
              "Models that WOULD be compared: %s\nFeatures that would be judged:" % ", ".join(DEFAULT))
        for f in sorted({f for m in DEFAULT for f in MODELS[m]}):
            print("   ", f)
        print("Expected deviations (own-scheme colliders):", "; ".join("%s: %s" % kv for kv in OWN_SCHEME.items()))
        return 2
    names = argv or DEFAULT
    print("backend: %s   tolerance %g (relative to max(1, |value|), one mj_step)" % (B.name, tol))
    results = {}
    for name in names:
        try:
            results[name] = compare(B, name, tol)
        except Exception as e:      # a model MuJoCo or the oracle refuses is a finding of its own
            results[name] = dict(error="%s: %s" % (type(e).__name__, e), ok=False, reset_ok=False)
        r = results[name]
        if "error" in r:
            print("%-14s ERROR %s" % (name, r["error"]))
        else:
            print("%-14s one step %.2e %s | 200-step trajectory %.2e | reset %s%s"
                  % (name, r["one_step"], "PASS" if r["ok"] else "FAIL", r["traj"],
                     "skipped (mujoco_py raises)" if r["reset"] is None else "%.2e %s" % (r["reset"], "PASS" if r["reset_ok"] else "FAIL"),
                     "   [expected deviation: %s]" % OWN_SCHEME[name] if name in OWN_SCHEME else ""))
    # ---- the per-feature table
    feats = {}
    for name in names:
        for f in MODELS.get(name, ["random model (generator of tests/test_random_models_gpu.py)"]):
            feats.setdefault(f, []).append(name)
    print("\nfeature                                                          verdict   models")
    failed = []
    for f in sorted(feats):
        judged = [m for m in feats[f] if m not in OWN_SCHEME]
        ok = all(results[m]["ok"] for m in judged)
        verdict = "PASS" if (judged and ok) else ("FAIL" if judged else "n/a *")
        if judged and not ok:
            failed.append(f)
        print("%-64s %-9s %s" % (f, verdict, ", ".join("%s%s" % (m, "" if m not in OWN_SCHEME else "*") for m in feats[f])))
    rs_models = [m for m in names if results[m].get("reset") is not None]
    if rs_models:
        ok = all(results[m]["reset_ok"] for m in rs_models if m not in OWN_SCHEME)
        print("%-64s %-9s %s" % ("reset on instability (mj_checkVel -> mj_resetData)", "PASS" if ok else "FAIL", ", ".join(rs_models)))
        if not ok:
            failed.append("reset on instability")
    if any(m in OWN_SCHEME for m in names):
        print("* expected deviations - colliders that are this repository's closed forms, not MuJoCo's routines; not judged:")
        for m in names:
            if m in OWN_SCHEME and "error" not in results[m]:
                print("    %-12s one step %.2e  (%s)" % (m, results[m]["one_step"], OWN_SCHEME[m]))
    if "pins NOTHING" in B.name:
        print("\nSELF-TEST ONLY: the oracle compared with itself through the exported MJCF - the tool runs; nothing is pinned")
    else:
        print("\n%s" % ("PINNED: every judged feature within tolerance - DESIGN 2 / README may drop 'parity unpinned' for them"
                        if not failed else "NOT PINNED: %s" % "; ".join(failed)))
    return 0 if not failed else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

"""Put the oracle beside MuJoCo itself, wherever MuJoCo is installed (it is NOT in this repository's image: physics parity
is unpinned here, DESIGN 2 - this script is the way to pin it, and has never met a MuJoCo).

    python tools/pin_with_mujoco.py [reacher|half_cheetah|swimmer|hand24|pen_hand|cartpole|tray|door|fourbar|random:SEED ...]

For every model: export it as MJCF (mjmpc_amd/models/export_mjcf.py), load that text into MuJoCo (the `mujoco` bindings, or
`mujoco_py` 2.0 - the version the reference pins), and compare ONE mj_step from random states, and a short trajectory, with
oracle/reacher_ref.c (or_step).  With the `mujoco` bindings (>= 2.1.2) a capsule's volume counts its end caps fully:
the model is exported and the oracle built with capsule_cap_factor = 4/3; mujoco_py 2.0 keeps MuJoCo 2.0's pi r^2 (h + r).
Prints the worst absolute differences of qpos / qvel; writes nothing."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _backend():
    try:
        import mujoco

        class New:
            name, cap = "mujoco %s" % mujoco.__version__, 4.0 / 3.0

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
            name, cap = "mujoco_py %s" % getattr(mujoco_py, "__version__", "?"), 1.0

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


def main():
    B = _backend()
    if B is None:
        print("neither `mujoco` nor `mujoco_py` can be imported here: nothing to compare with (the oracle stays unpinned)")
        return 2
    from mjmpc_amd.models.export_mjcf import to_mjcf
    from oracle.physics_ref import RefArm
    names = sys.argv[1:] or ["reacher", "half_cheetah", "swimmer", "hand24", "cartpole", "door"]
    print("backend:", B.name)
    for name in names:
        raw = _model(name)
        raw.capsule_cap_factor = B.cap
        mj, ref = B(to_mjcf(raw)), RefArm(raw.to_flat())
        rs = np.random.RandomState(0)
        worst_q = worst_v = 0.0
        for _ in range(32):                                     # one step from random states
            q = raw.qpos0 + 0.2 * rs.standard_normal(raw.nq) * (np.abs(raw.qpos0) < 1e-12)
            v = rs.standard_normal(raw.nv)
            u = rs.uniform(-1, 1, len(raw.actuators))
            q1, v1 = mj.step(q, v, u)
            q2, v2, _, _ = ref.step(q, v, u)
            worst_q, worst_v = max(worst_q, np.abs(q1 - q2).max()), max(worst_v, np.abs(v1 - v2).max())
        q, v = raw.qpos0.copy(), np.zeros(raw.nv)               # a trajectory: both sides from their own previous state
        qo, vo = q.copy(), v.copy()
        for _ in range(200):
            u = rs.uniform(-1, 1, len(raw.actuators))
            q, v = mj.step(q, v, u)
            qo, vo, _, _ = ref.step(qo, vo, u)
        print("%-14s one step: |dqpos| %.2e |dqvel| %.2e;  after 200 steps: |dqpos| %.2e |dqvel| %.2e"
              % (name, worst_q, worst_v, np.abs(q - qo).max(), np.abs(v - vo).max()))
    return 0


if __name__ == "__main__":
    sys.exit(main())

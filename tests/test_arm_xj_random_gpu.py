"""GPU: serial chains drawn from seeds on the arm kernels' extended-joint build - 1 to 7 dofs, hinge and slide joints in any
order and along any axis, limits, dry friction on a random subset of the dofs (own solreffriction), armature, gravity on or off,
an optional tip sphere over a frictionless floor, motors on the first nu <= nv dofs - against oracle/reacher_ref.c at 1e-9 per
env step and over short rollouts, and against the general tree engine.  (The physics these chains exercise - slide joints in the
contact Jacobian, friction-loss zones with the exact line search, limit rows on slide dofs - is what round 6 added to the arm
kernels; the tree engine and the oracle had it.)   MJMPC_XJ_SEEDS=a:b runs another range (a soak, not part of the suite)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def chain_xml(seed):
    rs = np.random.RandomState(seed)
    nv = int(rs.randint(1, 8))
    nu = int(rs.randint(1, nv + 1))
    grav = "0 0 -9.81" if rs.rand() < 0.7 else "0 0 0"
    floor = rs.rand() < 0.5
    dt = float(rs.choice([0.002, 0.005, 0.01]))
    sf = "%.3f %.2f" % (rs.uniform(0.01, 0.05), rs.uniform(0.8, 1.3))
    body, close = "", ""
    z = 0.0
    for k in range(nv):
        slide = rs.rand() < 0.4
        ax = rs.standard_normal(3)
        ax /= np.linalg.norm(ax)
        if rs.rand() < 0.5:
            ax = np.eye(3)[rs.randint(3)]
        length = rs.uniform(0.15, 0.4)
        lim = rs.rand() < 0.6
        rng = (rs.uniform(-0.5, -0.1), rs.uniform(0.1, 0.5)) if slide else (rs.uniform(-2.0, -0.3), rs.uniform(0.3, 2.0))
        fl = rs.uniform(0.02, 0.5) if rs.rand() < 0.6 else 0.0
        body += ('<body name="b%d" pos="0 0 %.4f"><joint name="j%d" type="%s" axis="%.6f %.6f %.6f" damping="%.3f" armature="%.4f"%s%s/>'
                 '<geom type="capsule" fromto="0 0 0 0 0 %.4f" size="%.3f" density="%.1f"/>'
                 % (k, z, k, "slide" if slide else "hinge", ax[0], ax[1], ax[2], rs.uniform(0.02, 0.5), rs.uniform(0.0, 0.02),
                    (' limited="true" range="%.4f %.4f"' % rng) if lim else "",
                    (' frictionloss="%.4f"' % fl) if fl > 0 else "", -length, rs.uniform(0.02, 0.05), rs.uniform(300, 1500)))
        close += "</body>"
        z = -length
    body += '<geom name="tip" type="sphere" pos="0 0 %.4f" size="0.04" %s/><site name="finger" pos="0 0 %.4f"/>' % (
        z, 'contype="1" conaffinity="1"' if floor else "", z)
    acts = "".join('<motor joint="j%d" gear="%.2f" ctrlrange="-1 1" ctrllimited="true"/>' % (k, rs.uniform(2, 20)) for k in range(nu))
    xml = ('<mujoco><compiler angle="radian" coordinate="local" inertiafromgeom="true"/>'
           '<option timestep="%g" gravity="%s" integrator="Euler"/>'
           '<default><joint solreffriction="%s"/><geom contype="0" conaffinity="0" condim="1"/></default>'
           '<worldbody><site name="target" pos="0.2 0.1 -0.5"/>%s'
           '%s%s</worldbody><actuator>%s</actuator></mujoco>'
           % (dt, grav, sf, ('<geom name="floor" type="plane" pos="0 0 %.3f" size="3 3 0.1" contype="1" conaffinity="1" condim="1"/>'
                             % (-0.9 * sum(1 for _ in range(nv)) * 0.25 - 0.1)) if floor else "", body, close, acts))
    return xml, nv, nu


_SEEDS = range(*[int(x) for x in os.environ["MJMPC_XJ_SEEDS"].split(":")]) if os.environ.get("MJMPC_XJ_SEEDS") else range(32)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_chain_matches_oracle(seed, tmp_path):
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from mjmpc_amd.models.mjcf import load_mjcf
    from oracle.physics_ref import RefArm
    xml, nv, nu = chain_xml(seed)
    (tmp_path / "c.xml").write_text(xml)
    raw = load_mjcf(str(tmp_path / "c.xml"), frame_skip=2)
    eng, ref = ArmRolloutEngine(raw, dtype="f64"), RefArm(raw.to_flat())
    rs = np.random.RandomState(1000 + seed)
    tgt = np.asarray(raw.target_pos, float)
    worst = 0.0
    for k in range(24):
        q = raw.qpos0 + rs.uniform(-1, 1, raw.nq) * np.where(eng.model.field("jtype")[:nv] > 0, 0.4, 1.5)
        v = rs.standard_normal(nv) * (0.0 if k % 4 == 0 else (0.03 if k % 4 == 1 else 2.0))
        u = rs.uniform(-1.3, 1.3, nu)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        _, _, r1, o1 = ref.env_step(q, v, u, tgt)
        assert np.isfinite(o1).all()
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("seed %d: nv %d nu %d slides %d lossy %d contact %d: worst relative error of one env step %.2e"
          % (seed, nv, nu, int(eng.model.field("jtype").sum()), int((eng.model.field("frictionloss") > 0).sum()),
             int(eng.model.field("n_sphere")[0]), worst))
    assert worst < 1e-9, worst
    # short rollouts on two launch shapes, and the tree engine on the same inputs
    for P in (40, 4104):
        H = 6
        eps = 0.4 * rs.standard_normal((P, H, nu))
        q = raw.qpos0 + 0.3 * rs.uniform(-1, 1, raw.nq)
        v = 0.5 * rs.standard_normal(nv)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, act, _, _, nobs = eng.rollout(P, H, np.zeros((H, nu)), eps, "open_loop")
        sl = slice(0, 40)
        _, o_rew, o_act, _, o_nobs = ref.rollout(q, v, tgt, np.zeros((H, nu)), eps[sl])
        assert np.array_equal(act[sl], o_act)
        scale = max(1.0, np.abs(o_nobs).max())
        np.testing.assert_allclose(nobs[sl], o_nobs, rtol=1e-8, atol=1e-8 * scale)
        np.testing.assert_allclose(rew[sl], o_rew, rtol=1e-8, atol=1e-8 * scale)
    assert eng.solver_failures() == 0
    if seed % 4 == 0:
        tree = TreeRolloutEngine(raw, dtype="f64")
        tree.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, t_rew, _, _, _, _ = tree.rollout(40, H, np.zeros((H, nu)), eps[:40], "open_loop")
        np.testing.assert_allclose(rew[:40], t_rew, rtol=1e-8, atol=1e-8 * scale)

"""GPU parity on RANDOM models: kinematic trees drawn from a seed - topology, joint kinds (hinge / slide, a ball or a free
root now and then), limits, springs, friction loss, armature, capsule and sphere geoms against a plane with friction,
geom-geom pairs (some with <pair> overrides), boxes (a static slab under the spheres, a box's corners on the plane), affine
actuators with force limits, on joints or on a tendon, joint / connect / weld equalities, a limited tendon, solver parameters
per element -
compiled for the kernel and for the C oracle from the same RawModel and compared over one env step from random states
and over a short rollout.  Whatever instantiation the model asks for (lean / full, 16 or 32 lanes, sparse or dense,
general or not) is what runs.  Tolerances: 1e-9 per env step (SURVEY 8d's gate), 1e-7 over a 6-step rollout (contacts
amplify rounding).  Models the compiler refuses (more than 16 contact records ...) are redrawn."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _quat(rs, scale):
    w = scale * rs.standard_normal(3)
    a = np.linalg.norm(w)
    return np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * w / max(a, 1e-12)])


def random_model(seed):
    from mjmpc_amd.models.raw import (EQ_CONNECT, EQ_JOINT, EQ_WELD, GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, JOINT_BALL, JOINT_FREE,
                                      JOINT_HINGE, JOINT_SLIDE, RawActuator, RawBody, RawEquality, RawGeom, RawJoint, RawModel,
                                      RawPlane, RawTendon)
    rs = np.random.RandomState(seed)
    rs5 = np.random.RandomState(seed + 1000003)     # round 5's attributes (joint margin / ref, geom gap) from a stream of their own:
                                                    # the models of the earlier rounds' seeds keep everything else they had
    deep = seed % 6 == 5                    # every sixth model: long chains (elimination paths beyond 16 links)
    n_bodies = int(rs.randint(18, 27)) if deep else int(rs.randint(3, 24))
    general = rs.rand() < 0.6
    bodies, budget = [], 30
    for i in range(n_bodies):
        # (bodies in depth-first order, as an MJCF file lists them: the parent is the previous body or one of its ancestors -
        # or the world, which starts another tree)
        if i == 0:
            parent = -1
        else:
            chain, k = [], i - 1
            while k >= 0:
                chain.append(k)
                k = bodies[k].parent
            parent = -1 if rs.rand() < (0.03 if deep else 0.12) else int(chain[min(len(chain) - 1, int(rs.exponential(0.15 if deep else 1.0)))])
        kind = JOINT_HINGE
        r = rs.rand()
        if parent < 0 and general and r < 0.3:
            kind = JOINT_FREE
        elif general and r < 0.15 and parent >= 0:
            kind = JOINT_BALL
        elif r < 0.35:
            kind = JOINT_SLIDE
        ndof = {JOINT_HINGE: 1, JOINT_SLIDE: 1, JOINT_BALL: 3, JOINT_FREE: 6}[kind]
        if budget - ndof < 0:
            break
        budget -= ndof
        axis = rs.standard_normal(3)
        axis /= np.linalg.norm(axis)
        limited = kind in (JOINT_HINGE, JOINT_SLIDE) and rs.rand() < 0.6
        lo = -rs.uniform(0.2, 1.0) * (0.2 if kind == JOINT_SLIDE else 1.0)
        hi = rs.uniform(0.2, 1.0) * (0.2 if kind == JOINT_SLIDE else 1.0)
        jt = RawJoint(axis=tuple(axis), range=(lo, hi), limited=limited, damping=float(rs.uniform(0.02, 0.3)),
                      armature=float(rs.uniform(0.0, 0.01)), name="j%d" % i, type=kind,
                      stiffness=float(rs.uniform(0, 2.0)) if (kind in (JOINT_HINGE, JOINT_SLIDE) and rs.rand() < 0.3) else 0.0,
                      springref=float(rs.uniform(-0.2, 0.2)),
                      pos=tuple(0.03 * rs.standard_normal(3)) if (general and kind in (JOINT_HINGE, JOINT_BALL) and rs.rand() < 0.4) else (0.0, 0.0, 0.0),
                      frictionloss=float(rs.uniform(0.01, 0.1)) if (general and kind in (JOINT_HINGE, JOINT_SLIDE) and rs.rand() < 0.25) else 0.0)
        if kind in (JOINT_BALL, JOINT_FREE):            # (axis and spring reference mean nothing there)
            jt.axis, jt.springref = (0.0, 0.0, 1.0), 0.0
        if kind == JOINT_FREE:                          # (<freejoint/>: no damping, no armature, no range)
            jt.range, jt.damping, jt.armature = (0.0, 0.0), 0.0, 0.0
        if kind == JOINT_BALL and not jt.limited:
            jt.range = (0.0, 0.0)
        if kind == JOINT_BALL and rs.rand() < 0.5:
            jt.limited, jt.range = True, (0.0, float(rs.uniform(0.4, 1.0)))
        if jt.limited and rs.rand() < 0.3:
            jt.solref_limit, jt.solimp_limit = (float(rs.uniform(0.01, 0.04)), 1.0), (0.9, 0.95, 0.001, 0.5, 2.0)
        if general and kind in (JOINT_HINGE, JOINT_SLIDE):
            if jt.limited and rs5.rand() < 0.3:
                jt.margin = float(rs5.uniform(0.01, 0.1)) * (0.2 if kind == JOINT_SLIDE else 1.0)
            if rs5.rand() < 0.2:
                jt.ref = float(rs5.uniform(-0.3, 0.3)) * (0.2 if kind == JOINT_SLIDE else 1.0)
        length = rs.uniform(0.08, 0.25)
        d = rs.standard_normal(3)
        d *= length / np.linalg.norm(d)
        rad = float(rs.uniform(0.015, 0.04))
        geoms = [RawGeom(GEOM_CAPSULE, rad, (0.0, 0.0, 0.0), tuple(d), density=float(rs.uniform(500, 1500)), margin=0.002,
                         name="c%d" % i, friction=float(rs.uniform(0.3, 1.0)), condim=3 if rs.rand() < 0.7 else 1)]
        if rs.rand() < 0.3:
            geoms.append(RawGeom(GEOM_SPHERE, 1.2 * rad, tuple(d), density=800.0, margin=0.002, name="s%d" % i,
                                 friction=float(rs.uniform(0.3, 1.0)), condim=3))
        if general and rs5.rand() < 0.2:
            geoms[0].margin, geoms[0].gap = 0.004, float(rs5.uniform(0.0005, 0.003))
        if rs.rand() < 0.25:
            geoms[0].solref, geoms[0].solimp, geoms[0].solmix = (float(rs.uniform(0.01, 0.03)), 1.0), (0.9, 0.95, 0.001, 0.5, 2.0), float(rs.uniform(0.5, 2))
        pos = (float(0.4 * rs.standard_normal()), float(0.4 * rs.standard_normal()), float(rs.uniform(0.3, 0.7))) if parent < 0 \
            else tuple(np.asarray(bodies[parent].geoms[0].b) * rs.uniform(0.5, 1.0))
        quat = tuple(_quat(rs, 0.5)) if rs.rand() < 0.5 else (1.0, 0.0, 0.0, 0.0)
        bodies.append(RawBody("b%d" % i, parent, pos, quat=quat, joint=jt, geoms=geoms))
    # what collides with the plane: a few geoms (each capsule is two records)
    want_box = general and rs.rand() < 0.25             # (a box on the plane: eight records)
    records = 8 if want_box else 0
    order = list(rs.permutation(len(bodies)))
    for i in order:
        for g in bodies[i].geoms:
            cost = 2 if g.type == GEOM_CAPSULE else 1
            if records + cost <= (11 if want_box else 8) and rs.rand() < 0.5:
                g.collide = True
                records += cost
    # geom-geom pairs between bodies that are not parent and child
    pairs = []
    for _ in range(int(rs.randint(0, 4))):
        a, b = rs.choice(len(bodies), 2, replace=False) if len(bodies) > 2 else (0, 1)
        if bodies[a].parent == b or bodies[b].parent == a or a == b:
            continue
        if records + 1 <= 12:
            pairs.append((bodies[max(a, b)].geoms[0].name, bodies[min(a, b)].geoms[0].name))
            records += 1
    # general models: a static box of the world body that the spheres may hit, a box riding on a body (corners against
    # the plane), a connect or a weld between two bodies, a <pair> with its own parameters
    world_geoms, pair_params = [], {}
    spheres = [(i, g) for i, b in enumerate(bodies) for g in b.geoms if g.type == GEOM_SPHERE]
    if general and spheres and rs.rand() < 0.35 and records < 12:
        world_geoms.append(RawGeom(GEOM_BOX, 0.0, (0.3, 0.0, 0.1), (0.3, 0.3, 0.1), name="slab", friction=0.6, condim=3, margin=0.002,
                                   quat=tuple(_quat(rs, 0.2))))
        for i, g in spheres[:2]:
            if records < 13:
                pairs.append((g.name, "slab"))
                records += 1
    # round 5: capsules against the static slab (three records each), a cylinder riding on a body (four records on the plane)
    if general and world_geoms and records + 3 <= 13 and rs5.rand() < 0.6:
        k = int(rs5.randint(len(bodies)))
        pairs.append((bodies[k].geoms[0].name, "slab"))
        records += 3
    if general and records + 4 <= 13 and rs5.rand() < 0.4:
        k = int(rs5.randint(len(bodies)))
        ax = rs5.standard_normal(3)
        ax *= float(rs5.uniform(0.02, 0.06)) / np.linalg.norm(ax)
        bodies[k].geoms.append(RawGeom(GEOM_CYLINDER, float(rs5.uniform(0.02, 0.05)), tuple(-ax), tuple(ax), density=600.0, margin=0.002,
                                       name="cyl%d" % k, friction=0.6, condim=3, collide=True))
        records += 4
        if records + 3 <= 12 and rs5.rand() < 0.8:      # ... and against a static capsule (three records) or sphere (one) beside the tree
            kind = GEOM_CAPSULE if rs5.rand() < 0.6 else GEOM_SPHERE
            world_geoms.append(RawGeom(kind, 0.05, (0.25, -0.2, 0.12), (0.25, 0.2, 0.2), name="bar", friction=0.5, condim=3, margin=0.002))
            pairs.append(("cyl%d" % k, "bar"))
            records += 3 if kind == GEOM_CAPSULE else 1
    elif general and records + 3 <= 12 and rs5.rand() < 0.5:
        # a static cylinder (a post) that the first capsule or sphere of a body may meet
        k = int(rs5.randint(len(bodies)))
        g0 = bodies[k].geoms[0]
        if g0.type in (GEOM_CAPSULE, GEOM_SPHERE) and g0.name:
            world_geoms.append(RawGeom(GEOM_CYLINDER, 0.08, (0.2, 0.1, 0.0), (0.2, 0.1, 0.5), name="post", friction=0.7, condim=3, margin=0.002))
            pairs.append((g0.name, "post"))
            records += 3 if g0.type == GEOM_CAPSULE else 1
    if want_box:
        k = int(rs.randint(len(bodies)))
        bodies[k].geoms.append(RawGeom(GEOM_BOX, 0.0, (0.0, 0.0, 0.0), (0.04, 0.03, 0.02), density=700.0, margin=0.002, name="box%d" % k,
                                       friction=0.5, condim=3, collide=True, quat=tuple(_quat(rs, 0.5))))
        if any(g.name == "slab" for g in world_geoms) and records + 4 <= 16 and rs5.rand() < 0.7:      # round 5: ... and against the static slab (box-box: four records)
            pairs.append(("box%d" % k, "slab"))
            records += 4
    elif general and records + 4 <= 14 and (seed % 5 == 2 or (world_geoms and rs5.rand() < 0.4)):
        if not any(g.name == "slab" for g in world_geoms):
            world_geoms.append(RawGeom(GEOM_BOX, 0.0, (0.3, 0.0, 0.1), (0.3, 0.3, 0.1), name="slab", friction=0.6, condim=3, margin=0.002,
                                       quat=tuple(_quat(rs5, 0.2))))
        k = int(rs5.randint(len(bodies)))
        bodies[k].geoms.append(RawGeom(GEOM_BOX, 0.0, (0.0, 0.0, 0.0), (0.05, 0.03, 0.04), density=700.0, margin=0.002, name="bx%d" % k,
                                       friction=0.5, condim=3, collide=False, quat=tuple(_quat(rs5, 0.5))))
        pairs.append(("bx%d" % k, "slab"))
        records += 4
    if pairs and rs.rand() < 0.4:
        pair_params[tuple(pairs[0])] = dict(condim=int(rs.choice([1, 3])), friction=float(rs.uniform(0.2, 1.2)), margin=0.003,
                                            solref=(float(rs.uniform(0.008, 0.03)), 1.0))
    single = [b.joint.name for b in bodies if b.joint.type in (JOINT_HINGE, JOINT_SLIDE)]
    if not single:
        bodies[-1].joint = RawJoint(axis=(0.0, 1.0, 0.0), range=(-1.0, 1.0), limited=False, damping=0.1, name=bodies[-1].joint.name)
        single = [bodies[-1].joint.name]
    # actuators on a random subset of the hinge / slide joints
    acts = []
    for n in single:
        if rs.rand() < 0.7 or not acts:
            r = rs.rand()
            kw = {}
            if r < 0.3:
                kw = dict(kp=float(rs.uniform(2, 20)))
            elif r < 0.45:
                kw = dict(gainprm=float(rs.uniform(0.5, 3)), biasprm=(float(rs.uniform(-0.1, 0.1)), -float(rs.uniform(0, 5)), -float(rs.uniform(0, 0.3))))
            if rs.rand() < 0.2:
                kw["forcerange"] = (-float(rs.uniform(0.3, 1)), float(rs.uniform(0.3, 1)))
            acts.append(RawActuator(n, float(rs.uniform(0.3, 2.0)), (-1.0, 1.0), ctrllimited=bool(rs.rand() < 0.85), **kw))
    equalities, tendons = [], []
    if general and len(single) >= 2 and rs.rand() < 0.4:
        a, b = rs.choice(len(single), 2, replace=False)
        equalities.append(RawEquality(EQ_JOINT, single[a], single[b], polycoef=(0.0, float(rs.uniform(-1, 1)), 0.0, 0.0, 0.0)))
        records += 1
    if general and len(single) >= 2 and rs.rand() < 0.4 and records < 14:
        a, b = rs.choice(len(single), 2, replace=False)
        tendons.append(RawTendon("t0", [(single[a], 1.0), (single[b], float(rs.uniform(-1, 1)))], limited=True, range=(-0.3, 0.3)))
        records += 1
        if rs.rand() < 0.6:                             # ... with an actuator pulling on it instead of its joints' own
            acts = [a_ for a_ in acts if a_.joint not in (single[a], single[b])]
            acts.append(RawActuator("", float(rs.uniform(0.5, 1.5)), (-1.0, 1.0), kp=float(rs.uniform(2, 10)), tendon="t0"))
    if general and len(bodies) >= 3 and rs.rand() < 0.3 and records + 2 <= 15:
        a, b = rs.choice(len(bodies), 2, replace=False)
        if rs.rand() < 0.6:
            equalities.append(RawEquality(EQ_CONNECT, bodies[max(a, b)].name, bodies[min(a, b)].name if rs.rand() < 0.7 else "",
                                          anchor=tuple(0.05 * rs.standard_normal(3))))
            records += 1
        else:
            equalities.append(RawEquality(EQ_WELD, bodies[max(a, b)].name, bodies[min(a, b)].name if rs.rand() < 0.7 else ""))
            records += 2
    plane = RawPlane(pos=(0.0, 0.0, 0.0), normal=(0.0, 0.0, 1.0), margin=0.002, friction=float(rs.uniform(0.3, 1.0)),
                     condim=3 if rs.rand() < 0.7 else 1)
    # round 5: now and then the floor gives stiffness and damping directly (solref < 0); mixed with a geom's standard set the
    # contact takes the element-wise minimum
    plane_solref = None
    if general and rs5.rand() < 0.2:
        plane_solref = (-float(rs5.uniform(5e3, 5e4)), -float(rs5.uniform(50, 400)))
    # round 5: a third of the general models put their condim-3 contacts under ELLIPTIC cones, impratio 1 or not
    cone, impratio = "pyramidal", 1.0
    if general and rs5.rand() < 0.35:
        cone, impratio = "elliptic", float(rs5.choice([1.0, 1.0, 0.5, 3.0, 10.0]))
    if plane_solref is not None:
        plane.solref = plane_solref
    return RawModel(bodies=bodies, actuators=acts, site_body=len(bodies) - 1, site_pos=(0.05, 0.0, 0.0), target_pos=(0.3, 0.1, 0.4),
                    plane=plane, timestep=0.002, frame_skip=2, gravity=(0.0, 0.0, -9.81), pairs=pairs, equalities=equalities,
                    tendons=tendons, world_geoms=world_geoms, pair_params=pair_params, cone=cone, impratio=impratio)


def random_state(raw, rs):
    from mjmpc_amd.models.raw import JOINT_BALL, JOINT_FREE
    q, v = raw.qpos0.copy(), rs.standard_normal(raw.nv)
    adr = 0
    for b in raw.bodies:
        jt = b.joint
        if jt.type == JOINT_FREE:
            q[adr:adr + 3] += 0.05 * rs.standard_normal(3)
            q[adr + 2] = rs.uniform(0.05, 0.6)
            q[adr + 3:adr + 7] = _quat(rs, 1.0)
        elif jt.type == JOINT_BALL:
            q[adr:adr + 4] = _quat(rs, 0.8)
        else:
            q[adr] = jt.ref + rs.uniform(-1.2, 1.2) * (0.25 if jt.type == 2 else 1.0)
        adr += jt.nq
    return q, v


def _assert_rollouts_close(got, want, scale_from=None):
    """Several env steps in a row: 1e-7 of the particle's largest observation (at least 1e-7 absolute).  The one-step checks
    above hold 1e-9; over a rollout a random model with stiff rows can amplify rounding by many orders - soak seeds 15007 and
    27612 (of 24 000) reach velocities of 1e6 and 6e2 rad/s, the oracle's OWN next observation moves by 1e-2 and 6e-6 there when
    its state is perturbed by 1e-12, and kernel and oracle differ by 2e-10 and 6e-9 of the values (5e-4 and 4e-6 absolute)."""
    ref = want if scale_from is None else scale_from
    scale = np.maximum(1.0, np.abs(ref).reshape(len(ref), -1).max(axis=1))
    err = np.abs(got - want).reshape(len(want), -1).max(axis=1)
    assert (err <= 1e-7 * scale).all(), (err / scale).max()


def _beyond_one_step_tolerance(got, want):
    """How many rollouts of a multi-step comparison needed the scaled tolerance at all: particles whose worst entry differs by
    more than the one-step bound (1e-9 of the particle's largest observation) - the bookkeeping VERDICT r5 asked for."""
    scale = np.maximum(1.0, np.abs(want).reshape(len(want), -1).max(axis=1))
    err = np.abs(got - want).reshape(len(want), -1).max(axis=1)
    return int((err > 1e-9 * scale).sum())


MAX_REDRAWS = 5     # models drawn again per seed because engine or oracle refuses them; more is a failure

# (MJMPC_FUZZ_SEEDS=a:b in the environment runs another range of seeds - a soak run, not part of the suite)
_SEEDS = range(*[int(x) for x in os.environ["MJMPC_FUZZ_SEEDS"].split(":")]) if os.environ.get("MJMPC_FUZZ_SEEDS") else range(48)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_model_matches_oracle(seed):
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw, eng, tries, refusals = None, None, 0, []
    while eng is None:
        raw = random_model(1000 * tries + seed)
        try:
            eng = TreeRolloutEngine(raw, dtype="f64")
            ref = RefArm(raw.to_flat())
        except (ValueError, NotImplementedError, AssertionError) as e:
            # the generator drew a model engine or oracle refuses (DESIGN 7's list): drawn again, COUNTED and bounded
            refusals.append("%s: %s" % (type(e).__name__, str(e)[:80]))
            eng, tries = None, tries + 1
            assert tries <= MAX_REDRAWS, "seed %d: %d models in a row refused: %s" % (seed, tries, refusals)
    stats = dict(seed=seed, redraws=tries, refusals=refusals, scaled_tolerance_hits=0, nonfinite_oracle=0)
    m = eng.model
    rs = np.random.RandomState(seed + 77)
    tgt = np.asarray(raw.target_pos, float)
    A = eng.d_action
    worst = 0.0
    for k in range(12):
        q, v = random_state(raw, rs)
        v *= (0.0 if k % 4 == 0 else 1.0)
        u = rs.uniform(-1.5, 1.5, A)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        # (the reset emulation keeps the oracle finite: a non-finite observation here is a finding, not a case to skip)
        assert np.isfinite(o1).all() and np.isfinite(r1), "seed %d state %d: the oracle left a non-finite step" % (seed, k)
        worst = max(worst, np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()), abs(rew[0, 0] - r1) / max(1.0, abs(r1)))
    print("seed %d (%d re-draws): nv %d nq %d, %d bodies, %d records, general %s, max path %d: worst relative error %.2e"
          % (seed, tries, m.nv, raw.nq, len(raw.bodies), int(m.field("n_sphere")[0]), m.general, m.max_path, worst))
    assert worst < 1e-9, worst
    P, H = 32, 6
    q, v = random_state(raw, rs)
    eps = 0.5 * rs.standard_normal((P, H, A))
    eng.set_env_state(dict(qp=q, qv=0.3 * v, target_pos=tgt))
    obs, rew, act, done, info, nobs = eng.rollout(P, H, np.zeros((H, A)), eps, "open_loop")
    o_obs, o_rew, _, _, o_nobs = ref.rollout(q, 0.3 * v, tgt, np.zeros((H, A)), eps)
    ok = np.isfinite(o_nobs).all(axis=(1, 2))
    stats["nonfinite_oracle"] += int((~ok).sum())
    assert ok.all(), "seed %d: %d oracle rollouts turned non-finite" % (seed, int((~ok).sum()))
    stats["scaled_tolerance_hits"] += _beyond_one_step_tolerance(nobs, o_nobs)
    _assert_rollouts_close(nobs[ok], o_nobs[ok])
    np.testing.assert_allclose(rew[ok], o_rew[ok], rtol=1e-7, atol=1e-7)
    # the same start in mode="closed_loop_linear" (gym_env_wrapper.py:135-136): actions from the observation each step starts from
    W = 0.05 * rs.standard_normal((eng.d_obs + 1, A))
    obs, rew, act, done, info, nobs = eng.rollout(8, 4, W, eps[:8, :4], "closed_loop_linear")
    o_obs, o_rew, o_act, _, o_nobs = ref.rollout(q, 0.3 * v, tgt, W, eps[:8, :4], mode="closed_loop_linear")
    ok = np.isfinite(o_nobs).all(axis=(1, 2))
    stats["nonfinite_oracle"] += int((~ok).sum())
    assert ok.all(), "seed %d: %d closed-loop oracle rollouts turned non-finite" % (seed, int((~ok).sum()))
    stats["scaled_tolerance_hits"] += _beyond_one_step_tolerance(nobs, o_nobs)
    _assert_rollouts_close(act[ok], o_act[ok], o_nobs[ok])
    _assert_rollouts_close(nobs[ok], o_nobs[ok])
    if os.environ.get("MJMPC_FUZZ_STATS"):      # soak runs: one JSON line per seed (tools/soak_summary.py)
        import json
        with open(os.environ["MJMPC_FUZZ_STATS"], "a") as f:
            f.write(json.dumps(stats) + "\n")


@pytest.mark.parametrize("seed", range(0, 48, 4))
def test_random_model_f32_stays_close(seed):
    """The same models in single precision: one env step from random states against the f64 oracle - the median error of
    the next observation stays at single-precision level (1e-4 relative stated - velocities of 1e1 rad/s after stiff equality
    rows; measured medians 2e-6 ... 4e-5; contacts amplify it in a few states,
    which the 95th percentile bound of 1e-2 allows for), nothing turns non-finite where the oracle is finite."""
    from mjmpc_amd.envs.tree_engine import TreeRolloutEngine
    from oracle.physics_ref import RefArm
    raw, eng, tries = None, None, 0
    while eng is None:
        raw = random_model(1000 * tries + seed)
        try:
            eng = TreeRolloutEngine(raw, dtype="f32")
            ref = RefArm(raw.to_flat())
        except (ValueError, NotImplementedError, AssertionError):
            eng, tries = None, tries + 1
            assert tries <= MAX_REDRAWS
    rs = np.random.RandomState(seed + 99)
    tgt = np.asarray(raw.target_pos, float)
    errs = []
    for k in range(16):
        q, v = random_state(raw, rs)
        u = rs.uniform(-1.0, 1.0, eng.d_action)
        eng.set_env_state(dict(qp=q, qv=v, target_pos=tgt))
        _, rew, _, _, _, nobs = eng.rollout(1, 1, u[None], None, "open_loop")
        q1, v1, r1, o1 = ref.env_step(q, v, u, tgt)
        if np.isfinite(o1).all():
            assert np.isfinite(nobs[0, 0]).all()
            errs.append(np.abs(nobs[0, 0] - o1).max() / max(1.0, np.abs(o1).max()))
    errs = np.sort(errs)
    assert np.median(errs) < 1e-4 and errs[int(0.95 * (len(errs) - 1))] < 1e-2, errs

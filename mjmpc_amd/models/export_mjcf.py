"""RawModel -> MJCF text: what ``load_mjcf`` reads back to the same tables (tests/test_export_mjcf_cpu.py holds the round
trip on random models), and what MuJoCo itself can load - the way to put this repository's physics beside the real thing
wherever MuJoCo is installed (it is not in this image: tools/pin_with_mujoco.py).

Conventions: angles in radians, every geom masked out of MuJoCo's derived collisions except against the plane (plane
contype 0 / conaffinity 1, colliding geoms contype 1 / conaffinity 0 - no geom-geom pair is derived from masks), geom-geom
candidates as explicit ``<pair>`` elements, inertia from the geoms unless the body carries an ``<inertial>``.  A chain of
massless single-joint bodies (what the loader makes of a body with several joints) is written back as one body with
several joints."""
import numpy as np

from .raw import (EQ_CONNECT, EQ_JOINT, EQ_WELD, GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, JOINT_BALL, JOINT_FREE, JOINT_HINGE, JOINT_SLIDE,
                  RawModel)


def _f(x):
    return "%.17g" % float(x)


def _v(xs):
    return " ".join(_f(x) for x in xs)


def _sol(prefix, solref, solimp):
    out = ""
    if solref is not None:
        out += ' %s="%s"' % (prefix[0], _v(solref))
    if solimp is not None:
        out += ' %s="%s"' % (prefix[1], _v(solimp))
    return out


def _geom_xml(g, plane_collide):
    t = {GEOM_SPHERE: "sphere", GEOM_CAPSULE: "capsule", GEOM_BOX: "box", GEOM_CYLINDER: "cylinder"}[g.type]
    a = ' name="%s" type="%s"' % (g.name, t) if g.name else ' type="%s"' % t
    if g.type == GEOM_SPHERE:
        a += ' size="%s" pos="%s"' % (_f(g.radius), _v(g.a))
    elif g.type in (GEOM_CAPSULE, GEOM_CYLINDER):
        a += ' size="%s" fromto="%s %s"' % (_f(g.radius), _v(g.a), _v(g.b))
    else:
        a += ' size="%s" pos="%s" quat="%s"' % (_v(g.b), _v(g.a), _v(g.quat))
    a += ' density="%s" margin="%s" friction="%s 0.005 0.0001" condim="%d"' % (_f(g.density), _f(g.margin), _f(g.friction), int(g.condim))
    if g.gap:
        a += ' gap="%s"' % _f(g.gap)
    a += ' contype="%d" conaffinity="0"' % (1 if (plane_collide and g.collide) else 0)
    a += _sol(("solref", "solimp"), g.solref, g.solimp)
    if g.solmix != 1.0:
        a += ' solmix="%s"' % _f(g.solmix)
    if g.priority != 0:
        a += ' priority="%d"' % int(g.priority)
    return "<geom%s/>" % a


def _joint_xml(j):
    if j.type == JOINT_FREE:
        return '<freejoint name="%s"/>' % j.name
    t = {JOINT_HINGE: "hinge", JOINT_SLIDE: "slide", JOINT_BALL: "ball"}[j.type]
    a = ' name="%s" type="%s"' % (j.name, t)
    if j.type != JOINT_BALL:
        a += ' axis="%s"' % _v(j.axis)
    if j.type in (JOINT_HINGE, JOINT_BALL):
        a += ' pos="%s"' % _v(j.pos)
    a += ' limited="%s" range="%s" damping="%s" armature="%s"' % ("true" if j.limited else "false", _v(j.range), _f(j.damping), _f(j.armature))
    if j.type in (JOINT_HINGE, JOINT_SLIDE):
        a += ' stiffness="%s" springref="%s"' % (_f(j.stiffness), _f(j.springref))
        if j.margin or j.ref:
            a += ' margin="%s" ref="%s"' % (_f(j.margin), _f(j.ref))
    if j.frictionloss:
        a += ' frictionloss="%s"' % _f(j.frictionloss)
    a += _sol(("solreflimit", "solimplimit"), j.solref_limit, j.solimp_limit)
    a += _sol(("solreffriction", "solimpfriction"), j.solref_friction, j.solimp_friction)
    return "<joint%s/>" % a


def to_mjcf(raw: RawModel, hand_site="finger", target_site="target") -> str:
    lines = ['<mujoco model="mjmpc_amd_export">',
             '  <compiler angle="radian" coordinate="local" inertiafromgeom="auto"/>',
             '  <option timestep="%s" gravity="%s" density="%s" viscosity="%s" integrator="Euler" cone="%s" impratio="%s"/>'
             % (_f(raw.timestep), _v(raw.gravity), _f(raw.density), _f(raw.viscosity), raw.cone, _f(raw.impratio)),
             # the model's own sets as the defaults every element without its own falls back to
             '  <default><geom solref="%s" solimp="%s"/><joint solreflimit="%s" solimplimit="%s" solreffriction="%s" solimpfriction="%s"/></default>'
             % (_v(raw.solref), _v(raw.solimp), _v(raw.solref if raw.solref_limit is None else raw.solref_limit),
                _v(raw.solimp if raw.solimp_limit is None else raw.solimp_limit), _v(raw.solref_friction), _v(raw.solimp_friction)),
             "  <worldbody>"]
    if raw.plane is not None:
        p = raw.plane
        n = np.asarray(p.normal, float) / np.linalg.norm(p.normal)
        # a frame whose z axis is the normal
        z = np.array([0.0, 0.0, 1.0])
        ax = np.cross(z, n)
        s, c = np.linalg.norm(ax), float(z @ n)
        quat = (1.0, 0.0, 0.0, 0.0) if s < 1e-15 and c > 0 else ((0.0, 1.0, 0.0, 0.0) if s < 1e-15 else
                                                                  tuple(np.r_[np.cos(np.arctan2(s, c) / 2), np.sin(np.arctan2(s, c) / 2) * ax / s]))
        lines.append('    <geom name="floor" type="plane" size="5 5 0.1" pos="%s" quat="%s" margin="%s" gap="%s" friction="%s 0.005 0.0001" condim="%d" '
                     'contype="0" conaffinity="1"%s%s%s/>'
                     % (_v(p.pos), _v(quat), _f(p.margin), _f(p.gap), _f(p.friction), int(p.condim), _sol(("solref", "solimp"), p.solref, p.solimp),
                        ' solmix="%s"' % _f(p.solmix) if p.solmix != 1.0 else "", ' priority="%d"' % p.priority if p.priority else ""))
    lines.append('    <site name="%s" pos="%s"/>' % (target_site, _v(raw.target_pos)))
    for g in raw.world_geoms:
        lines.append("    " + _geom_xml(g, False))
    children = {}
    for i, b in enumerate(raw.bodies):
        children.setdefault(b.parent, []).append(i)

    def massless_link(i):
        b = raw.bodies[i]
        kids = children.get(i, [])
        return (not b.geoms and b.inertial is None and len(kids) == 1 and b.joint is not None and b.joint.type in (JOINT_HINGE, JOINT_SLIDE)
                and np.allclose(raw.bodies[kids[0]].pos, 0) and np.allclose(raw.bodies[kids[0]].quat, (1, 0, 0, 0))
                and raw.bodies[kids[0]].joint is not None and raw.site_body != i)

    def emit(i, indent, joints_above=(), pose=None):
        b = raw.bodies[i]
        pos, quat = pose if pose is not None else (b.pos, b.quat)
        if massless_link(i):                    # its joint joins the child's
            emit(children[i][0], indent, tuple(joints_above) + (b.joint,), (pos, quat))
            return
        pad = " " * indent
        lines.append('%s<body name="%s" pos="%s" quat="%s">' % (pad, b.name, _v(pos), _v(quat)))
        for j in tuple(joints_above) + ((b.joint,) if b.joint is not None else ()):
            lines.append(pad + "  " + _joint_xml(j))
        if b.inertial is not None:
            I = np.asarray(b.inertial.inertia, float)
            lines.append('%s  <inertial pos="%s" mass="%s" fullinertia="%s"/>'
                         % (pad, _v(b.inertial.pos), _f(b.inertial.mass), _v([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])))
        for g in b.geoms:
            lines.append(pad + "  " + _geom_xml(g, raw.plane is not None))
        if i == raw.site_body:
            lines.append('%s  <site name="%s" pos="%s"/>' % (pad, hand_site, _v(raw.site_pos)))
        for c in children.get(i, []):
            emit(c, indent + 2)
        lines.append("%s</body>" % pad)

    for r in children.get(-1, []):
        emit(r, 4)
    lines.append("  </worldbody>")
    if raw.pairs:
        lines.append("  <contact>")
        for ga, gb in raw.pairs:
            o = raw.pair_params.get((ga, gb), raw.pair_params.get((gb, ga), {}))
            a = ""
            if "condim" in o:
                a += ' condim="%d"' % int(o["condim"])
            if "friction" in o:
                a += ' friction="%s %s 0.005 0.0001 0.0001"' % (_f(o["friction"]), _f(o["friction"]))
            if "margin" in o:
                a += ' margin="%s"' % _f(o["margin"])
            a += _sol(("solref", "solimp"), o.get("solref"), o.get("solimp"))
            lines.append('    <pair geom1="%s" geom2="%s"%s/>' % (ga, gb, a))
        lines.append("  </contact>")
    if raw.equalities:
        lines.append("  <equality>")
        for e in raw.equalities:
            sol = _sol(("solref", "solimp"), e.solref, e.solimp)
            if e.type == EQ_CONNECT:
                lines.append('    <connect body1="%s"%s anchor="%s"%s/>' % (e.obj1, ' body2="%s"' % e.obj2 if e.obj2 else "", _v(e.anchor), sol))
            elif e.type == EQ_WELD:
                lines.append('    <weld body1="%s"%s%s/>' % (e.obj1, ' body2="%s"' % e.obj2 if e.obj2 else "", sol))
            elif e.type == EQ_JOINT:
                lines.append('    <joint joint1="%s"%s polycoef="%s"%s/>' % (e.obj1, ' joint2="%s"' % e.obj2 if e.obj2 else "", _v(e.polycoef), sol))
        lines.append("  </equality>")
    if raw.tendons:
        lines.append("  <tendon>")
        for t in raw.tendons:
            lines.append('    <fixed name="%s" limited="%s" range="%s" margin="%s"%s>' % (
                t.name, "true" if t.limited else "false", _v(t.range), _f(t.margin), _sol(("solreflimit", "solimplimit"), t.solref_limit, t.solimp_limit)))
            for jn, c in t.joints:
                lines.append('      <joint joint="%s" coef="%s"/>' % (jn, _f(c)))
            lines.append("    </fixed>")
        lines.append("  </tendon>")
    lines.append("  <actuator>")
    for a in raw.actuators:
        where = 'tendon="%s"' % a.tendon if a.tendon else 'joint="%s"' % a.joint
        b0, b1, b2 = a.bias
        extra = ' ctrllimited="%s" ctrlrange="%s" gear="%s"' % ("true" if a.ctrllimited else "false", _v(a.ctrlrange), _f(a.gear))
        if a.forcerange is not None:
            extra += ' forcelimited="true" forcerange="%s"' % _v(a.forcerange)
        if a.gainprm is None and a.kp > 0:
            lines.append('    <position %s kp="%s"%s/>' % (where, _f(a.kp), extra))
        elif a.gainprm is None:
            lines.append('    <motor %s%s/>' % (where, extra))
        else:
            lines.append('    <general %s gainprm="%s" biastype="affine" biasprm="%s"%s/>' % (where, _f(a.gain), _v((b0, b1, b2)), extra))
    lines.append("  </actuator>")
    lines.append("</mujoco>")
    return "\n".join(lines) + "\n"

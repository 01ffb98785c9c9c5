"""MJCF-subset loader: the part of MuJoCo's XML the arm kernel can execute.

Supported: `<compiler angle="radian" coordinate="local" inertiafromgeom="true">`, `<option timestep
gravity>`, `<default>` for joint (armature, damping, limited) and geom (margin, contype, conaffinity,
density), nested `<body pos quat>` with at most one hinge `<joint>` each, sphere / capsule(fromto)
geoms, one world `<geom type="plane">`, world and body `<site>`s, `<motor joint gear ctrlrange>`.
Anything else raises ValueError, so that a model is never silently simulated wrongly.
"""
import xml.etree.ElementTree as ET

import numpy as np

from .raw import (GEOM_CAPSULE, GEOM_SPHERE, RawActuator, RawBody, RawGeom, RawJoint, RawModel, RawPlane)


def _floats(s, n=None, default=None):
    if s is None:
        return default
    v = [float(x) for x in s.split()]
    if n is not None and len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


def load_mjcf(path, hand_site="finger", target_site="target", frame_skip=2):
    root = ET.parse(path).getroot()
    comp = root.find("compiler")
    if comp is not None:
        if comp.get("angle", "degree") != "radian" or comp.get("coordinate", "local") != "local":
            raise ValueError("only angle='radian', coordinate='local' are supported")
        if comp.get("inertiafromgeom", "auto") not in ("true", "auto"):
            raise ValueError("inertiafromgeom must be true")
    opt = root.find("option")
    timestep = float(opt.get("timestep", "0.002")) if opt is not None else 0.002
    gravity = _floats(opt.get("gravity") if opt is not None else None, 3, [0.0, 0.0, -9.81])
    if opt is not None and opt.get("integrator", "Euler") != "Euler":
        raise ValueError("only the Euler integrator is supported")
    dj, dg = {}, {}
    dflt = root.find("default")
    if dflt is not None:
        if dflt.find("default") is not None:
            raise ValueError("nested default classes are not supported")
        dj = dict(dflt.find("joint").attrib) if dflt.find("joint") is not None else {}
        dg = dict(dflt.find("geom").attrib) if dflt.find("geom") is not None else {}

    def g_attr(e, k, fallback):
        return e.get(k, dg.get(k, fallback))

    def j_attr(e, k, fallback):
        return e.get(k, dj.get(k, fallback))

    bodies, sites, plane = [], {}, None
    world = root.find("worldbody")

    def collides(e):
        return int(g_attr(e, "contype", "1")) != 0 and int(g_attr(e, "conaffinity", "1")) != 0

    def parse_geom(e):
        t = e.get("type", "sphere")
        size = _floats(e.get("size"))
        margin = float(g_attr(e, "margin", "0"))
        density = float(g_attr(e, "density", "1000"))
        if t == "sphere":
            return RawGeom(GEOM_SPHERE, size[0], _floats(e.get("pos"), 3, [0.0, 0.0, 0.0]), density=density,
                           collide=collides(e), margin=margin, name=e.get("name", ""))
        if t == "capsule":
            ft = _floats(e.get("fromto"), 6)
            if ft is None:
                raise ValueError("capsules need fromto")
            return RawGeom(GEOM_CAPSULE, size[0], ft[:3], ft[3:], density=density, collide=collides(e), margin=margin,
                           name=e.get("name", ""))
        raise ValueError("unsupported geom type %r" % t)

    for e in world:
        if e.tag == "geom":
            if e.get("type") != "plane":
                raise ValueError("only a plane may be attached to the world body")
            if collides(e):
                q = _floats(e.get("quat"), 4, [1.0, 0.0, 0.0, 0.0])
                if q != [1.0, 0.0, 0.0, 0.0]:
                    raise ValueError("rotated planes are not supported")
                plane = RawPlane(pos=_floats(e.get("pos"), 3, [0.0, 0.0, 0.0]), normal=(0.0, 0.0, 1.0),
                                 margin=float(g_attr(e, "margin", "0")))
        elif e.tag == "site":
            sites[e.get("name")] = (-1, _floats(e.get("pos"), 3, [0.0, 0.0, 0.0]))
        elif e.tag not in ("body", "light", "camera"):
            raise ValueError("unsupported worldbody element <%s>" % e.tag)

    def walk(e, parent):
        idx = len(bodies)
        joints = e.findall("joint")
        if len(joints) > 1:
            raise ValueError("body %s: at most one joint per body" % e.get("name"))
        joint = None
        if joints:
            j = joints[0]
            if j.get("type", "hinge") != "hinge":
                raise ValueError("only hinge joints are supported")
            if _floats(j.get("pos"), 3, [0.0, 0.0, 0.0]) != [0.0, 0.0, 0.0]:
                raise ValueError("joint anchors must be at the body origin")
            joint = RawJoint(axis=_floats(j.get("axis"), 3, [0.0, 0.0, 1.0]), range=_floats(j.get("range"), 2, [0.0, 0.0]),
                             limited=j_attr(j, "limited", "false") == "true", damping=float(j_attr(j, "damping", "0")),
                             armature=float(j_attr(j, "armature", "0")), name=j.get("name", ""))
        b = RawBody(e.get("name", "body%d" % idx), parent, _floats(e.get("pos"), 3, [0.0, 0.0, 0.0]),
                    _floats(e.get("quat"), 4, [1.0, 0.0, 0.0, 0.0]), joint, [parse_geom(g) for g in e.findall("geom")])
        bodies.append(b)
        for s in e.findall("site"):
            sites[s.get("name")] = (idx, _floats(s.get("pos"), 3, [0.0, 0.0, 0.0]))
        for c in e.findall("body"):
            walk(c, idx)
        for c in e:
            if c.tag not in ("joint", "geom", "site", "body"):
                raise ValueError("unsupported body element <%s>" % c.tag)

    for e in world.findall("body"):
        walk(e, -1)
    acts = []
    act = root.find("actuator")
    for m in (list(act) if act is not None else []):
        if m.tag != "motor" or m.get("ctrllimited", "false") != "true":
            raise ValueError("only ctrllimited <motor> actuators are supported")
        gear = _floats(m.get("gear"), None, [1.0])[0]
        acts.append(RawActuator(m.get("joint"), gear, _floats(m.get("ctrlrange"), 2)))
    if hand_site not in sites or sites[hand_site][0] < 0:
        raise ValueError("tracked site %r must be attached to a body" % hand_site)
    target = sites.get(target_site, (-1, [0.0, 0.0, 0.0]))[1]
    return RawModel(bodies=bodies, actuators=acts, site_body=sites[hand_site][0], site_pos=sites[hand_site][1],
                    target_pos=target, plane=plane, timestep=timestep, frame_skip=frame_skip, gravity=gravity)

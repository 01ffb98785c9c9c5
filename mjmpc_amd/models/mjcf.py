"""MJCF-subset loader: the part of MuJoCo's XML the rollout kernels can execute.

Enough for the three models the reference vendors (mjmpc/envs/assets/xml/sawyer.xml, swimmer.xml, half_cheetah.xml):

* ``<compiler angle="radian" coordinate="local" inertiafromgeom="true" settotalmass>``,
  ``<option timestep gravity density viscosity integrator="Euler">``;
* ``<default>`` with nested classes, ``class=`` / ``childclass=`` (joint, geom and motor attributes);
* nested ``<body pos quat>`` with any number of hinge / slide ``<joint>``s anchored at the body origin - a body with
  several joints becomes a chain of massless bodies, one joint each, which is what MuJoCo's kinematics does with it;
  joint ``axis range limited damping armature stiffness springref``, ``solreflimit`` / ``solimplimit`` (one set per model);
* sphere and capsule geoms (``fromto``, or ``size pos`` with ``quat`` / ``axisangle``), ``density``, ``margin``,
  ``friction``, ``condim``, ``contype`` / ``conaffinity`` (what collides is decided against the one world plane,
  with MuJoCo's rule), ``solref`` / ``solimp`` (one set per model);
* one world ``<geom type="plane">``, world and body ``<site>``s, ``<motor joint gear ctrlrange ctrllimited>`` and
  ``<position joint kp ctrlrange ctrllimited>`` actuators, ``<contact><pair geom1 geom2>`` (sphere / capsule geoms of a
  manipulator against those of one free object).

Anything that would change the simulation and is not modelled raises ValueError, so that a model is never silently
simulated wrongly; purely visual elements (asset, light, camera, material, rgba ...) are skipped.
"""
import xml.etree.ElementTree as ET

import numpy as np

from .compile import _geom_inertial
from .raw import (GEOM_CAPSULE, GEOM_SPHERE, JOINT_HINGE, JOINT_SLIDE, MJ20_CAPSULE_CAP, TASK_FORWARD, TASK_REACH, RawActuator,
                  RawBody, RawGeom, RawJoint, RawModel, RawPlane)

_VISUAL_BODY_TAGS = ("light", "camera")
_TOP_TAGS = ("compiler", "option", "default", "worldbody", "actuator", "asset", "size", "visual", "statistic", "custom")


def _floats(s, n=None, default=None):
    if s is None:
        return default
    v = [float(x) for x in s.split()]
    if n is not None and len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


def _axisangle_z(v):
    """Direction the local z axis takes under MuJoCo's axisangle="x y z a"."""
    ax, ang = np.asarray(v[:3], float), v[3]
    ax = ax / np.linalg.norm(ax)
    z = np.array([0.0, 0.0, 1.0])
    return z * np.cos(ang) + np.cross(ax, z) * np.sin(ang) + ax * (ax @ z) * (1 - np.cos(ang))


def _quat_z(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])


class _Defaults:
    """``<default>`` classes: attributes of joint / geom / motor, inherited from the enclosing class."""

    def __init__(self, root):
        self.cls = {"main": {"joint": {}, "geom": {}, "motor": {}}}
        top = root.find("default")
        if top is not None:
            self._read(top, "main", None)

    def _read(self, node, name, parent):
        base = {k: dict(v) for k, v in self.cls[parent].items()} if parent is not None else {"joint": {}, "geom": {}, "motor": {}}
        for tag in ("joint", "geom", "motor"):
            e = node.find(tag)
            if e is not None:
                base[tag].update(e.attrib)
        self.cls[name] = base
        for child in node.findall("default"):
            cname = child.get("class")
            if not cname:
                raise ValueError("a nested <default> needs a class name")
            self._read(child, cname, name)

    def attr(self, tag, elem, active, key, fallback=None):
        if key in elem.attrib:
            return elem.get(key)
        c = elem.get("class", active or "main")
        if c not in self.cls:
            raise ValueError("unknown default class %r" % c)
        return self.cls[c][tag].get(key, fallback)


def load_mjcf(path, hand_site="finger", target_site="target", frame_skip=2, task=TASK_REACH, ctrl_cost=0.0, obs_skip=0,
              self_collision=True):
    """``task=TASK_FORWARD`` (with ``ctrl_cost`` / ``obs_skip``) loads a locomotion model: no tracked site is needed.
    ``self_collision=False`` leaves out the geom-geom pairs MuJoCo would derive from contype / conaffinity."""
    root = ET.parse(path).getroot()
    for e in root:
        if e.tag == "contact" and all(c.tag == "pair" for c in e):
            continue                    # empty (sawyer.xml:94-99 holds comments only) or <pair geom1 geom2> entries, read below
        if e.tag not in _TOP_TAGS:
            raise ValueError("unsupported element <%s>" % e.tag)
    comp = root.find("compiler")
    totalmass = None
    if comp is not None:
        if comp.get("angle", "degree") != "radian" or comp.get("coordinate", "local") != "local":
            raise ValueError("only angle='radian', coordinate='local' are supported")
        if comp.get("inertiafromgeom", "auto") not in ("true", "auto"):
            raise ValueError("inertiafromgeom must be true")
        if comp.get("settotalmass") is not None:
            totalmass = float(comp.get("settotalmass"))
    else:
        raise ValueError("angle='radian' must be set (MuJoCo's default is degrees)")
    opt = root.find("option")
    oget = (lambda k, d: opt.get(k, d)) if opt is not None else (lambda k, d: d)
    timestep = float(oget("timestep", "0.002"))
    gravity = _floats(oget("gravity", None), 3, [0.0, 0.0, -9.81])
    if oget("integrator", "Euler") != "Euler":
        raise ValueError("only the Euler integrator is supported")
    if oget("cone", "pyramidal") != "pyramidal" or float(oget("impratio", "1")) != 1.0:
        raise ValueError("only pyramidal friction cones with impratio 1 are supported")
    if _floats(oget("wind", None), 3, [0.0, 0.0, 0.0]) != [0.0, 0.0, 0.0]:
        raise ValueError("wind is not supported")
    density, viscosity = float(oget("density", "0")), float(oget("viscosity", "0"))
    dfl = _Defaults(root)

    bodies, sites = [], {}
    plane_elem = None
    world = root.find("worldbody")
    geom_solver, limit_solver = set(), set()

    def parse_geom(e, active):
        ga = lambda k, d=None: dfl.attr("geom", e, active, k, d)        # noqa: E731
        t = ga("type", "sphere")
        size = _floats(ga("size"))
        common = dict(density=float(ga("density", "1000")), margin=float(ga("margin", "0")),
                      friction=_floats(ga("friction", "1 0.005 0.0001"))[0], condim=int(ga("condim", "3")),
                      name=e.get("name", ""))
        if float(ga("gap", "0")) != 0.0:
            raise ValueError("geom gap is not supported")
        pos = np.asarray(_floats(e.get("pos"), 3, [0.0, 0.0, 0.0]))
        if t == "sphere":
            g = RawGeom(GEOM_SPHERE, size[0], tuple(pos), **common)
        elif t == "capsule":
            ft = _floats(e.get("fromto"), 6)
            if ft is not None:
                a, b = ft[:3], ft[3:]
            else:
                if e.get("quat") is not None:
                    z = _quat_z(_floats(e.get("quat"), 4))
                elif e.get("axisangle") is not None:
                    z = _axisangle_z(_floats(e.get("axisangle"), 4))
                elif e.get("euler") is not None or e.get("zaxis") is not None or e.get("xyaxes") is not None:
                    raise ValueError("geom orientation must be fromto, quat or axisangle")
                else:
                    z = np.array([0.0, 0.0, 1.0])
                a, b = tuple(pos - size[1] * z), tuple(pos + size[1] * z)
            g = RawGeom(GEOM_CAPSULE, size[0], tuple(a), tuple(b), **common)
        else:
            raise ValueError("unsupported geom type %r" % t)
        g._contype, g._conaffinity = int(ga("contype", "1")), int(ga("conaffinity", "1"))
        g._solver = (tuple(_floats(ga("solref", "0.02 1"))), tuple(_floats(ga("solimp", "0.9 0.95 0.001 0.5 2"))))
        return g

    for e in world:
        if e.tag == "geom":
            if dfl.attr("geom", e, None, "type", "sphere") != "plane":
                raise ValueError("only a plane may be attached to the world body")
            if plane_elem is not None:
                raise ValueError("one world plane at most")
            plane_elem = e
        elif e.tag == "site":
            sites[e.get("name")] = (-1, _floats(e.get("pos"), 3, [0.0, 0.0, 0.0]))
        elif e.tag != "body" and e.tag not in _VISUAL_BODY_TAGS:
            raise ValueError("unsupported worldbody element <%s>" % e.tag)

    def parse_joint(j, active):
        ja = lambda k, d=None: dfl.attr("joint", j, active, k, d)       # noqa: E731
        t = ja("type", "hinge")
        if t not in ("hinge", "slide"):
            raise ValueError("only hinge and slide joints are supported, got %r" % t)
        if _floats(j.get("pos"), 3, [0.0, 0.0, 0.0]) != [0.0, 0.0, 0.0]:
            raise ValueError("joint anchors must be at the body origin")
        if float(ja("frictionloss", "0")) != 0.0 or float(ja("margin", "0")) != 0.0:
            raise ValueError("joint frictionloss / margin are not supported")
        limited = ja("limited", "false") == "true"
        if limited:
            limit_solver.add((tuple(_floats(ja("solreflimit", "0.02 1"))),
                              tuple(_floats(ja("solimplimit", "0.9 0.95 0.001 0.5 2")))))
        return RawJoint(axis=_floats(ja("axis"), 3, [0.0, 0.0, 1.0]), range=_floats(ja("range"), 2, [0.0, 0.0]),
                        limited=limited, damping=float(ja("damping", "0")), armature=float(ja("armature", "0")),
                        name=j.get("name", ""), type=JOINT_SLIDE if t == "slide" else JOINT_HINGE,
                        stiffness=float(ja("stiffness", "0")), springref=float(ja("springref", "0")))

    xml_body, xml_parent = [], []       # per RawBody: the XML body it belongs to; per XML body: its parent XML body

    def walk(e, parent, active, xparent=-1):
        active = e.get("childclass", active)
        xid = len(xml_parent)
        xml_parent.append(xparent)
        joints = [parse_joint(j, active) for j in e.findall("joint")]
        name = e.get("name", "body%d" % len(bodies))
        pos, quat = _floats(e.get("pos"), 3, [0.0, 0.0, 0.0]), _floats(e.get("quat"), 4, [1.0, 0.0, 0.0, 0.0])
        for k in ("axisangle", "euler", "xyaxes", "zaxis"):
            if e.get(k) is not None:
                raise ValueError("body orientation must be given as quat")
        # joints 0 .. n-2 ride on massless bodies; the last one (or none) on the body itself
        for k, jt in enumerate(joints[:-1]):
            if not jt.name:
                jt.name = "%s_joint%d" % (name, k)
            bodies.append(RawBody("%s~%d" % (name, k), parent, pos if k == 0 else [0.0, 0.0, 0.0],
                                  quat if k == 0 else [1.0, 0.0, 0.0, 0.0], jt, []))
            xml_body.append(xid)
            parent = len(bodies) - 1
        idx = len(bodies)
        first = len(joints) <= 1
        jt = joints[-1] if joints else None
        if jt is not None and not jt.name:
            jt.name = "%s_joint%d" % (name, len(joints) - 1)
        bodies.append(RawBody(name, parent, pos if first else [0.0, 0.0, 0.0], quat if first else [1.0, 0.0, 0.0, 0.0], jt,
                              [parse_geom(g, active) for g in e.findall("geom")]))
        xml_body.append(xid)
        for s in e.findall("site"):
            sites[s.get("name")] = (idx, _floats(s.get("pos"), 3, [0.0, 0.0, 0.0]))
        for c in e:
            if c.tag not in ("joint", "geom", "site", "body", "inertial") + _VISUAL_BODY_TAGS:
                raise ValueError("unsupported body element <%s>" % c.tag)
            if c.tag == "inertial":
                raise ValueError("explicit <inertial> is not supported (inertiafromgeom only)")
        for c in e.findall("body"):
            walk(c, idx, active, xid)

    for e in world.findall("body"):
        walk(e, -1, world.get("childclass"))

    # what collides: every body geom against the one world plane, MuJoCo's contype / conaffinity rule
    plane = None
    if plane_elem is not None:
        pa = lambda k, d=None: dfl.attr("geom", plane_elem, None, k, d)     # noqa: E731
        pct, pca = int(pa("contype", "1")), int(pa("conaffinity", "1"))
        hit = False
        for b in bodies:
            for g in b.geoms:
                g.collide = bool((g._contype & pca) or (pct & g._conaffinity))
                hit = hit or g.collide
                if g.collide:
                    geom_solver.add(g._solver)
        if hit:
            q = _floats(plane_elem.get("quat"), 4, [1.0, 0.0, 0.0, 0.0])
            if q != [1.0, 0.0, 0.0, 0.0]:
                raise ValueError("rotated planes are not supported")
            geom_solver.add((tuple(_floats(pa("solref", "0.02 1"))), tuple(_floats(pa("solimp", "0.9 0.95 0.001 0.5 2")))))
            plane = RawPlane(pos=_floats(plane_elem.get("pos"), 3, [0.0, 0.0, 0.0]), normal=(0.0, 0.0, 1.0),
                             margin=float(pa("margin", "0")), friction=_floats(pa("friction", "1 0.005 0.0001"))[0],
                             condim=int(pa("condim", "3")))
    # ... and body geoms against each other (self_collision): MuJoCo's rule - the contype / conaffinity masks match, the
    # two bodies differ and are not parent and child.  Later geom first (on a chain: the deeper one, which is the order
    # compile_tree asks for).  swimmer.xml's segments are the vendored case (default contype = conaffinity = 1).
    auto_pairs = []
    if self_collision:
        flat = [(bi, g) for bi, b in enumerate(bodies) for g in b.geoms]
        for ib in range(len(flat)):
            for ia in range(ib):
                (ba, ga_), (bb, gb_) = flat[ia], flat[ib]
                xa, xb = xml_body[ba], xml_body[bb]
                if xa == xb or xml_parent[xa] == xb or xml_parent[xb] == xa:
                    continue
                if not ((ga_._contype & gb_._conaffinity) or (gb_._contype & ga_._conaffinity)):
                    continue
                for k, (bi, g) in ((ia, flat[ia]), (ib, flat[ib])):
                    if not g.name:
                        g.name = "%s_geom%d" % (bodies[bi].name, k)
                    geom_solver.add(g._solver)
                auto_pairs.append((gb_.name, ga_.name))
    if len(geom_solver) > 1 or len(limit_solver) > 1:
        raise ValueError("one solref / solimp set for contacts and one for joint limits")

    def full_solimp(si):
        return tuple(si) + (0.9, 0.95, 0.001, 0.5, 2.0)[len(si):]

    solref, solimp = geom_solver.pop() if geom_solver else ((0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0))
    lsolref, lsolimp = limit_solver.pop() if limit_solver else ((0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0))
    for b in bodies:
        for g in b.geoms:
            del g._contype, g._conaffinity, g._solver
    if totalmass is not None:           # MuJoCo scales every body mass and inertia by the same factor
        total = sum(_geom_inertial(g, MJ20_CAPSULE_CAP)[0] for b in bodies for g in b.geoms)
        for b in bodies:
            for g in b.geoms:
                g.density *= totalmass / total

    acts = []
    act = root.find("actuator")
    for m in (list(act) if act is not None else []):
        ma = lambda k, d=None: dfl.attr("motor", m, None, k, d)         # noqa: E731
        if m.tag not in ("motor", "position") or ma("ctrllimited", "false") != "true":
            raise ValueError("only ctrllimited <motor> / <position> actuators are supported")
        gear = _floats(ma("gear"), None, [1.0])[0]
        kp = float(m.get("kp", "1")) if m.tag == "position" else 0.0       # MJCF <position>: kp defaults to 1
        acts.append(RawActuator(m.get("joint"), gear, _floats(ma("ctrlrange"), 2), kp=kp))
    # explicit geom-geom collision candidates (<contact><pair geom1=... geom2=...>): manipulator geom first, object second
    pairs = list(auto_pairs)
    con = root.find("contact")
    for pr in (list(con) if con is not None else []):
        pairs.append((pr.get("geom1"), pr.get("geom2")))
    if task == TASK_REACH:
        if hand_site not in sites or sites[hand_site][0] < 0:
            raise ValueError("tracked site %r must be attached to a body" % hand_site)
        site_body, site_pos = sites[hand_site]
    else:
        site_body, site_pos = len(bodies) - 1, [0.0, 0.0, 0.0]
    target = sites.get(target_site, (-1, [0.0, 0.0, 0.0]))[1]
    return RawModel(bodies=bodies, actuators=acts, site_body=site_body, site_pos=site_pos, target_pos=target, plane=plane,
                    timestep=timestep, frame_skip=frame_skip, gravity=gravity, solref=solref, solimp=full_solimp(solimp),
                    solref_limit=lsolref, solimp_limit=full_solimp(lsolimp), density=density, viscosity=viscosity,
                    task=task, ctrl_cost=ctrl_cost, obs_skip=obs_skip, pairs=pairs)

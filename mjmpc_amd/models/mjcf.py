"""MJCF-subset loader: the part of MuJoCo's XML the rollout kernels can execute.

The three models the reference vendors (mjmpc/envs/assets/xml/sawyer.xml, swimmer.xml, half_cheetah.xml) and, since
round 4, the kinds of model its other experiment files name (examples/configs/classic_control/cartpole*.yml,
panda/tray_glass-v0.yml, sawyer/door-v0.yml, hand/*-v0.yml):

* ``<compiler angle="radian|degree" coordinate="local" inertiafromgeom="true|auto" settotalmass>``,
  ``<option timestep gravity density viscosity integrator="Euler" cone impratio collision>`` with ``<flag>`` (the parts a flag
  switches off are taken out of the model);
* ``<default>`` with nested classes, ``class=`` / ``childclass=`` (joint, geom and motor attributes);
* nested ``<body pos quat|axisangle|euler|xyaxes|zaxis>`` with any number of hinge / slide ``<joint>``s (anchor ``pos`` anywhere in
  the body) - a body with several joints becomes a chain of massless bodies, one joint each, which is what MuJoCo's
  kinematics does with it - or ONE ball joint (``limited range="0 max"``: a cone on its rotation angle), or a ``<freejoint/>`` /
  free joint (children of the world body);
  joint ``axis range limited|auto damping armature stiffness springref frictionloss margin ref``, ``solreflimit`` / ``solimplimit``,
  ``solreffriction`` / ``solimpfriction`` (per joint); explicit ``<inertial pos quat mass
  diaginertia|fullinertia>``;
* sphere, capsule, box and cylinder geoms (``fromto``, or ``size pos`` with ``quat`` / ``axisangle`` / ``euler``), ``density``,
  ``mass``, ``margin``, ``gap``, ``friction``, ``condim``, ``contype`` / ``conaffinity`` (what collides with the world plane and -
  ``self_collision`` - with other bodies is decided by MuJoCo's rule), ``solref`` / ``solimp`` / ``solmix`` / ``priority`` per geom (a contact's
  set is mixed from its two geoms', mj_contactParam; up to eight distinct sets per model; ``solref`` in the standard
  format ``(timeconst, dampratio)`` or the direct one ``(-stiffness, -damping)``, round 5);
* one world ``<geom type="plane">`` in any orientation, static sphere / capsule / box geoms on the world body (they
  collide with moving geoms through MuJoCo's contype / conaffinity rule or an explicit ``<pair>``), world and body
  ``<site>``s, ``<motor>`` / ``<position kp>`` / ``<velocity kv>`` / ``<general gainprm biasprm biastype=affine>`` actuators
  (``joint gear ctrlrange ctrllimited forcerange forcelimited``; no activation dynamics),
  ``<contact><pair geom1 geom2 [condim friction margin gap solref solimp]>``; geom pairs: sphere / capsule against sphere /
  capsule, sphere / capsule / box against box, sphere / capsule against cylinder (round 5);
* ``<equality><connect body1 body2 anchor>``, ``<weld body1 body2>`` and ``<joint joint1 joint2 polycoef>`` (``solref`` /
  ``solimp`` each),
  ``<tendon><fixed limited range><joint joint coef/>`` over one or two joints.

* the layout of robot model files: ``<include file>`` (anywhere, nested; sections that then occur several times are
  merged), ``<contact><exclude body1 body2>``, mesh / ellipsoid geoms (and cylinders there) as VISUALS (masked out of every
  collision, on a body with an explicit ``<inertial>``), ``<sensor>`` and ``<keyframe>`` sections (read by nobody here).

Anything that would change the simulation and is not modelled raises ValueError, so that a model is never silently
simulated wrongly (colliding meshes, mocap bodies, noslip iterations, option flags that are not modelled, other solvers, activation
dynamics ...); purely visual elements (asset, light, camera, material, rgba ...) are skipped.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .compile import _geom_inertial, _quat2mat
from .raw import (EQ_CONNECT, EQ_JOINT, EQ_WELD, GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, JOINT_BALL, JOINT_FREE, JOINT_HINGE, JOINT_SLIDE,
                  MJ20_CAPSULE_CAP, TASK_FORWARD, TASK_REACH, RawActuator, RawBody, RawEquality, RawGeom, RawInertial,
                  RawJoint, RawModel, RawPlane, RawTendon)

_VISUAL_BODY_TAGS = ("light", "camera")
_TOP_TAGS = ("compiler", "option", "default", "worldbody", "actuator", "asset", "size", "visual", "statistic", "custom",
             "equality", "tendon", "sensor", "keyframe")     # (sensors and keyframes do not enter the simulation)
_SHAPE_ONLY_TYPES = ("mesh", "ellipsoid")       # geom types that are accepted only where they cannot matter


def _expand_includes(node, basedir, depth=0):
    """MJCF ``<include file="...">``: the children of the included file's root take the element's place (paths relative
    to the main model's directory, as MuJoCo resolves them)."""
    if depth > 16:
        raise ValueError("<include> nested too deeply (a cycle?)")
    k = 0
    while k < len(node):
        c = node[k]
        if c.tag == "include":
            sub = ET.parse(os.path.join(basedir, c.get("file"))).getroot()
            _expand_includes(sub, basedir, depth + 1)
            node.remove(c)
            for j, x in enumerate(list(sub)):
                node.insert(k + j, x)
            k += len(sub)
        else:
            _expand_includes(c, basedir, depth)
            k += 1


def _mat2quat(R):
    """Unit quaternion (w, x, y, z) of a rotation matrix."""
    R = np.asarray(R, float)
    tr = np.trace(R)
    if tr > 0:
        sq = np.sqrt(tr + 1.0) * 2
        q = [0.25 * sq, (R[2, 1] - R[1, 2]) / sq, (R[0, 2] - R[2, 0]) / sq, (R[1, 0] - R[0, 1]) / sq]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        sq = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / sq, 0.25 * sq, (R[0, 1] + R[1, 0]) / sq, (R[0, 2] + R[2, 0]) / sq]
    elif R[1, 1] > R[2, 2]:
        sq = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / sq, (R[0, 1] + R[1, 0]) / sq, 0.25 * sq, (R[1, 2] + R[2, 1]) / sq]
    else:
        sq = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / sq, (R[0, 2] + R[2, 0]) / sq, (R[1, 2] + R[2, 1]) / sq, 0.25 * sq]
    return np.asarray(q, float)


def _rodrigues(ax, ang):
    ax = np.asarray(ax, float) / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _orientation(e, deg):
    """Rotation matrix of an element's ``quat`` / ``axisangle`` / ``euler`` (MuJoCo's default intrinsic x-y-z sequence) /
    ``xyaxes`` / ``zaxis`` attribute, or None when it carries none; angles in degrees unless the model says radian."""
    if e.get("quat") is not None:
        return _quat2mat(_floats(e.get("quat"), 4))
    if e.get("axisangle") is not None:
        v = _floats(e.get("axisangle"), 4)
        return _rodrigues(v[:3], v[3] * deg)
    if e.get("euler") is not None:
        a = [x * deg for x in _floats(e.get("euler"), 3)]
        return _rodrigues([1, 0, 0], a[0]) @ _rodrigues([0, 1, 0], a[1]) @ _rodrigues([0, 0, 1], a[2])
    if e.get("xyaxes") is not None:        # the frame's x axis and a vector in its xy plane (MuJoCo orthogonalises the second)
        v = np.asarray(_floats(e.get("xyaxes"), 6), float)
        x = v[:3] / np.linalg.norm(v[:3])
        y = v[3:] - (x @ v[3:]) * x
        if np.linalg.norm(y) < 1e-12:
            raise ValueError("xyaxes: the two vectors are parallel")
        y = y / np.linalg.norm(y)
        return np.stack([x, y, np.cross(x, y)], axis=1)
    if e.get("zaxis") is not None:         # the minimal rotation that takes (0, 0, 1) to the given direction
        z = np.asarray(_floats(e.get("zaxis"), 3), float)
        z = z / np.linalg.norm(z)
        ax = np.cross([0.0, 0.0, 1.0], z)
        sn, cs = np.linalg.norm(ax), z[2]
        if sn < 1e-12:
            return np.eye(3) if cs > 0 else _rodrigues([1.0, 0.0, 0.0], np.pi)
        return _rodrigues(ax / sn, np.arctan2(sn, cs))
    return None


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
        self.cls = {"main": {"joint": {}, "geom": {}, "motor": {}}}     # ("motor" holds the actuator defaults of every shortcut)
        top = root.find("default")
        if top is not None:
            self._read(top, "main", None)

    def _read(self, node, name, parent):
        base = {k: dict(v) for k, v in self.cls[parent].items()} if parent is not None else {"joint": {}, "geom": {}, "motor": {}}
        for tag in ("joint", "geom", "motor"):
            e = node.find(tag)
            if e is not None:
                base[tag].update(e.attrib)
        for tag in ("general", "position", "velocity"):     # MuJoCo keeps ONE actuator default per class: the shortcuts all set it
            e = node.find(tag)
            if e is not None:
                base["motor"].update(e.attrib)
        # defaults of elements whose attributes are read from the elements themselves only: they would change the simulation
        # unseen (site / mesh / material / camera / light defaults change nothing that is simulated)
        for tag in ("pair", "equality", "tendon", "cylinder", "muscle"):
            if node.find(tag) is not None and node.find(tag).attrib:
                raise ValueError("<default><%s>: defaults of this element are not supported (state the attributes on the elements)" % tag)
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
    _expand_includes(root, os.path.dirname(os.path.abspath(path)))
    # (several <default> / <worldbody> / <actuator> / <contact> ... sections, as included files bring them: merged in order)
    for tag in ("worldbody", "actuator", "contact", "equality", "tendon", "asset", "sensor"):
        same = root.findall(tag)
        for extra in same[1:]:
            same[0].extend(list(extra))
            root.remove(extra)
    tops = root.findall("default")
    for extra in tops[1:]:
        tops[0].extend(list(extra))
        root.remove(extra)
    for e in root:
        if e.tag == "contact" and all(c.tag in ("pair", "exclude") for c in e):
            continue                    # empty (sawyer.xml:94-99 holds comments only), <pair geom1 geom2> or <exclude body1 body2>, read below
        if e.tag not in _TOP_TAGS:
            raise ValueError("unsupported element <%s>" % e.tag)
    comp = root.find("compiler")
    totalmass = None
    deg = np.pi / 180.0                 # MuJoCo's default angle unit is the degree
    if comp is not None:
        if comp.get("angle", "degree") not in ("radian", "degree") or comp.get("coordinate", "local") != "local":
            raise ValueError("angle must be 'radian' or 'degree', coordinate 'local'")
        if comp.get("angle", "degree") == "radian":
            deg = 1.0
        if comp.get("inertiafromgeom", "auto") not in ("true", "auto"):
            raise ValueError("inertiafromgeom must be true or auto")
        if comp.get("eulerseq", "xyz") != "xyz":
            raise ValueError("eulerseq must be xyz")
        if comp.get("settotalmass") is not None:
            totalmass = float(comp.get("settotalmass"))
        # compiler attributes that would change masses and inertias and are not modelled (the rest - mesh / texture directories,
        # strippath, fusestatic, discardvisual, convexhull ... - leave the simulation alone)
        if float(comp.get("boundmass", "0")) != 0 or float(comp.get("boundinertia", "0")) != 0:
            raise ValueError("compiler boundmass / boundinertia are not supported")
        if comp.get("balanceinertia", "false") != "false":
            raise ValueError("compiler balanceinertia is not supported")
        if [int(x) for x in comp.get("inertiagrouprange", "0 5").split()] != [0, 5]:
            raise ValueError("compiler inertiagrouprange is not supported (every geom of a body counts towards its inertia)")
    opt = root.find("option")
    oget = (lambda k, d: opt.get(k, d)) if opt is not None else (lambda k, d: d)
    timestep = float(oget("timestep", "0.002"))
    gravity = _floats(oget("gravity", None), 3, [0.0, 0.0, -9.81])
    if oget("integrator", "Euler") != "Euler":
        raise ValueError("only the Euler integrator is supported")
    cone, impratio = oget("cone", "pyramidal"), float(oget("impratio", "1"))
    if cone not in ("pyramidal", "elliptic"):
        raise ValueError("cone must be pyramidal or elliptic")
    if _floats(oget("wind", None), 3, [0.0, 0.0, 0.0]) != [0.0, 0.0, 0.0]:
        raise ValueError("wind is not supported")
    if oget("solver", "Newton") != "Newton" or int(oget("noslip_iterations", "0")) != 0:
        raise ValueError("the Newton solver without noslip iterations is what is modelled")
    # <option><flag>: what a flag switches off is taken out of the model below (_apply_flags); flags that would change the
    # arithmetic in ways that are not modelled raise
    flags = dict(opt.find("flag").attrib) if opt is not None and opt.find("flag") is not None else {}
    _check_flags(flags)
    density, viscosity = float(oget("density", "0")), float(oget("viscosity", "0"))
    # <option collision>: "all" (MuJoCo's default) - pairs derived from contype / conaffinity AND the explicit <pair>s;
    # "predefined" - the explicit <pair>s only (no geom meets the plane through its masks); "dynamic" - the derived ones only
    collision = oget("collision", "all")
    if collision not in ("all", "predefined", "dynamic"):
        raise ValueError("<option collision> must be all, predefined or dynamic")
    dfl = _Defaults(root)

    bodies, sites = [], {}
    plane_elem = None
    world = root.find("worldbody")
    geom_solver, limit_solver, friction_solver = [], [], []     # the sets in use: the most frequent becomes the model's own
    DEF_SOL = ((0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0))

    inertia_from_geom_always = comp is not None and comp.get("inertiafromgeom", "auto") == "true"
    autolimits = comp is not None and comp.get("autolimits", "false") == "true"

    # (geoms an explicit <pair> names collide whatever their masks say)
    paired_names = {pr.get(k) for c_ in root.findall("contact") for pr in c_.findall("pair") for k in ("geom1", "geom2")}

    def parse_geom(e, active, has_inertial=False):
        ga = lambda k, d=None: dfl.attr("geom", e, active, k, d)        # noqa: E731
        t = ga("type", "sphere")
        if ga("mesh") is not None and e.get("type") is None:
            t = "mesh"                  # (a geom that names a mesh is a mesh geom)
        if (t == "cylinder" and int(ga("contype", "1")) == 0 and int(ga("conaffinity", "1")) == 0 and has_inertial
                and not inertia_from_geom_always and e.get("name") not in paired_names):
            return None                 # (a purely visual cylinder: neither collided nor integrated)
        if t in _SHAPE_ONLY_TYPES:
            # meshes, cylinders and ellipsoids are neither collided nor integrated here: such a geom is accepted where it
            # can do neither - masked out of every collision and on a body whose mass comes from an explicit <inertial>
            # (the usual layout of robot models: visual meshes, primitive collision geoms, inertials from CAD)
            if int(ga("contype", "1")) == 0 and int(ga("conaffinity", "1")) == 0 and has_inertial and not inertia_from_geom_always:
                return None
            raise ValueError("unsupported geom type %r (accepted only as a visual: contype = conaffinity = 0 on a body with an "
                             "explicit <inertial>)" % t)
        size = _floats(ga("size"))
        common = dict(density=float(ga("density", "1000")), margin=float(ga("margin", "0")), gap=float(ga("gap", "0")),
                      friction=_floats(ga("friction", "1 0.005 0.0001"))[0], condim=int(ga("condim", "3")),
                      name=e.get("name", ""))
        pos = np.asarray(_floats(e.get("pos"), 3, [0.0, 0.0, 0.0]))
        Rg = _orientation(e, deg)
        if t == "sphere":
            g = RawGeom(GEOM_SPHERE, size[0], tuple(pos), **common)
        elif t in ("capsule", "cylinder"):
            ft = _floats(e.get("fromto"), 6)
            if ft is not None:
                a, b = ft[:3], ft[3:]
            else:
                z = np.array([0.0, 0.0, 1.0]) if Rg is None else Rg[:, 2]
                a, b = tuple(pos - size[1] * z), tuple(pos + size[1] * z)
            g = RawGeom(GEOM_CAPSULE if t == "capsule" else GEOM_CYLINDER, size[0], tuple(a), tuple(b), **common)
        elif t == "box":
            if e.get("fromto") is not None:
                raise ValueError("box geoms take size / pos / orientation, not fromto")
            if size is None or len(size) != 3:
                raise ValueError("a box needs three half sizes")
            g = RawGeom(GEOM_BOX, 0.0, tuple(pos), tuple(size), quat=tuple(_mat2quat(Rg) if Rg is not None else (1.0, 0.0, 0.0, 0.0)),
                        **common)
        else:
            raise ValueError("unsupported geom type %r" % t)
        if ga("mass") is not None:          # MJCF: mass overrides density
            g.density = 1.0
            vol = _geom_inertial(g, MJ20_CAPSULE_CAP)[0]
            g.density = float(ga("mass")) / vol
        g._contype, g._conaffinity = int(ga("contype", "1")), int(ga("conaffinity", "1"))
        g._solver = (tuple(_floats(ga("solref", "0.02 1"))), tuple(_floats(ga("solimp", "0.9 0.95 0.001 0.5 2"))))
        g.solmix, g.priority = float(ga("solmix", "1")), int(ga("priority", "0"))
        return g

    world_geoms = []
    for e in world:
        if e.tag == "geom":
            if dfl.attr("geom", e, world.get("childclass"), "type", "sphere") == "plane":
                if plane_elem is not None:
                    raise ValueError("one world plane at most")
                plane_elem = e
            else:
                g = parse_geom(e, world.get("childclass"), has_inertial=True)      # (the world body has no mass to get wrong)
                if g is None:
                    continue
                if not g.name:
                    g.name = "world_geom%d" % len(world_geoms)
                world_geoms.append(g)
        elif e.tag == "site":
            sites[e.get("name")] = (-1, _floats(e.get("pos"), 3, [0.0, 0.0, 0.0]))
        elif e.tag != "body" and e.tag not in _VISUAL_BODY_TAGS:
            raise ValueError("unsupported worldbody element <%s>" % e.tag)

    def parse_joint(j, active):
        ja = lambda k, d=None: dfl.attr("joint", j, active, k, d)       # noqa: E731
        t = "free" if j.tag == "freejoint" else ja("type", "hinge")
        if t not in ("hinge", "slide", "ball", "free"):
            raise ValueError("unknown joint type %r" % t)
        jmargin, jref = float(ja("margin", "0")), float(ja("ref", "0"))
        if t in ("ball", "free") and (jmargin != 0.0 or jref != 0.0):
            raise ValueError("joint margin / ref are modelled for hinge and slide joints")
        if j.tag == "freejoint":            # MJCF: <freejoint/> takes no defaults: no damping, armature or friction loss
            return RawJoint(axis=[0.0, 0.0, 1.0], range=[0.0, 0.0], limited=False, name=j.get("name", ""), type=JOINT_FREE)
        lim_attr = ja("limited", "false")
        if lim_attr == "auto":          # MuJoCo >= 2.2.2: with <compiler autolimits="true"> a range implies the limit
            if autolimits:
                lim_attr = "true" if ja("range") is not None else "false"
            elif ja("range") is not None:
                raise ValueError("joint %r: limited=\"auto\" with a range needs <compiler autolimits=\"true\"> (MuJoCo "
                                 "refuses the model too)" % j.get("name", "?"))
            else:
                lim_attr = "false"
        if lim_attr not in ("true", "false"):
            raise ValueError("joint limited must be true, false or auto")
        limited = lim_attr == "true" and t != "free"      # (MuJoCo ignores limits on free joints)
        lim_set = fric_set = None
        if limited:
            lim_set = (tuple(_floats(ja("solreflimit", "0.02 1"))), tuple(_floats(ja("solimplimit", "0.9 0.95 0.001 0.5 2"))))
            limit_solver.append(lim_set)
        floss = float(ja("frictionloss", "0"))
        if floss != 0.0:
            fric_set = (tuple(_floats(ja("solreffriction", "0.02 1"))), tuple(_floats(ja("solimpfriction", "0.9 0.95 0.001 0.5 2"))))
            friction_solver.append(fric_set)
        ang = deg if t == "hinge" else 1.0      # (a hinge's range and spring reference are angles ...
        rng_ang = deg if t in ("hinge", "ball") else 1.0        # ... and so is a ball joint's range: the cone's half angle)
        rng = [x * rng_ang for x in _floats(ja("range"), 2, [0.0, 0.0])]
        jtype = {"hinge": JOINT_HINGE, "slide": JOINT_SLIDE, "ball": JOINT_BALL, "free": JOINT_FREE}[t]
        if t == "hinge":
            jmargin, jref = jmargin * deg, jref * deg       # (angles)
        return RawJoint(axis=_floats(ja("axis"), 3, [0.0, 0.0, 1.0]), range=rng, margin=jmargin, ref=jref,
                        limited=limited, damping=float(ja("damping", "0")), armature=float(ja("armature", "0")),
                        name=j.get("name", ""), type=jtype,
                        stiffness=float(ja("stiffness", "0")), springref=float(ja("springref", "0")) * ang,
                        pos=_floats(j.get("pos"), 3, [0.0, 0.0, 0.0]) if t in ("hinge", "ball") else [0.0, 0.0, 0.0],
                        frictionloss=floss, solref_limit=lim_set and lim_set[0], solimp_limit=lim_set and lim_set[1],
                        solref_friction=fric_set and fric_set[0], solimp_friction=fric_set and fric_set[1])

    xml_body, xml_parent, xml_name = [], [], []     # per RawBody: the XML body it belongs to; per XML body: its parent XML body, its name

    def walk(e, parent, active, xparent=-1):
        active = e.get("childclass", active)
        xid = len(xml_parent)
        xml_parent.append(xparent)
        joints = [parse_joint(j, active) for j in e if j.tag in ("joint", "freejoint")]
        name = e.get("name", "body%d" % len(bodies))
        xml_name.append(name)
        pos = _floats(e.get("pos"), 3, [0.0, 0.0, 0.0])
        Rb = _orientation(e, deg)
        quat = [1.0, 0.0, 0.0, 0.0] if Rb is None else list(_mat2quat(Rb))
        if any(j.type in (JOINT_BALL, JOINT_FREE) for j in joints) and len(joints) > 1:
            raise ValueError("a ball or free joint must be its body's only joint")
        if any(j.type == JOINT_FREE for j in joints) and parent >= 0:
            raise ValueError("a free joint belongs to a child of the world body")
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
        inertial = None
        for c in e:
            if c.tag not in ("joint", "freejoint", "geom", "site", "body", "inertial") + _VISUAL_BODY_TAGS:
                raise ValueError("unsupported body element <%s>" % c.tag)
            if c.tag == "inertial":
                ipos = _floats(c.get("pos"), 3, [0.0, 0.0, 0.0])
                Ri = _orientation(c, deg)
                Ri = np.eye(3) if Ri is None else Ri
                if c.get("fullinertia") is not None:
                    xx, yy, zz, xy, xz, yz = _floats(c.get("fullinertia"), 6)
                    I = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
                else:
                    I = Ri @ np.diag(_floats(c.get("diaginertia"), 3)) @ Ri.T
                inertial = RawInertial(float(c.get("mass")), ipos, I)
        if inertia_from_geom_always and inertial is not None and e.findall("geom"):
            inertial = None             # inertiafromgeom="true": MuJoCo computes the body's inertia from its geoms and ignores <inertial>
        if e.get("mocap", "false") == "true":
            raise ValueError("mocap bodies are not supported")
        geoms_ = [parse_geom(g, active, has_inertial=inertial is not None) for g in e.findall("geom")]
        bodies.append(RawBody(name, parent, pos if first else [0.0, 0.0, 0.0], quat if first else [1.0, 0.0, 0.0, 0.0], jt,
                              [g for g in geoms_ if g is not None], inertial))
        xml_body.append(xid)
        for s_ in e.findall("site"):
            sites[s_.get("name")] = (idx, _floats(s_.get("pos"), 3, [0.0, 0.0, 0.0]))
        for c in e.findall("body"):
            walk(c, idx, active, xid)

    for e in world.findall("body"):
        walk(e, -1, world.get("childclass"))

    # what collides: every body geom against the one world plane, MuJoCo's contype / conaffinity rule
    plane = None
    if plane_elem is not None:
        pa = lambda k, d=None: dfl.attr("geom", plane_elem, world.get("childclass"), k, d)     # noqa: E731
        pct, pca = int(pa("contype", "1")), int(pa("conaffinity", "1"))
        hit = False
        for b in bodies:
            for g in b.geoms:
                g.collide = collision != "predefined" and bool((g._contype & pca) or (pct & g._conaffinity))
                hit = hit or g.collide
                if g.collide:
                    geom_solver.append(g._solver)
        if hit:
            Rp = _orientation(plane_elem, deg)
            normal = (0.0, 0.0, 1.0) if Rp is None else tuple(Rp[:, 2])
            plane_set = (tuple(_floats(pa("solref", "0.02 1"))), tuple(_floats(pa("solimp", "0.9 0.95 0.001 0.5 2"))))
            geom_solver.append(plane_set)
            plane = RawPlane(pos=_floats(plane_elem.get("pos"), 3, [0.0, 0.0, 0.0]), normal=normal,
                             margin=float(pa("margin", "0")), gap=float(pa("gap", "0")), friction=_floats(pa("friction", "1 0.005 0.0001"))[0],
                             condim=int(pa("condim", "3")), solmix=float(pa("solmix", "1")), priority=int(pa("priority", "0")))
            plane._solver = plane_set
    # ... and geoms against each other (self_collision): MuJoCo's rule - the contype / conaffinity masks match, the two
    # bodies differ and are not parent and child (the world body is nobody's parent in that rule), and at least one of
    # them moves.  Later geom first (on a chain: the deeper one, which is the order compile_tree asks for).
    # swimmer.xml's segments are the vendored case (default contype = conaffinity = 1).
    auto_pairs = []
    con = root.find("contact")
    excluded = set()                    # <contact><exclude body1 body2>: no derived pair between geoms of these two bodies
    for ex in (con.findall("exclude") if con is not None else []):
        excluded.add(frozenset((ex.get("body1"), ex.get("body2"))))
    if self_collision and collision != "predefined":
        moving = [False] * len(bodies)
        for bi, b in enumerate(bodies):
            moving[bi] = b.joint is not None or (b.parent >= 0 and moving[b.parent])
        flat = [(-1, g) for g in world_geoms] + [(bi, g) for bi, b in enumerate(bodies) for g in b.geoms]
        for ib in range(len(flat)):
            for ia in range(ib):
                (ba, ga_), (bb, gb_) = flat[ia], flat[ib]
                if not ((ba >= 0 and moving[ba]) or (bb >= 0 and moving[bb])):
                    continue
                if ba >= 0 and bb >= 0:
                    xa, xb = xml_body[ba], xml_body[bb]
                    if xa == xb or xml_parent[xa] == xb or xml_parent[xb] == xa:
                        continue
                    if frozenset((xml_name[xa], xml_name[xb])) in excluded:
                        continue
                if not ((ga_._contype & gb_._conaffinity) or (gb_._contype & ga_._conaffinity)):
                    continue
                kinds = sorted((ga_.type, gb_.type))
                if GEOM_CYLINDER in kinds and kinds not in ([GEOM_SPHERE, GEOM_CYLINDER], [GEOM_CAPSULE, GEOM_CYLINDER]):
                    raise ValueError("geoms %r / %r would collide (contype / conaffinity) but a cylinder collides with the plane, "
                                     "spheres and capsules only (MuJoCo sends cylinder pairs to its general convex collider; "
                                     "cylinder-box and cylinder-cylinder have no closed form here): mask the pair out or use a "
                                     "capsule" % (ga_.name or "?", gb_.name or "?"))
                for k, (bi, g) in ((ia, flat[ia]), (ib, flat[ib])):
                    if not g.name:
                        g.name = "%s_geom%d" % (bodies[bi].name if bi >= 0 else "world", k)
                    geom_solver.append(g._solver)
                auto_pairs.append((gb_.name, ga_.name))
    # explicit geom-geom collision candidates (<contact><pair geom1=... geom2=...>)
    pairs = list(auto_pairs)
    every = {g.name: g for b in bodies for g in b.geoms if g.name}
    every.update({g.name: g for g in world_geoms})
    pair_params = {}
    for pr in (con.findall("pair") if con is not None and collision != "dynamic" else []):
        over = {}
        if pr.get("condim") is not None:
            over["condim"] = int(pr.get("condim"))
        if pr.get("friction") is not None:
            fr = _floats(pr.get("friction"))
            if len(fr) > 1 and fr[1] != fr[0]:
                raise ValueError("<pair friction>: the two tangential coefficients must be equal (isotropic pyramids)")
            over["friction"] = fr[0]
        if pr.get("margin") is not None or pr.get("gap") is not None:
            # (a <pair>'s own margin and gap, MuJoCo defaults 0: the contacts enter the solver while dist < margin - gap)
            over["margin"] = float(pr.get("margin", "0")) - float(pr.get("gap", "0"))
        if pr.get("solref") is not None:
            over["solref"] = tuple(_floats(pr.get("solref"), 2))
            if (over["solref"][0] > 0) != (over["solref"][1] > 0):
                raise ValueError("solref must be (timeconst, dampratio), both positive, or (-stiffness, -damping), both negative")
        if pr.get("solimp") is not None:
            over["solimp"] = tuple(_floats(pr.get("solimp")))
        if over:
            pair_params[(pr.get("geom1"), pr.get("geom2"))] = over
        pairs.append((pr.get("geom1"), pr.get("geom2")))
        for nm in (pr.get("geom1"), pr.get("geom2")):
            if nm in every:
                geom_solver.append(every[nm]._solver)
    for sets in (geom_solver, limit_solver, friction_solver):
        if any((r[0] > 0) != (r[1] > 0) for r, _ in sets):
            # (MuJoCo replaces such a pair by the default with a warning; here it is an error)
            raise ValueError("solref must be (timeconst, dampratio), both positive, or (-stiffness, -damping), both negative")

    def full_solimp(si):
        return tuple(si) + (0.9, 0.95, 0.001, 0.5, 2.0)[len(si):]

    def common(sets, tag, kref, kimp):
        # the model's own set: what the top-level <default> says, else the most frequent one in use; elements keep theirs
        # only where it differs
        d = dfl.cls["main"][tag]
        if kref in d or kimp in d:
            return (tuple(_floats(d.get(kref, "0.02 1"))), tuple(_floats(d.get(kimp, "0.9 0.95 0.001 0.5 2"))))
        return max(sets, key=sets.count) if sets else DEF_SOL

    solref, solimp = common(geom_solver, "geom", "solref", "solimp")
    lsolref, lsolimp = common(limit_solver, "joint", "solreflimit", "solimplimit")
    fsolref, fsolimp = common(friction_solver, "joint", "solreffriction", "solimpfriction")
    for g in [g for b in bodies for g in b.geoms] + world_geoms:
        if g._solver != (solref, solimp):
            g.solref, g.solimp = g._solver[0], full_solimp(g._solver[1])
        del g._contype, g._conaffinity, g._solver
    if plane is not None:
        if plane._solver != (solref, solimp):
            plane.solref, plane.solimp = plane._solver[0], full_solimp(plane._solver[1])
        del plane._solver
    for b in bodies:
        jt = b.joint
        if jt is None:
            continue
        if jt.solref_limit is not None:
            same = (tuple(jt.solref_limit), tuple(jt.solimp_limit)) == (lsolref, lsolimp)
            jt.solref_limit, jt.solimp_limit = (None, None) if same else (jt.solref_limit, full_solimp(jt.solimp_limit))
        if jt.solref_friction is not None:
            same = (tuple(jt.solref_friction), tuple(jt.solimp_friction)) == (fsolref, fsolimp)
            jt.solref_friction, jt.solimp_friction = (None, None) if same else (jt.solref_friction, full_solimp(jt.solimp_friction))
    if totalmass is not None:           # MuJoCo scales every body mass and inertia by the same factor
        total = sum(_geom_inertial(g, MJ20_CAPSULE_CAP)[0] for b in bodies for g in b.geoms if b.inertial is None)
        total += sum(b.inertial.mass for b in bodies if b.inertial is not None)
        for b in bodies:
            for g in b.geoms:
                g.density *= totalmass / total
            if b.inertial is not None:
                b.inertial.mass *= totalmass / total
                b.inertial.inertia = np.asarray(b.inertial.inertia, float) * (totalmass / total)

    # fixed tendons
    tendons = []
    tn = root.find("tendon")
    for t in (list(tn) if tn is not None else []):
        if t.tag != "fixed":
            raise ValueError("only fixed tendons are supported, got <%s>" % t.tag)
        if float(t.get("stiffness", "0")) != 0 or float(t.get("damping", "0")) != 0 or float(t.get("frictionloss", "0")) != 0:
            raise ValueError("tendon stiffness / damping / frictionloss are not supported")
        lim = t.get("limited", "false") == "true"
        tset = None
        if lim:
            tset = (tuple(_floats(t.get("solreflimit", "0.02 1"))), tuple(_floats(t.get("solimplimit", "0.9 0.95 0.001 0.5 2"))))
            if (tset[0][0] > 0) != (tset[0][1] > 0):
                raise ValueError("solref must be (timeconst, dampratio), both positive, or (-stiffness, -damping), both negative")
            if not limit_solver:
                lsolref, lsolimp = tset             # (no limited joint: the tendons' set is the model's limit set)
                limit_solver.append(tset)
            if tset == (tuple(lsolref), tuple(lsolimp)):
                tset = None
        tendons.append(RawTendon(t.get("name", "tendon%d" % len(tendons)),
                                 [(j.get("joint"), float(j.get("coef", "1"))) for j in t.findall("joint")],
                                 limited=lim, range=_floats(t.get("range"), 2, [0.0, 0.0]), margin=float(t.get("margin", "0")),
                                 solref_limit=tset and tset[0], solimp_limit=tset and full_solimp(tset[1])))

    acts = []
    act = root.find("actuator")
    for m in (list(act) if act is not None else []):
        ma = lambda k, d=None: dfl.attr("motor", m, None, k, d)         # noqa: E731
        if m.tag not in ("motor", "position", "velocity", "general"):
            raise ValueError("unsupported actuator <%s> (motor, position, velocity and general are)" % m.tag)
        for k in ("dyntype", "gaintype", "biastype"):
            if m.tag == "general" and ma(k, "none" if k != "gaintype" else "fixed") not in (
                    ("none",) if k == "dyntype" else (("fixed",) if k == "gaintype" else ("none", "affine"))):
                raise ValueError("general actuators: dyntype none, gaintype fixed, biastype none / affine")
        forcerange = tuple(_floats(ma("forcerange"), 2)) if ma("forcelimited", "false") == "true" else None
        limited = ma("ctrllimited", "false") == "true"
        if ma("ctrlrange") is None:
            raise ValueError("an actuator needs a ctrlrange (it bounds the action space; ctrllimited says whether the physics clamp)")
        gear = _floats(ma("gear"), None, [1.0])[0]
        gainprm, biasprm, kp = None, None, 0.0
        if m.tag == "position":
            kp = float(ma("kp", "1"))                      # MJCF <position>: kp defaults to 1
        elif m.tag == "velocity":
            kv = float(ma("kv", "1"))
            gainprm, biasprm = kv, (0.0, 0.0, -kv)
        elif m.tag == "general":
            gainprm = _floats(ma("gainprm"), None, [1.0])[0]
            bp = list(_floats(ma("biasprm"), None, [0.0])) + [0.0, 0.0, 0.0]
            biasprm = tuple(bp[:3]) if ma("biastype", "none") == "affine" else (0.0, 0.0, 0.0)
        if m.get("joint") is None and m.get("tendon") is None:
            raise ValueError("an actuator acts on a joint or on a fixed tendon")
        if m.get("joint") is not None:
            jt = next((b.joint for b in bodies if b.joint is not None and b.joint.name == m.get("joint")), None)
            if jt is None:
                raise ValueError("actuator on an unknown joint %r" % m.get("joint"))
            if jt.type not in (JOINT_HINGE, JOINT_SLIDE):
                raise ValueError("actuators act on hinge and slide joints (a ball / free joint takes a gear vector, which is not modelled)")
        if len(_floats(ma("gear"), None, [1.0])) > 1 and any(x != 0 for x in _floats(ma("gear"))[1:]):
            raise ValueError("actuator gear: only the first (scalar) entry is modelled")
        acts.append(RawActuator(m.get("joint") or "", gear, _floats(ma("ctrlrange"), 2), kp=kp, tendon=m.get("tendon") or "",
                                gainprm=gainprm, biasprm=biasprm, ctrllimited=limited, forcerange=forcerange))

    # equality constraints
    equalities = []
    eqs = root.find("equality")
    for q in (list(eqs) if eqs is not None else []):
        if q.get("active", "true") != "true":
            continue
        kw = dict(solref=tuple(_floats(q.get("solref", "0.02 1"))), solimp=full_solimp(_floats(q.get("solimp", "0.9 0.95 0.001 0.5 2"))))
        if q.tag == "connect":
            equalities.append(RawEquality(EQ_CONNECT, q.get("body1"), q.get("body2") or "", anchor=_floats(q.get("anchor"), 3), **kw))
        elif q.tag == "joint":
            equalities.append(RawEquality(EQ_JOINT, q.get("joint1"), q.get("joint2") or "",
                                          polycoef=_floats(q.get("polycoef", "0 1 0 0 0"), 5), **kw))
        elif q.tag == "weld":
            if q.get("relpose") is not None:
                raise ValueError("weld relpose is not supported (the relative pose at qpos0 is what a weld keeps)")
            equalities.append(RawEquality(EQ_WELD, q.get("body1"), q.get("body2") or "", **kw))
        else:
            raise ValueError("equality <%s> is not supported (connect, weld and joint are)" % q.tag)

    if task != TASK_FORWARD:                # (reach and reorientation tasks track a site; forward progress reads qpos[0])
        if hand_site not in sites or sites[hand_site][0] < 0:
            raise ValueError("tracked site %r must be attached to a body" % hand_site)
        site_body, site_pos = sites[hand_site]
    else:
        site_body, site_pos = len(bodies) - 1, [0.0, 0.0, 0.0]
    target = sites.get(target_site, (-1, [0.0, 0.0, 0.0]))[1]
    sensors = {}
    for sec in root.findall("sensor"):
        for e in sec:
            if e.get("name"):
                sensors[e.get("name")] = float(e.get("noise", "0"))
    raw = RawModel(sensors=sensors, bodies=bodies, actuators=acts, site_body=site_body, site_pos=site_pos, target_pos=target, plane=plane,
                   timestep=timestep, frame_skip=frame_skip, gravity=gravity, solref=solref, solimp=full_solimp(solimp),
                   solref_limit=lsolref, solimp_limit=full_solimp(lsolimp), density=density, viscosity=viscosity, cone=cone, impratio=impratio,
                   task=task, ctrl_cost=ctrl_cost, obs_skip=obs_skip, pairs=pairs, pair_params=pair_params, world_geoms=world_geoms,
                   equalities=equalities, tendons=tendons, solref_friction=fsolref, solimp_friction=full_solimp(fsolimp))
    return _apply_flags(raw, flags)


# MJCF <option><flag> (mjtDisableBit / mjtEnableBit [EXT]).  "enable" by default: constraint, equality, frictionloss, limit,
# contact, passive, gravity, clampctrl, warmstart, filterparent, actuation, refsafe, sensor, midphase; "disable" by default:
# override, energy, fwdinv, sensornoise.
_FLAGS_WITHOUT_EFFECT = ("warmstart", "energy", "fwdinv", "sensornoise", "sensor", "midphase")     # (the solver here runs to
#   convergence from any start; energies, inverse dynamics and sensors are read by nobody; midphase only prunes)
_FLAGS_MODELLED = ("constraint", "equality", "frictionloss", "limit", "contact", "gravity", "clampctrl", "actuation")


def _check_flags(flags):
    for k, v in flags.items():
        if v not in ("enable", "disable"):
            raise ValueError("<option><flag %s=%r>: enable or disable" % (k, v))
        if k in _FLAGS_WITHOUT_EFFECT or k in _FLAGS_MODELLED:
            continue
        if k == "override" and v == "disable" or k in ("passive", "filterparent", "refsafe") and v == "enable":
            continue
        # override (contact parameters replaced by o_margin / o_solref / o_solimp), passive off (MuJoCo 2.0's Euler step keeps the
        # implicit damping term while the damping force is gone), filterparent off (another collision set), refsafe off
        raise ValueError("<option><flag %s=%r> is not modelled" % (k, v))


def _apply_flags(raw, flags):
    off = lambda k: flags.get(k, "enable") == "disable"        # noqa: E731
    every = off("constraint")
    if off("gravity"):
        raw.gravity = (0.0, 0.0, 0.0)
    if every or off("contact"):
        raw.plane, raw.pairs, raw.pair_params = None, [], {}
    if every or off("equality"):
        raw.equalities = []
    for b in raw.bodies:
        if b.joint is None:
            continue
        if every or off("limit"):
            b.joint.limited = False
        if every or off("frictionloss"):
            b.joint.frictionloss = 0.0
    if every or off("limit"):
        for t in raw.tendons:
            t.limited = False
    for a in raw.actuators:
        if off("clampctrl"):
            a.ctrllimited = False
        if off("actuation"):
            a.gear = 0.0                    # (mj_fwdActuation returns with no actuator force at all)
    return raw

"""A synthetic 24-DoF hand-on-an-arm kinematic TREE (not a reference asset).

BASELINE config 5 names ``pen-v0`` (the Adroit hand of mj_envs: 24 actuated hinge dofs + a free pen), whose MJCF,
meshes and tendons are not in the reference tree (hard part 6).  This model has the same SHAPE of work for
the rollout engine - a branching tree of 24 hinges, gravity, joint limits on every dof, several sphere/plane
contacts - so the tree kernel, its oracle parity and the DMD-MPC 65536 x 64 throughput line run on the right kind of
model; it makes no claim to Adroit's numbers.  Layout (bodies in depth-first order, as an MJCF file lists them):

    world - forearm: 4-dof arm (pan z, lift y, elbow y, wrist y) - palm (welded) - 5 fingers x 4 hinges
    (abduction z, then three flexion joints y), a collision sphere on every fingertip, a table plane below.
"""
import numpy as np

from .raw import (GEOM_CAPSULE, GEOM_SPHERE, RawActuator, RawBody, RawGeom, RawJoint, RawModel, RawPlane)

_MARGIN = 0.002


def _cap(name, r, a, b, density=1000.0):
    return RawGeom(GEOM_CAPSULE, r, a, b, density=density, margin=_MARGIN, name=name)


def _hinge(name, axis, lo, hi, damping, armature):
    return RawJoint(axis=axis, range=(lo, hi), limited=True, damping=damping, armature=armature, name=name)


def hand24_raw(gravity=(0.0, 0.0, -9.81)) -> RawModel:
    X, Y, Z = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
    bodies = [
        RawBody("arm_pan", -1, (0.0, -0.4, 0.0), joint=_hinge("arm_pan", Z, -1.6, 1.6, 2.0, 0.01),
                geoms=[_cap("g_pan", 0.05, (0, 0, -0.2), (0, 0, 0.05))]),
        RawBody("arm_lift", 0, (0.05, 0.0, 0.05), joint=_hinge("arm_lift", Y, -1.2, 1.2, 2.0, 0.01),
                geoms=[_cap("g_upper", 0.04, (0, 0, 0), (0.3, 0, 0))]),
        RawBody("arm_elbow", 1, (0.3, 0.0, 0.0), joint=_hinge("arm_elbow", Y, -2.0, 0.3, 1.0, 0.005),
                geoms=[_cap("g_fore", 0.035, (0, 0, 0), (0.25, 0, 0))]),
        RawBody("arm_wrist", 2, (0.25, 0.0, 0.0), joint=_hinge("arm_wrist", Y, -0.8, 0.8, 0.5, 0.002),
                geoms=[_cap("g_wrist", 0.02, (0, -0.02, 0), (0, 0.02, 0))]),
        # welded palm: a slightly rotated plate carried by the wrist link
        RawBody("palm", 3, (0.03, 0.0, 0.0), quat=(np.cos(0.1), 0.0, np.sin(0.1), 0.0),
                geoms=[_cap("g_palm_a", 0.012, (0.0, -0.04, 0), (0.08, -0.04, 0)),
                       _cap("g_palm_b", 0.012, (0.0, 0.04, 0), (0.08, 0.04, 0)),
                       _cap("g_palm_c", 0.012, (0.08, -0.04, 0), (0.08, 0.04, 0))]),
    ]
    palm = 4
    # five fingers fanned over the palm's far edge; the thumb (f0) sits on the side and points outwards
    roots = [((0.02, -0.05, 0.0), -0.9), ((0.085, -0.036, 0.0), -0.12), ((0.09, -0.012, 0.0), 0.0),
             ((0.085, 0.012, 0.0), 0.08), ((0.08, 0.036, 0.0), 0.2)]
    lens = [(0.035, 0.03, 0.025), (0.04, 0.028, 0.02), (0.045, 0.03, 0.022), (0.04, 0.028, 0.02), (0.032, 0.022, 0.018)]
    for f, ((pos, yaw), (l1, l2, l3)) in enumerate(zip(roots, lens)):
        q = (np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2))
        r = 0.008 if f else 0.009
        b0 = len(bodies)
        bodies.append(RawBody("f%d_abd" % f, palm, pos, quat=q,
                              joint=_hinge("f%d_abd" % f, Z, -0.35, 0.35, 0.05, 0.0005),
                              geoms=[RawGeom(GEOM_SPHERE, r, (0, 0, 0), margin=_MARGIN, name="g_f%d_k" % f)]))
        bodies.append(RawBody("f%d_prox" % f, b0, (0.0, 0.0, 0.0),
                              joint=_hinge("f%d_prox" % f, Y, -0.2, 1.5, 0.05, 0.0005),
                              geoms=[_cap("g_f%d_p" % f, r, (0, 0, 0), (l1, 0, 0))]))
        bodies.append(RawBody("f%d_mid" % f, b0 + 1, (l1, 0.0, 0.0),
                              joint=_hinge("f%d_mid" % f, Y, 0.0, 1.6, 0.03, 0.0003),
                              geoms=[_cap("g_f%d_m" % f, 0.9 * r, (0, 0, 0), (l2, 0, 0))]))
        bodies.append(RawBody("f%d_dist" % f, b0 + 2, (l2, 0.0, 0.0),
                              joint=_hinge("f%d_dist" % f, Y, 0.0, 1.2, 0.02, 0.0002),
                              geoms=[_cap("g_f%d_d" % f, 0.8 * r, (0, 0, 0), (l3, 0, 0)),
                                     RawGeom(GEOM_SPHERE, 0.8 * r, (l3, 0, 0), collide=True, margin=_MARGIN,
                                             name="tip%d" % f)]))
    names = [b.joint.name for b in bodies if b.joint is not None]
    gears = [8.0, 8.0, 4.0, 1.0] + [0.05, 0.08, 0.05, 0.03] * 5
    actuators = [RawActuator(n, g, (-1.0, 1.0)) for n, g in zip(names, gears)]
    index_tip = [i for i, b in enumerate(bodies) if b.name == "f1_dist"][0]
    return RawModel(bodies=bodies, actuators=actuators, site_body=index_tip, site_pos=(0.02, 0.0, 0.0),
                    target_pos=(0.45, -0.2, 0.05), plane=RawPlane(pos=(0.0, 0.0, -0.12), normal=Z, margin=_MARGIN),
                    timestep=0.005, frame_skip=2, gravity=gravity)

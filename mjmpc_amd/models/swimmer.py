"""``Swimmer-v0``'s model as the reference vendors it (mjmpc/envs/assets/xml/swimmer.xml), restated as a RawModel.

A planar five-segment swimmer in a dense, slightly viscous medium: a floating root (slide x, slide y, hinge z on the
torso - three joints on one body, i.e. a chain of two massless bodies and the torso) and four limited hinges driven
by motors; no gravity-relevant dof, no ground contact (the plane has contype = conaffinity = 0, swimmer.xml:28).
What moves it is MuJoCo's inertia-box fluid model (``<option viscosity density>``, swimmer.xml:16).

Self-collision: every segment carries the default contype = conaffinity = 1, so MuJoCo collides the six pairs of
segments that are not parent and child (capsule-capsule, condim 3, friction 1; one contact point per pair here).  By the
geometry of the joint ranges (+-1.5 rad, segments 0.3 long) segments two apart can never touch (gap >= 0.17 m); segments
three or four apart touch when the joints between them are all bent the same way beyond ~1.15 rad - the chain curled
into a loop (tests/test_locomotion_cpu.py).  ``self_collision=False`` leaves the pairs out.  Task (mjmpc/envs/basic/swimmer.py:7-24): frame_skip 4, reward = forward progress of qpos[0] / dt - 1e-4 |a|^2,
observation = [qpos[2:], qvel].
"""

from .raw import (GEOM_CAPSULE, JOINT_HINGE, JOINT_SLIDE, TASK_FORWARD, RawActuator, RawBody, RawGeom, RawJoint, RawModel)

_RADII = (0.07, 0.065, 0.06, 0.055, 0.05)           # swimmer.xml:36,39,42,45,48


def swimmer_raw(frame_skip=4, self_collision=True) -> RawModel:
    free = dict(range=(-1.5, 1.5), limited=False)   # swimmer.xml:33-35: limited="false" (the default range is inherited, unused)
    # size="r 0.15" pos="0.15 0 0" quat="0.707 0 -0.707 0": the capsule's z axis turned onto -x, spanning x in [0, 0.3]
    def seg(i):
        return RawGeom(GEOM_CAPSULE, _RADII[i], (0.3, 0.0, 0.0), (0.0, 0.0, 0.0), density=1000.0, condim=3, name="seg%d" % i)

    bodies = [
        RawBody("root_x", -1, (0.0, 0.0, 0.03), joint=RawJoint((1, 0, 0), name="root_x", type=JOINT_SLIDE, **free)),
        RawBody("root_y", 0, (0.0, 0.0, 0.0), joint=RawJoint((0, 1, 0), name="root_y", type=JOINT_SLIDE, **free)),
        RawBody("torso", 1, (0.0, 0.0, 0.0), joint=RawJoint((0, 0, 1), name="root_z", type=JOINT_HINGE, **free),
                geoms=[seg(0)]),
    ]
    for i in range(1, 5):                           # swimmer.xml:37-50: pos="0.3 0 0", hinge z, default range +-1.5
        bodies.append(RawBody("seg%d" % i, len(bodies) - 1, (0.3, 0.0, 0.0),
                              joint=RawJoint((0, 0, 1), range=(-1.5, 1.5), limited=True, name="j%d" % i), geoms=[seg(i)]))
    actuators = [RawActuator("j%d" % i, 20.0, (-1.0, 1.0)) for i in range(1, 5)]      # swimmer.xml:58-63
    return RawModel(bodies=bodies, actuators=actuators, site_body=len(bodies) - 1, site_pos=(0.0, 0.0, 0.0), target_pos=(0.0, 0.0, 0.0),
                    plane=None, timestep=0.005, frame_skip=frame_skip, gravity=(0.0, 0.0, -9.81),
                    density=1000.0, viscosity=0.000894, task=TASK_FORWARD, ctrl_cost=1e-4, obs_skip=2,
                    pairs=[("seg%d" % b, "seg%d" % a) for b in range(5) for a in range(b - 1)] if self_collision else [])

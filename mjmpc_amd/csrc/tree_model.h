// Constant block of a compiled kinematic TREE of hinge / slide links (mjmpc_amd/models/compile_tree.py, TREE_LAYOUT).
// Offsets are in scalars; per-link fields are stored [component][32 lanes].  Links are numbered depth-first
// (a link's subtree is the contiguous index range [i, i + subsize_i)), link frames are world-aligned at qpos0.
#pragma once

namespace mjmpc {

constexpr int TL = 32;              // lanes per particle (one per link / dof)
constexpr int TREE_MAX_SPHERES = 16; // contact points: spheres against the plane (a colliding capsule is its two end
                                     // spheres) and geom-geom pairs (one point each)
// contact record: [0] link A, [1:4] centre / segment start on A, [4] radius, [5] margin, [6] invweight (both bodies),
// [7] mu (0: frictionless row), [8:11] capsule axis (plane contacts: frame hint) / segment vector on A (geom-geom),
// [11] depth of link A in the elimination tree, [12] kind (0 sphere-plane, 1 geom-geom), [13] link B, [14:17] segment
// start on B, [17] radius B, [18:21] segment vector on B
constexpr int TREE_SPH_STRIDE = 24;
constexpr int TREE_SOL_CLASSES = 8;

enum TreeOffset : int {
    // ---- staged in LDS by the kernel.  The first 14 per-link fields are the arm block's, 32 lanes wide
    T_OFF = 0,                          // 3 x 32   joint anchor minus the parent link's, world axes at qpos0
    T_AXIS = T_OFF + 3 * TL,            // 3 x 32
    T_MASS = T_AXIS + 3 * TL,
    T_COM = T_MASS + TL,                // 3 x 32   relative to the joint anchor
    T_INERTIA = T_COM + 3 * TL,         // 6 x 32   xx yy zz xy xz yz about the link's centre of mass
    T_ARMATURE = T_INERTIA + 6 * TL,
    T_DAMPING = T_ARMATURE + TL,
    T_RANGE_LO = T_DAMPING + TL,
    T_RANGE_HI = T_RANGE_LO + TL,
    T_LIMITED = T_RANGE_HI + TL,
    T_GEAR = T_LIMITED + TL,
    T_CTRL_LO = T_GEAR + TL,
    T_CTRL_HI = T_CTRL_LO + TL,
    T_DOF_INVW = T_CTRL_HI + TL,
    // springs, medium, servos
    T_STIFFNESS = T_DOF_INVW + TL,
    T_SPRINGREF = T_STIFFNESS + TL,
    T_FBOX = T_SPRINGREF + TL,          // 3 x 32   box of equal inertia (fluid model), 0 for massless links
    T_FROT = T_FBOX + 3 * TL,           // 9 x 32   principal axes of the link's inertia in the link frame (row-major)
    T_KPG = T_FROT + 9 * TL,            // actuator bias on the length, at the joint: -gear^2 biasprm[1] (position servo: gear^2 kp)
    T_KVG = T_KPG + TL,                 // ... on the velocity: -gear^2 biasprm[2] (<velocity kv>: gear^2 kv); explicit, not part of M + hB
    T_TAU0 = T_KVG + TL,                // ... constant: gear biasprm[0]
    T_TAU_LO = T_TAU0 + TL,             // forcerange of the dof's actuator at the joint: gear * forcerange, lower / upper (+-inf: none)
    T_TAU_HI = T_TAU_LO + TL,
    // actuators on fixed tendons (over one or two joints): the dof's coefficient in the tendon (1: a joint actuator), the
    // other dof of the tendon (-1: none) and its coefficient; length = tcoef q + tpcoef q_partner, torque = tcoef * (the
    // tendon's clamped force through the gear); T_GEAR / T_KPG / T_KVG / T_TAU0 / T_TAU_LO / _HI then act on the tendon
    T_TCOEF = T_TAU_HI + TL,
    T_TPARTNER = T_TCOEF + TL,
    T_TPCOEF = T_TPARTNER + TL,
    // scalars
    T_NV = T_TPCOEF + TL,
    T_TIMESTEP,
    T_FRAME_SKIP,
    T_JUMPS,                            // pointer-jumping rounds = ceil(log2(tree depth))
    T_SITE_LINK,
    T_SITE_POS,                         // 3
    T_N_SPHERE = T_SITE_POS + 3,
    T_PLANE_N,                          // 3
    T_PLANE_D = T_PLANE_N + 3,
    T_GRAVITY,                          // 3
    T_NU = T_GRAVITY + 3,
    T_TASK,                             // 0 reach a target with the site, 1 forward progress of qpos[0], 2 reorient an object
    T_CTRL_COST,
    T_OBS_SKIP,
    T_DENSITY,
    T_VISCOSITY,
    T_ANY_FRICTION,                     // the model needs the full instantiation (friction cones, geom-geom pairs, servos)
    T_SITE_AXIS,                        // 3: task 2, the object's axis in the site link's frame
    T_TARGET_DIR = T_SITE_AXIS + 3,     // 3: ... and the direction it should point in
    // the solver-parameter sets of the model ({K, B, dmin, dmax, width, mid, power} each): a contact record names its set in
    // [21] (the two geoms' solref / solimp mixed by mj_contactParam; tendon limits: the tendon's), a dof in T_DOFCLS
    T_SOLTAB = T_TARGET_DIR + 3,        // TREE_SOL_CLASSES x 7
    T_DOFCLS = T_SOLTAB + 8 * 7,        // 32: limit-row set + 8 * friction-loss-row set of the dof
    T_SPH = T_DOFCLS + TL,              // TREE_MAX_SPHERES x TREE_SPH_STRIDE
    // ---- read once per launch, from global memory: topology, joint kinds, action map
    T_TOPO = T_SPH + TREE_MAX_SPHERES * TREE_SPH_STRIDE,
    T_PARENT = T_TOPO,                  // parent link, -1 for a root
    T_SUBSIZE = T_PARENT + TL,          // links in my subtree, myself included
    T_ANC = T_SUBSIZE + TL,             // 5 x 32: ancestor at distance 1, 2, 4, 8, 16 (-1: none)
    T_ANCMASK = T_ANC + 5 * TL,         // 2 x 32: bits 0-15 / 16-31 of {j : link j is me or one of my ancestors}
                                        // (two halves so that the f32 copy of the block holds them exactly)
    T_JTYPE = T_ANCMASK + 2 * TL,       // 1 hinge, 2 slide
    T_ACT = T_JTYPE + TL,               // index of the action that drives this dof (-1: none)
    T_EPARENT = T_ACT + TL,             // parent in the ELIMINATION tree of the sparse factorisation: the kinematic parent,
                                        // except that a manipulator's root hangs under the last link of the object it
                                        // touches (geom-geom contacts couple the two trees; see compile_tree.py)
    // tree-sparse L'DL: links of equal height above their deepest leaf are eliminated together (one round per height)
    T_DEPTH = T_EPARENT + TL,           // 32: strict ancestors of the link in the elimination tree
    T_N_ROUNDS = T_DEPTH + TL,          // max height + 1
    T_ELIM,                             // 31 x 32, [entry][lane]: my descendants sorted by height, packed
                                        // k | distance << 8 | height << 16; -1 terminates the list
    // ---- constants of the GENERAL instantiation (round 4): ball / free joints, friction loss, boxes, equalities, tendon
    // limits.  Models that need none of it (T_GEN = 0) run the instantiations of the earlier rounds, which never read this.
    T_GEN = T_ELIM + (TL - 1) * TL,     // 1: the model needs the general instantiation
    T_NQ,                               // entries of MuJoCo's qpos (a ball joint: 4 for 3 dofs, a free joint 7 for 6)
    T_HAS_BALL,
    T_FRICTIONLOSS,                     // 32: dry friction per dof (0: no row)
    T_QADR = T_FRICTIONLOSS + TL,       // 32: the link's entry in qpos (BALL_X link: the quaternion's w); -1: none
    T_QOFF = T_QADR + TL,               // 32: added to the link's coordinate in qpos (free joint translations: the body position)
    T_PEXT = T_QOFF + TL,               // TREE_MAX_SPHERES x TREE_PEXT_STRIDE: what the new record kinds need beyond [24]
    T_QW0 = T_PEXT + TREE_MAX_SPHERES * 24,     // 32: BALL_X links: w of the joint's qpos0 quaternion q0 (x, y, z in the three
                                        // links' T_QOFF): identity for a ball joint, the body's orientation for a free
                                        // joint, whose qpos quaternion is ABSOLUTE - the kernel's is relative, qpos = q0 * q_link
    T_JMARGIN = T_QW0 + TL,             // 32 (round 5): MJCF joint margin - a hinge / slide dof's limit row exists while dist < margin
    TREE_BLOB_LEN = T_JMARGIN + TL
};
constexpr int TREE_PEXT_STRIDE = 24;    // [0:3] box half sizes, [3:12] box orientation in its link's frame | dof row: [0] 0 joint
                                        // equality / 1 tendon limit, [1] coef A (joint equality: 1 = anchor dof is joint 2),
                                        // [2] coef B, [3:5] range, [5] margin, [6:11] polycoef | [12:19] the row's own solver
                                        // set (equalities), [19] bilateral, [20] weld: +1 link A carries body 1, -1 body 2; weld: [0:9] body 2's orientation at qpos0
// link kinds (T_JTYPE): a ball joint is three links - the first holds the quaternion and turns the frame, the others ride
// along with the body's own y / z axes; a free joint is three slides along the world axes and a ball
enum TreeLinkKind : int { LINK_HINGE = 1, LINK_SLIDE = 2, LINK_BALL_X = 3, LINK_BALL_Y = 4, LINK_BALL_Z = 5 };
// contact-record kinds ([12])
enum TreePointKind : int { PT_PLANE = 0, PT_SEGSEG = 1, PT_SPHERE_BOX = 2, PT_BOX_SPHERE = 3, PT_CONNECT = 4, PT_DOFROW = 5,
                           PT_WELD = 6,        // (a weld equality is a PT_CONNECT record at body 2's origin plus a PT_WELD record: its rotation rows)
                           // round 5 - records that come in GROUPS of [23] consecutive ones, [22] = the record's index in its group:
                           PT_PLANE_CYL = 7,   // a cylinder on the plane (mjc_PlaneCylinder): candidate point k of four; [1:4] centre, [4] radius,
                                               // [8:11] axis, [14] half height
                           PT_BOX_BOX = 10,    // two boxes: contact k of four (SAT, then face clipping or the edge pair; see the oracle's box_box);
                                               // box A in PEXT [0:12] as PT_SPHERE_BOX's, box B's orientation in ITS link's frame as a
                                               // quaternion in PEXT [12:16], its half sizes in [16:19]
                           PT_SEG_CYL = 11,    // a sphere / capsule (geom A) against a cylinder (geom B: its axis end to end in [14:17] + [18:21],
                           PT_CYL_SEG = 12,    // radius [17]) and the other way round: candidate [22] of [23] = 1 (sphere) or 3 (capsule)
                           PT_CAPSULE_BOX = 8, PT_BOX_CAPSULE = 9 };   // a capsule against a box: candidate k of three (the axis' nearest point,
                                               // the two ends); segment as PT_SEGSEG, box as PT_SPHERE_BOX.  (A box's corners on the plane
                                               // are PT_PLANE records with [23] = 8, [22] = corner, [14:17] = the box centre: mjc_PlaneBox
                                               // skips corners above the centre and keeps four contacts)

// device state: qpos[32] | qvel[32] | target[3] | site of the fresh observation[3] | quaternion w[32], one entry per LINK
// (a BALL_X link keeps x, y, z in the qpos entries of its three links and w in its own w entry)
constexpr int TREE_STATE_LEN = 3 * TL + 6;
constexpr int TREE_QW = 2 * TL + 6;
// reset record (TreeFusion::reset_rec): a device state vector, then the site and the object axis at the reset state
constexpr int TREE_RESET_LEN = TREE_STATE_LEN + 6;
constexpr int TREE_NQ_MAX = 40;
// the C ABI's state vectors (mjmpc_tree_set_shard_states): MuJoCo's layout - qpos[40] | qvel[32] | target[3] | reserved[3]
constexpr int TREE_PUBLIC_STATE_LEN = TREE_NQ_MAX + TL + 6;
static_assert(TREE_BLOB_LEN == 3961, "keep in sync with mjmpc_amd/models/compile_tree.py::TREE_LAYOUT");

}  // namespace mjmpc

// Constant block of a compiled arm, as produced by mjmpc_amd/models/compile.py (ARM_LAYOUT).
// Offsets are in scalars; per-link fields are stored [component][lane] with 8 lanes.
#pragma once

namespace mjmpc {

constexpr int LANES = 8;        // lanes per particle
constexpr int MAX_LINKS = 7;

enum ArmOffset : int {
    O_OFF = 0,                       // 3 x 8
    O_AXIS = O_OFF + 24,             // 3 x 8
    O_MASS = O_AXIS + 24,            // 8
    O_COM = O_MASS + 8,              // 3 x 8
    O_INERTIA = O_COM + 24,          // 6 x 8   xx yy zz xy xz yz
    O_ARMATURE = O_INERTIA + 48,
    O_DAMPING = O_ARMATURE + 8,
    O_RANGE_LO = O_DAMPING + 8,
    O_RANGE_HI = O_RANGE_LO + 8,
    O_LIMITED = O_RANGE_HI + 8,
    O_GEAR = O_LIMITED + 8,
    O_CTRL_LO = O_GEAR + 8,
    O_CTRL_HI = O_CTRL_LO + 8,
    O_DOF_INVW = O_CTRL_HI + 8,
    O_NV = O_DOF_INVW + 8,
    O_TIMESTEP,
    O_FRAME_SKIP,
    O_SITE_LINK,
    O_SITE_POS,                      // 3
    O_N_SPHERE = O_SITE_POS + 3,
    O_SPH_LINK,
    O_SPH_POS,                       // 3
    O_SPH_R = O_SPH_POS + 3,
    O_SPH_MARGIN,
    O_SPH_INVW,
    O_PLANE_N,                       // 3
    O_PLANE_D = O_PLANE_N + 3,
    O_SOL_K,
    O_SOL_B,
    O_SOL_DMIN,
    O_SOL_DMAX,
    O_SOL_WIDTH,
    O_SOL_MID,
    O_SOL_POWER,
    O_GRAVITY,                       // 3
    // round 6: what the EXTENDED-JOINT build of the arm kernels (arm_rollout_xj.hip) reads on top - slide joints, dry friction
    O_JTYPE = O_GRAVITY + 3,         // 8: 0 hinge, 1 slide
    O_FLOSS = O_JTYPE + 8,           // 8: dof_frictionloss
    O_FLOSS_D = O_FLOSS + 8,         // 8: D = 1 / R of the dof's friction-loss row
    O_FLOSS_B = O_FLOSS_D + 8,       // b of the rows' reference acceleration -b v
    O_NU,                            // motors (they drive dofs 0 .. nu - 1); host side only
    ARM_BLOB_LEN
};

static_assert(ARM_BLOB_LEN == 255, "keep in sync with mjmpc_amd/models/compile.py::ARM_LAYOUT");

}  // namespace mjmpc

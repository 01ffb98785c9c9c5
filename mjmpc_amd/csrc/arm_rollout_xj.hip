// The EXTENDED-JOINT build of the arm kernels (round 6): arm_rollout.hip compiled a second time with slide joints and dry
// friction (friction-loss rows) compiled in - the reference's classic-control models (examples/configs/classic_control/
// cartpole*.yml: a cart on a slide joint, frictionloss on both joints) on the serial-chain kernel instead of the general
// tree kernel's 16-lane instruction stream (cart-pole MPPI 4096 x 32 f64: profiles/r06_cartpole_*).  A build of its own, so that
// the code of the kernels every other arm model runs does not change; its entry points are launch_arm_rollout_xj /
// launch_arm_mppi_finish_xj (arm_rollout.h), chosen per engine by capi.hip.  The flag-synchronised shapes are left out of it:
// such models run the barrier shapes (SOLO / DUO).
#define MJMPC_ARM_XJ 1
#define ARM_NO_FLAGS_CODE 1
#include "arm_rollout.hip"

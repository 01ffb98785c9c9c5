// The tree rollout kernel's instantiations for models with ELLIPTIC friction cones (GEN = 3, round 5: MJCF <option
// cone="elliptic" impratio>) as a translation unit of their own: the same source, compiled beside the others.  Every
// other model runs the kernels it had.
#define TREE_CONE_TU
#include "tree_rollout.hip"

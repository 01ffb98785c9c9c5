// The 16-lane dense instantiations of the tree rollout kernel (models of up to 16 dofs: the reference's swimmer.xml and
// half_cheetah.xml) as a translation unit of their own: the same source, compiled with the iterative-ILP scheduling
// strategy (mjmpc_amd/build.py: PER_SOURCE_FLAGS) - these kernels run one wavefront per SIMD, where instruction-level
// parallelism is all the scheduler can offer.
#define TREE_DENSE_TU
#include "tree_rollout.hip"

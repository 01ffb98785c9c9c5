#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

// Optional fusions into the rollout launch (all pointers may be null):
//   filt    float64[3]  apply the recursive noise filter of control_utils.py:32-33 to `noise` on the fly
//                       (then `noise` holds the raw, unfiltered samples)
//   gseq    float64[H]  with q0_out: q0_out[p] = sum_t gseq[t] * cost[p][t]  (= cost_to_go(...)[:,0])
//   clw     float64[(2nv+7)][A]  mode "closed_loop_linear" (gym_env_wrapper.py:135-136): the nominal action of
//                       a step is clw^T [obs; 1] with obs the observation BEFORE the step; `mean` is ignored
//   shard_size          > 0: particles [k*shard_size, (k+1)*shard_size) use model block k (dynamics randomization:
//                       every shard of the reference's worker pool simulates its own perturbed model)
//   state_shard_size    > 0: likewise for the start state: shard k starts from state vector k
struct RolloutFusion {
    long shard_size = 0;
    long state_shard_size = 0;
    const double* clw = nullptr;
    const double* filt = nullptr;
    const double* gseq = nullptr;
    double* q0_out = nullptr;
};

// Fused (particles x horizon x frame_skip) rollout of a compiled arm; see arm_rollout.hip.
template <typename T>
hipError_t launch_arm_rollout(const T* model, const double* state, long P, int H, int A, const double* mean,
                              const T* noise, T* cost, T* act, T* obs, T* nobs, double* state_out,
                              unsigned* diag, hipStream_t stream, RolloutFusion fuse = RolloutFusion());

}  // namespace mjmpc

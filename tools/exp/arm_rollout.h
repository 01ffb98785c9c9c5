#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

// Fused (particles x horizon x frame_skip) rollout of a compiled arm; see arm_rollout.hip.
template <typename T>
hipError_t launch_arm_rollout(const T* model, const double* state, long P, int H, int A, const double* mean,
                              const T* noise, T* cost, T* act, T* obs, T* nobs, double* state_out,
                              unsigned* diag, hipStream_t stream);

}  // namespace mjmpc

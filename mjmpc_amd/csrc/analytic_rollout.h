#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

enum AnalyticKind : int { ANALYTIC_PENDULUM = 0, ANALYTIC_LQR = 1 };

// Pendulum: prm = [max_speed, max_torque, dt, g, m, l], state = [th, thdot], n = 2 (obs dim 3), m = 1.
// LQR:      prm = [A | B | Q | R] row-major, state = x[n], obs = x.
// closed != 0: `mean` is the (d_obs+1, m) weight matrix of mode 'closed_loop_linear'.
template <typename T>
hipError_t launch_analytic_rollout(int kind, const double* prm, int n, int m, const double* state, long P, int H,
                                   const double* mean, const T* noise, T* cost, T* act, T* obs, T* nobs,
                                   hipStream_t s, int closed = 0);

}  // namespace mjmpc

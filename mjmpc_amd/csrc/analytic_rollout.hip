// Rollout kernel for the two analytic numpy environments of the reference:
//   PendulumEnv (mjmpc/envs/basic/pendulum.py:33-50,62-64)   and   LQREnv (mjmpc/envs/basic/lqr.py:31-35),
// same loop and output layout as GymEnvWrapper.rollout (mjmpc/envs/gym_env_wrapper.py:125-153).
// These envs are a few flops per step, so the mapping is the plain one: one particle per lane.  They
// exist because the reference itself can run them (no MuJoCo needed): golden vectors captured from
// the reference pin this GPU path end to end (tests/test_analytic_envs_gpu.py).
#include <hip/hip_runtime.h>

#include "analytic_rollout.h"

namespace mjmpc {
namespace {

constexpr int MAXN = 8;     // LQR state / action dimension limit

// numpy's floating "%" (sign of the divisor), as used by pendulum.angle_normalize
__device__ __forceinline__ double pymod(double x, double m) {
    double r = fmod(x, m);
    if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m;
    return r;
}

template <typename T>
__global__ void pendulum_rollout_kernel(const double* __restrict__ prm, const double* __restrict__ state, long P, int H,
                                        const double* __restrict__ mean, const T* __restrict__ noise,
                                        T* __restrict__ cost, T* __restrict__ act, T* __restrict__ obs,
                                        T* __restrict__ nobs, int closed) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double max_speed = prm[0], max_torque = prm[1], dt = prm[2], g = prm[3], m = prm[4], l = prm[5];
    const double PI = 3.141592653589793;
    double th = state[0], thdot = state[1];
    double o0 = cos(th), o1 = sin(th), o2 = thdot;
    for (int t = 0; t < H; ++t) {
        // closed_loop_linear (gym_env_wrapper.py:135-136): mean is the (d_obs+1, 1) weight matrix, action = W^T [obs; 1]
        const double mu = closed ? mean[0] * o0 + mean[1] * o1 + mean[2] * o2 + mean[3] : mean[t];
        double u = mu + (noise ? (double)noise[p * H + t] : 0.0);
        if (act) act[p * H + t] = (T)u;
        const double uc = fmin(fmax(u, -max_torque), max_torque);
        const double an = pymod(th + PI, 2 * PI) - PI;
        const double c = an * an + .1 * thdot * thdot + .001 * (uc * uc);
        double nthdot = thdot + (-3 * g / (2 * l) * sin(th + PI) + 3. / (m * l * l) * uc) * dt;
        th = th + nthdot * dt;
        thdot = fmin(fmax(nthdot, -max_speed), max_speed);
        cost[p * H + t] = (T)c;
        const long o = (p * H + t) * 3;
        if (obs) { obs[o] = (T)o0; obs[o + 1] = (T)o1; obs[o + 2] = (T)o2; }
        o0 = cos(th); o1 = sin(th); o2 = thdot;
        if (nobs) { nobs[o] = (T)o0; nobs[o + 1] = (T)o1; nobs[o + 2] = (T)o2; }
    }
}

// prm = [A (n x n) | B (n x m) | Q (n x n) | R (m x m)], row-major
template <typename T>
__global__ void lqr_rollout_kernel(const double* __restrict__ prm, const double* __restrict__ state, int n, int m, long P,
                                   int H, const double* __restrict__ mean, const T* __restrict__ noise,
                                   T* __restrict__ cost, T* __restrict__ act, T* __restrict__ obs,
                                   T* __restrict__ nobs, int closed) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double *Am = prm, *Bm = prm + n * n, *Qm = Bm + n * m, *Rm = Qm + n * n;
    double x[MAXN], u[MAXN], y[MAXN];
    for (int i = 0; i < n; ++i) x[i] = state[i];
    for (int t = 0; t < H; ++t) {
        for (int a = 0; a < m; ++a) {
            double mu;
            if (closed) {                   // W^T [x; 1], W = mean (n+1, m)
                mu = 0.0;
                for (int i = 0; i < n; ++i) mu += mean[i * m + a] * x[i];
                mu += mean[n * m + a];
            } else {
                mu = mean[t * m + a];
            }
            u[a] = mu + (noise ? (double)noise[(p * H + t) * m + a] : 0.0);
            if (act) act[(p * H + t) * m + a] = (T)u[a];
        }
        // cost = x'Qx + u'Ru in numpy's evaluation order: (x.T.dot(Q)).dot(x)
        double c1 = 0.0, c2 = 0.0;
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += x[i] * Qm[i * n + j];
            c1 += s * x[j];
        }
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += u[i] * Rm[i * m + j];
            c2 += s * u[j];
        }
        const long o = (p * H + t) * n;
        if (obs) for (int i = 0; i < n; ++i) obs[o + i] = (T)x[i];
        for (int i = 0; i < n; ++i) {
            double s1 = 0.0, s2 = 0.0;
            for (int j = 0; j < n; ++j) s1 += Am[i * n + j] * x[j];
            for (int j = 0; j < m; ++j) s2 += Bm[i * m + j] * u[j];
            y[i] = s1 + s2;
        }
        for (int i = 0; i < n; ++i) x[i] = y[i];
        cost[p * H + t] = (T)(c1 + c2);
        if (nobs) for (int i = 0; i < n; ++i) nobs[o + i] = (T)x[i];
    }
}

}  // namespace

template <typename T>
hipError_t launch_analytic_rollout(int kind, const double* prm, int n, int m, const double* state, long P, int H,
                                   const double* mean, const T* noise, T* cost, T* act, T* obs, T* nobs,
                                   hipStream_t s, int closed) {
    if (P <= 0 || H <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((P + 255) / 256);
    if (kind == ANALYTIC_PENDULUM) {
        hipLaunchKernelGGL(pendulum_rollout_kernel<T>, dim3(grid), dim3(256), 0, s, prm, state, P, H, mean, noise, cost, act,
                           obs, nobs, closed);
    } else if (kind == ANALYTIC_LQR) {
        if (n < 1 || n > MAXN || m < 1 || m > MAXN) return hipErrorInvalidValue;
        hipLaunchKernelGGL(lqr_rollout_kernel<T>, dim3(grid), dim3(256), 0, s, prm, state, n, m, P, H, mean, noise, cost,
                           act, obs, nobs, closed);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template hipError_t launch_analytic_rollout<float>(int, const double*, int, int, const double*, long, int, const double*,
                                                   const float*, float*, float*, float*, float*, hipStream_t, int);
template hipError_t launch_analytic_rollout<double>(int, const double*, int, int, const double*, long, int,
                                                    const double*, const double*, double*, double*, double*, double*,
                                                    hipStream_t, int);

}  // namespace mjmpc

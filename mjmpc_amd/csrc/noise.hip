// On-device replacement for control_utils.generate_noise (reference mjmpc/utils/control_utils.py:24-34)
// in "performance mode": same distribution (N(0, cov) per step, then the in-place recursive 3-tap
// filter along the horizon), different bit stream (counter-based Philox4x32-10 instead of the legacy
// MT19937 stream, which is serial).  Bit-identical noise is the host path
// (mjmpc_amd.control.control_utils.generate_noise + upload).
//
// Every normal is a pure function of
// (seed, step offset, particle, channel, t/2), so a channel of a correlated sample,
// eps[a] = sum_{b<=a} L[a][b] z[b], is recomputed locally instead of exchanged between threads.
#include <hip/hip_runtime.h>

#include "update.h"

namespace mjmpc {
namespace {

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

// two independent standard normals from one Philox4x32-10 block (53-bit uniforms, Box-Muller)
__device__ __forceinline__ void normal_pair(unsigned long long seed, unsigned long long offset, unsigned long long chan,
                                            unsigned pair, double& z0, double& z1) {
    unsigned c0 = (unsigned)chan, c1 = (unsigned)(chan >> 32), c2 = pair, c3 = (unsigned)offset;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32) ^ (unsigned)(offset >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // 53-bit uniforms; the transcendental part of Box-Muller runs in single precision (relative error
    // ~1e-7 on a random variate: statistically invisible, 4x cheaper than the f64 library calls)
    const double two53 = 1.0 / 9007199254740992.0;
    const unsigned long long a = (((unsigned long long)c0 << 32) | c1) >> 11, b = (((unsigned long long)c2 << 32) | c3) >> 11;
    const float u1 = (float)(((double)a + 0.5) * two53), u2 = (float)((double)b * two53);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    z0 = (double)(r * c);
    z1 = (double)(r * s);
}

// pass 1: coloured normals, one thread per (particle, channel, t-pair)
template <typename T>
__global__ void noise_kernel(T* __restrict__ noise, long P, int H, int A, const double* __restrict__ chol,
                             unsigned long long seed, unsigned long long offset, long particle_offset,
                             const long long* __restrict__ d_step, int diag_only) {
    if (d_step) offset += (unsigned long long)*d_step;      // step counter kept on the device (graph replay)
    const int H2 = (H + 1) / 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P * H2 * A) return;
    const int a = (int)(gid % A);
    const int t2 = (int)((gid / A) % H2);
    const long p = gid / ((long)A * H2);
    double x0 = 0.0, x1 = 0.0;
    for (int b = diag_only ? a : 0; b <= a; ++b) {
        const double l = chol[a * A + b];
        if (l == 0.0) continue;
        double z0, z1;
        normal_pair(seed, offset, (unsigned long long)((p + particle_offset) * A + b), (unsigned)t2, z0, z1);
        x0 += l * z0;
        x1 += l * z1;
    }
    const int t = 2 * t2;
    noise[(p * H + t) * A + a] = (T)x0;
    if (t + 1 < H) noise[(p * H + t + 1) * A + a] = (T)x1;
}

// pass 2: eps[t] = b0 eps[t] + b1 eps[t-1] + b2 eps[t-2] for t >= 2, in place, in float64
// (control_utils.py:32-33: t-1 and t-2 are already filtered).  One thread per (particle, channel).
template <typename T>
__global__ void filter_kernel(T* __restrict__ noise, long P, int H, int A, const double* __restrict__ coeffs) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P * A) return;
    const long p = gid / A;
    const int a = (int)(gid % A);
    const double b0 = coeffs[0], b1 = coeffs[1], b2 = coeffs[2];
    if (b0 == 1.0 && b1 == 0.0 && b2 == 0.0) return;
    T* row = noise + p * H * A + a;
    double e2 = H > 0 ? (double)row[0] : 0.0, e1 = H > 1 ? (double)row[A] : 0.0;
    for (int t = 2; t < H; ++t) {
        const double v = b0 * (double)row[(long)t * A] + b1 * e1 + b2 * e2;
        row[(long)t * A] = (T)v;
        e2 = e1;
        e1 = v;
    }
}

}  // namespace

template <typename T>
hipError_t sample_noise(T* noise, long P, int H, int A, const double* chol, const double* coeffs,
                        unsigned long long seed, unsigned long long offset, long particle_offset, const long long* d_step,
                        hipStream_t s, int diag_only) {
    if (P <= 0 || H <= 0) return hipSuccess;
    const long n = P * A * ((H + 1) / 2), m = P * A;
    hipLaunchKernelGGL(noise_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, noise, P, H, A, chol, seed,
                       offset, particle_offset, d_step, diag_only);
    if (coeffs)     // null: leave the samples raw (the rollout kernel can apply the filter on the fly)
        hipLaunchKernelGGL(filter_kernel<T>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, noise, P, H, A, coeffs);
    return hipGetLastError();
}

template <typename T>
hipError_t filter_noise(T* noise, long P, int H, int A, const double* coeffs, hipStream_t s) {
    const long m = P * A;
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(filter_kernel<T>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, noise, P, H, A, coeffs);
    return hipGetLastError();
}
template hipError_t filter_noise<float>(float*, long, int, int, const double*, hipStream_t);
template hipError_t filter_noise<double>(double*, long, int, int, const double*, hipStream_t);

template hipError_t sample_noise<float>(float*, long, int, int, const double*, const double*, unsigned long long,
                                        unsigned long long, long, const long long*, hipStream_t, int);
template hipError_t sample_noise<double>(double*, long, int, int, const double*, const double*, unsigned long long,
                                         unsigned long long, long, const long long*, hipStream_t, int);

}  // namespace mjmpc

// On-device replacement for control_utils.generate_noise (reference mjmpc/utils/control_utils.py:24-34)
// in "performance mode": same distribution (N(0, cov) per step, then the in-place recursive 3-tap
// filter along the horizon), different bit stream (counter-based Philox4x32-10 instead of the legacy
// MT19937 stream, which is serial).  Bit-identical noise is the host path
// (mjmpc_amd.control.control_utils.generate_noise + upload).
//
// Every normal is a pure function of
// (seed, step offset, particle, channel, t/4), so a channel of a correlated sample,
// eps[a] = sum_{b<=a} L[a][b] z[b], is recomputed locally instead of exchanged between threads.
#include <hip/hip_runtime.h>

#include "noise_device.h"
#include "update.h"

namespace mjmpc {
namespace {

// pass 1: coloured normals, one thread per (particle, channel, t-quad)
template <typename T>
__global__ void noise_kernel(T* __restrict__ noise, long P, int H, int A, const double* __restrict__ chol,
                             unsigned long long seed, unsigned long long offset, long particle_offset,
                             const long long* __restrict__ d_step, int diag_only) {
    if (d_step) offset += (unsigned long long)*d_step;      // step counter kept on the device (graph replay)
    noise_element<T>(noise, (long)blockIdx.x * blockDim.x + threadIdx.x, P, H, A, chol, seed, offset, particle_offset,
                     diag_only);
}

// Full (lower-triangular) colouring: one thread per (particle, t-quad) draws the A independent normal quadruples ONCE and
// forms all A channels from them - the per-element kernel above would draw z_b again for every channel a >= b
// (A (A+1) / 2 Philox blocks instead of A; 56 -> 16 us at 16384 x 32 x 7).  Same draws, same order of summation.
constexpr int NOISE_MAXA = 8;      // loops are unrolled to this bound so that the draws stay in registers
template <typename T>
__global__ void noise_full_kernel(T* __restrict__ noise, long P, int H, int A, const double* __restrict__ chol,
                                  unsigned long long seed, unsigned long long offset, long particle_offset,
                                  const long long* __restrict__ d_step) {
    if (d_step) offset += (unsigned long long)*d_step;
    const int H4 = (H + 3) / 4;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = gid < P * H4;         // (idle lanes of the last workgroup stay for the staged store)
    const int t4 = (int)(gid % H4);
    const long p = gid / H4;
    float z[NOISE_MAXA][4];
#pragma unroll
    for (int b = 0; b < NOISE_MAXA; ++b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) z[b][k] = 0.0f;
        if (b < A && live) normal_quad(seed, offset, (unsigned long long)((p + particle_offset) * A + b), (unsigned)t4, z[b]);
    }
    const int t = 4 * t4;
    // H a multiple of 4: the thread's 4 x A samples are one contiguous run of the output and the workgroup's 64 runs
    // follow each other, so they go out through an LDS tile as full-width rows (direct stores are 8-byte words 4 A
    // scalars apart: 26 -> ~10 us at 16384 x 32 x 7)
    __shared__ T tile[64 * (4 * NOISE_MAXA + 1)];
    const bool staged = (H & 3) == 0;
    const int run = 4 * A, pad = run + 1;
#pragma unroll
    for (int a = 0; a < NOISE_MAXA; ++a) {
        if (a < A) {
            double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const double l = chol[a * A + b];
                if (l != 0.0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] += l * (double)z[b][k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (staged) tile[threadIdx.x * pad + k * A + a] = (T)x[k];
                else if (live && t + k < H) noise[(p * H + t + k) * A + a] = (T)x[k];
            }
        }
    }
    if (!staged) return;
    __builtin_amdgcn_wave_barrier();        // (the workgroup is one wavefront)
    const long first = (long)blockIdx.x * 64 * run, total = P * (long)H * A;
    for (int i = threadIdx.x; i < 64 * run; i += 64) {
        const int th = i / run, off = i - th * run;
        if (first + i < total) noise[first + i] = tile[th * pad + off];
    }
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
    const long n = P * A * ((H + 3) / 4), m = P * A;
    if (!diag_only && A <= NOISE_MAXA) {
        const long nf = P * ((H + 3) / 4);
        hipLaunchKernelGGL(noise_full_kernel<T>, dim3((unsigned)((nf + 63) / 64)), dim3(64), 0, s, noise, P, H, A, chol, seed,
                           offset, particle_offset, d_step);
    } else {
        hipLaunchKernelGGL(noise_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, noise, P, H, A, chol, seed,
                           offset, particle_offset, d_step, diag_only);
    }
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
// x[row][:] <- x[row][:] B  (rows of A standard normals -> N(0, cov) samples): numpy's multivariate_normal colours
// its standard-normal stream with B = sqrt(s)[:, None] * v from the SVD of cov (control_utils.py:30 reaches that
// through np.random.multivariate_normal); the seed-identical MT19937 mode applies the host-computed B here.
// One thread per row, fixed summation order.
template <typename T>
__global__ void color_rows_kernel(T* __restrict__ x, long rows, int A, const double* __restrict__ B) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double in[64], out[64];
    for (int i = 0; i < A; ++i) in[i] = (double)x[r * A + i];
    for (int j = 0; j < A; ++j) {
        double sacc = 0.0;
        for (int i = 0; i < A; ++i) sacc = fma(in[i], B[i * A + j], sacc);
        out[j] = sacc;
    }
    for (int j = 0; j < A; ++j) x[r * A + j] = (T)out[j];
}

template <typename T>
hipError_t color_rows(T* x, long rows, int A, const double* B, hipStream_t s) {
    if (A < 1 || A > 64) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(color_rows_kernel<T>, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, s, x, rows, A, B);
    return hipGetLastError();
}
template hipError_t color_rows<float>(float*, long, int, const double*, hipStream_t);
template hipError_t color_rows<double>(double*, long, int, const double*, hipStream_t);

template hipError_t filter_noise<float>(float*, long, int, int, const double*, hipStream_t);
template hipError_t filter_noise<double>(double*, long, int, int, const double*, hipStream_t);

template hipError_t sample_noise<float>(float*, long, int, int, const double*, const double*, unsigned long long,
                                        unsigned long long, long, const long long*, hipStream_t, int);
template hipError_t sample_noise<double>(double*, long, int, int, const double*, const double*, unsigned long long,
                                         unsigned long long, long, const long long*, hipStream_t, int);

}  // namespace mjmpc

// Device-side body of the Philox sampler (noise.hip), shared with the fused MPPI update (update.hip), whose
// first launch carries extra workgroups that draw the NEXT control step's samples while the partial sums run.
#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

// FOUR independent standard normals from one Philox4x32-10 block: every 32-bit word is one uniform, two Box-Muller
// pairs.  The transcendental part runs on the hardware's single-precision units - v_log_f32, v_sqrt_f32 and
// v_sin_f32 / v_cos_f32, whose argument is in REVOLUTIONS, i.e. the uniform itself, no range reduction - (relative error
// ~1e-6 on a random variate: statistically invisible; a draw costs ~1/3 of round 2's 53-bit / libm version, which matters
// where the rollout kernel draws its own samples).  Radius from (w + 0.5) 2^-32 in (0, 1]: |z| <= 6.8.  The normals ARE
// single-precision numbers; callers widen them.
__device__ __forceinline__ void normal_quad(unsigned long long seed, unsigned long long offset, unsigned long long chan,
                                            unsigned quad, float* z) {
    unsigned c0 = (unsigned)chan, c1 = (unsigned)(chan >> 32), c2 = quad, c3 = (unsigned)offset;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32) ^ (unsigned)(offset >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float two32 = 2.3283064365386963e-10f;                        // 2^-32
    const float ln2x2 = 1.3862943611198906f;                            // 2 ln 2
    const float u0 = ((float)c0 + 0.5f) * two32, u1 = ((float)c2 + 0.5f) * two32;     // (small words, the tail, are exact)
    const float r0 = __builtin_amdgcn_sqrtf(-ln2x2 * __builtin_amdgcn_logf(u0));      // sqrt(-2 ln u), v_log_f32 = log2
    const float r1 = __builtin_amdgcn_sqrtf(-ln2x2 * __builtin_amdgcn_logf(u1));
    const float a0 = (float)c1 * two32, a1 = (float)c3 * two32;                         // angle in revolutions, [0, 1]
    z[0] = r0 * __builtin_amdgcn_cosf(a0);
    z[1] = r0 * __builtin_amdgcn_sinf(a0);
    z[2] = r1 * __builtin_amdgcn_cosf(a1);
    z[3] = r1 * __builtin_amdgcn_sinf(a1);
}

// coloured normals of one (particle, channel, t-quad): gid enumerates P x ceil(H/4) x A
template <typename T>
__device__ __forceinline__ void noise_element(T* __restrict__ noise, long gid, long P, int H, int A,
                                              const double* __restrict__ chol, unsigned long long seed,
                                              unsigned long long offset, long particle_offset, int diag_only) {
    const int H4 = (H + 3) / 4;
    if (gid >= P * H4 * A) return;
    const int a = (int)(gid % A);
    const int t4 = (int)((gid / A) % H4);
    const long p = gid / ((long)A * H4);
    double x[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = diag_only ? a : 0; b <= a; ++b) {
        const double l = chol[a * A + b];
        if (l == 0.0) continue;
        float z[4];
        normal_quad(seed, offset, (unsigned long long)((p + particle_offset) * A + b), (unsigned)t4, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] += l * (double)z[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * t4 + k < H) noise[(p * H + 4 * t4 + k) * A + a] = (T)x[k];
}

// what the fused update needs to draw the next step's raw samples (noise == nullptr: nothing to draw)
struct NextNoise {
    void* noise;
    const double* chol;
    unsigned long long seed, offset;
    long particle_offset;
    const long long* d_step;
    int diag_only;
};

}  // namespace mjmpc

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

// coloured normals of one (particle, channel, t-pair): gid enumerates P x ceil(H/2) x A
template <typename T>
__device__ __forceinline__ void noise_element(T* __restrict__ noise, long gid, long P, int H, int A,
                                              const double* __restrict__ chol, unsigned long long seed,
                                              unsigned long long offset, long particle_offset, int diag_only) {
    const int H2 = (H + 1) / 2;
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

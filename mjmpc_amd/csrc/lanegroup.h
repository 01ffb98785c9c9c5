// Cross-lane primitives for 8-lane particle groups on CDNA4 (wave64).
//
// One particle = 8 lanes (one per link, link 7 a spare), 8 particles per wavefront, everything by DPP:
// row shifts for the scans along the kinematic chain, quad_perm + bank-masked shifts for broadcasts,
// rotations for the 8-lane sum.  No LDS traffic, no ds_bpermute.
#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

template <int CTRL, int BANK>
__device__ __forceinline__ float dpp(float old, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xF, BANK, false));
}
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp(double old, double x) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xF, BANK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}

// DPP whose source lane is valid for every lane (quad_perm): no `old` operand to preserve
template <int CTRL>
__device__ __forceinline__ float dpp_all(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_all(double x) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------
// Lane layout.  A DPP row is 16 lanes = two particles, INTERLEAVED: lane = 2 * link + particle (the two particles
// of a row alternate).  A shift by S links is then a row shift by 2S lanes, which never crosses from
// one particle into the other and falls off the END OF THE ROW exactly at the chain boundary: DPP's own
// bound_ctrl (zero) / `old` (fill) semantics do the masking, so scans need no v_cndmask and - in f32 -
// fold into a single v_add_f32_dpp.  (A blocked layout, lane = 8 * particle + link, saves one DPP per broadcast
// but needs a select after every scan step: measured slower in round 1 and removed.)

// zero-filling DPP (bound_ctrl): lanes whose source falls outside the row read 0
template <int CTRL>
__device__ __forceinline__ float dpp_zero(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_zero(double x) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int lane_link(int lane) { return (lane >> 1) & 7; }
__device__ __forceinline__ int lane_slot(int lane) { return ((lane >> 4) << 1) | (lane & 1); }
// wave lane holding `link` of the same particle as `lane`
__device__ __forceinline__ int lane_of_link(int lane, int link) { return (lane & 0x31) | (link << 1); }

// lane(link i) <- x[link i - S], `fill` where i < S
template <int S, typename T>
__device__ __forceinline__ T shr(T x, T fill, int) {
    return dpp<0x110 + 2 * S, 0xF>(fill, x);
}
template <int S, typename T>
__device__ __forceinline__ T shl(T x, T fill, int) {
    return dpp<0x100 + 2 * S, 0xF>(fill, x);
}
template <int S, typename T>
__device__ __forceinline__ T shr0(T x, int) { return dpp_zero<0x110 + 2 * S>(x); }
template <int S, typename T>
__device__ __forceinline__ T shl0(T x, int) { return dpp_zero<0x100 + 2 * S>(x); }
// row_shl by S links; lanes past the end of the chain read 0
template <int S, typename T>
__device__ __forceinline__ T shl_raw(T x) { return dpp_zero<0x100 + 2 * S>(x); }

// every lane of the particle <- x[link K]: pair-broadcast inside the quad holding links {2q, 2q+1}, then
// copy that quad over the row with two bank-masked shifts
template <int K, typename T>
__device__ __forceinline__ T bcast(T x) {
    constexpr int s = 2 * (K & 1);
    constexpr int QP = s | ((s + 1) << 2) | (s << 4) | ((s + 1) << 6);
    constexpr int Q = K >> 1;
    T t = dpp_all<QP>(x);
    if constexpr (Q == 0) { t = dpp<0x114, 0x2>(t, t); return dpp<0x118, 0xC>(t, t); }
    else if constexpr (Q == 1) { t = dpp<0x104, 0x1>(t, t); return dpp<0x118, 0xC>(t, t); }
    else if constexpr (Q == 2) { t = dpp<0x114, 0x8>(t, t); return dpp<0x108, 0x3>(t, t); }
    else { t = dpp<0x104, 0x4>(t, t); return dpp<0x108, 0x3>(t, t); }
}
// sum over the 8 links of the particle, result in every lane
template <typename T>
__device__ __forceinline__ T gsum(T x) {
    x += dpp_all<0x4E>(x);                          // quad_perm [2,3,0,1]: the other link of the quad
    x += dpp_all<0x124>(x);                         // row_ror:4
    x += dpp_all<0x128>(x);                         // row_ror:8
    return x;
}


// inclusive prefix / suffix sums over the links of a particle
template <typename T>
__device__ __forceinline__ T psum(T x, int l8) {
    x += shr0<1>(x, l8);
    x += shr0<2>(x, l8);
    x += shr0<4>(x, l8);
    return x;
}
template <typename T>
__device__ __forceinline__ T ssum(T x, int l8) {
    x += shl0<1>(x, l8);
    x += shl0<2>(x, l8);
    x += shl0<4>(x, l8);
    return x;
}

// reciprocal: hardware seed + Newton steps (1 for f32, 2 for f64) - avoids the IEEE divide expansion
__device__ __forceinline__ float rcp_(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double rcp_(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
// pivot reciprocal of the dense LDL^T (arm_rollout.hip, struct Dense): v_rcp_f64 is good to 4.6e-8 (measured on gfx950), one
// Newton step brings it to 2.2e-15 - enough for a factorisation whose backward error is O(eps) anyway
__device__ __forceinline__ float rcp_fast(float x) { return rcp_(x); }
__device__ __forceinline__ double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ void sincos_(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ __forceinline__ void sincos_(double x, double& s, double& c) { sincos(x, &s, &c); }
// sin / cos of a SMALL angle (|x| <= 0.25) by Taylor series in x^2: the joint angle advances by
// h * qvel per substep, so the rotation is updated by angle addition instead of a full sincos
__device__ __forceinline__ void sincos_small(float x, float& s, float& c) {
    const float z = x * x;
    s = x * fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
    c = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
}
__device__ __forceinline__ void sincos_small(double x, double& s, double& c) {
    const double z = x * x;
    double ps = 1.0 / 6227020800.0;                 // 1/13!
    ps = fma(z, ps, -1.0 / 39916800.0);
    ps = fma(z, ps, 1.0 / 362880.0);
    ps = fma(z, ps, -1.0 / 5040.0);
    ps = fma(z, ps, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    s = x * fma(z, ps, 1.0);
    double pc = -1.0 / 87178291200.0;               // -1/14!
    pc = fma(z, pc, 1.0 / 479001600.0);
    pc = fma(z, pc, -1.0 / 3628800.0);
    pc = fma(z, pc, 1.0 / 40320.0);
    pc = fma(z, pc, -1.0 / 720.0);
    pc = fma(z, pc, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    c = fma(z, pc, 1.0);
}
// 1 / sqrt(x), x > 0: hardware seed + Newton steps (as rcp_) - the contact normals need the length AND its reciprocal, which
// is one of these and a product instead of an IEEE square root and a reciprocal
__device__ __forceinline__ float rsqrt_(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    return r * fmaf(-0.5f * x * r, r, 1.5f);
}
__device__ __forceinline__ double rsqrt_(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r * fma(-0.5 * x * r, r, 1.5);
}
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }

}  // namespace mjmpc

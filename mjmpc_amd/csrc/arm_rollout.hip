// Fused rollout kernel for a serial hinge arm (reacher_7dof-v0): the whole
//   for b in particles: for t in horizon: env.step(mean[t] + noise[b,t])
// double loop of GymEnvWrapper.rollout (reference mjmpc/envs/gym_env_wrapper.py:125-153) with
// Reacher7DOFEnv.step (mjmpc/envs/basic/reacher_env.py:29-39) and MuJoCo's mj_step inlined,
// one launch per control iteration.
//
// Execution model (CDNA4, wave64): ONE PARTICLE = 8 LANES, lane l8 = link / dof index
// (lane 7 is a zero-mass spare), 8 particles per wavefront, one wavefront per workgroup.
// All per-particle state (q, v, frames, spatial inertias, the 7x7 mass matrix row of "my" dof)
// lives in VGPRs; links talk through DPP row shifts (lanegroup.h).  Everything is expressed in
// world coordinates about the world origin, so tree recursions become plain prefix / suffix
// sums over the lane group:
//   forward kinematics   = inclusive scan of affine transforms (3 DPP steps)
//   velocities, vel.-product accelerations = prefix sums of spatial vectors
//   RNE forces, composite inertias         = suffix sums
//   M[i][i+s] = S_i . (Ic_{i+s} S_{i+s})   = 7 shifted dot products  -> LDS 8x8 transpose
//   (M + E) x = r                          = dense LDL^T per solver lane, operands through the LDS tile
// The soft-constraint problem (joint limits + sphere/plane contact, all frictionless rows) is
// solved by a primal active-set Newton iteration: with the active set fixed the objective is
// quadratic, so each iteration is one 7x7 solve; it stops when the active set reproduces itself,
// which is the exact minimiser MuJoCo's Newton solver converges to.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "arm_model.h"
#include "arm_rollout.h"
#include "lanegroup.h"
#include "noise_device.h"

namespace mjmpc {
namespace {

// cap of the active-set iteration (8 rows: 7 limits and the contact).  f64 has never reached 12 (the sets settle in 1-3
// iterations); f32 - where a row within rounding of its switching point can flip back and forth - gets 32: round 4's f32 lines
// showed 8 cap hits in 8.4 x 10^6 particle-substeps of DMD-MPC 65536 x 64 at 12
template <typename T>
constexpr int newton_maxit() { return sizeof(T) == 4 ? 32 : 12; }

// The compiled-model block (arm_model.h) is staged once per workgroup in LDS; every phase of a
// substep reads the constants it needs from there (lane l8 reads slot l8 of a per-link field, all
// lanes read the same word of a global field) instead of pinning ~25 VGPR pairs and ~35 SGPR pairs
// for the whole rollout.  PHASE() is a compiler-only fence that keeps those reads inside their phase.
#define PHASE() asm volatile("" ::: "memory")
// A workgroup is ONE wavefront and the DS unit executes a wave's LDS instructions in order, so lanes
// exchanging data through LDS need no s_barrier (and none of the vmcnt(0) drain __syncthreads implies,
// which would stall every substep on the global stores of the previous env step): a wavefront-scope
// fence keeps the compiler from reordering the accesses and costs no instruction.
// The sched_barrier keeps the instruction scheduler from hoisting the next phase's work across the
// hand-over (what s_barrier did implicitly): without it the kernel needs ~50 more VGPRs.
#define LDS_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

// REG = true (DUO kernels, which have the registers to spare): my link's 25 per-link constants are read once and
// stay in VGPRs - a lone wavefront otherwise sits out the LDS latency at the head of every phase.  Every call site
// names the field with compile-time constants, so `reg` never becomes an indexed array, and the fields a wave's
// role does not touch are dropped by the compiler.
// The impedance constants of the constraint rows go to SGPRs the same way (with their reciprocals).
constexpr int N_LINK_FIELDS = O_NV / LANES;
__device__ __forceinline__ float uniform_(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ double uniform_(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// The EXTENDED-JOINT build (round 6): csrc/arm_rollout_xj.hip compiles this file a second time with MJMPC_ARM_XJ defined - the
// same kernels under their own exported names, with slide joints (joint type per link) and dry friction (one friction-loss
// row per dof, MuJoCo's mj_instantiateFriction) compiled in.  Models that need neither keep running the build without them:
// its code does not change by a single instruction.  (A translation-unit constant, not a template parameter: every helper
// sees it without its signature growing.)
#ifdef MJMPC_ARM_XJ
constexpr bool XJ = true;
#else
constexpr bool XJ = false;
#endif
template <typename T, bool REG>
struct Model {                 // view of the LDS copy of the model block
    static constexpr bool cached = REG;
    const T* m;
    int l8;
    T reg[REG ? N_LINK_FIELDS : 1];
    T inv_width, mid, inv_mid, inv_omid, dmin, ddiff, solB, solK;
    int ipower;
    __device__ __forceinline__ void cache() {
        if constexpr (REG) {
#pragma unroll
            for (int k = 0; k < N_LINK_FIELDS; ++k) reg[k] = m[k * LANES + l8];
            mid = uniform_(m[O_SOL_MID]);
            inv_width = uniform_(rcp_(m[O_SOL_WIDTH]));
            inv_mid = uniform_(rcp_(mid));
            inv_omid = uniform_(rcp_(T(1) - mid));
            dmin = uniform_(m[O_SOL_DMIN]);
            ddiff = uniform_(m[O_SOL_DMAX] - m[O_SOL_DMIN]);
            solB = uniform_(m[O_SOL_B]);
            solK = uniform_(m[O_SOL_K]);
            ipower = __builtin_amdgcn_readfirstlane((int)m[O_SOL_POWER]);
        }
        cache_xj();
    }
    __device__ __forceinline__ T link(int off, int c = 0) const {
        if constexpr (REG) return reg[off / LANES + c];
        else return m[off + c * LANES + l8];
    }
    __device__ __forceinline__ T glob(int off, int c = 0) const { return m[off + c]; }
    // XJ: my link's entries of the extended fields - cached like the others in the two-wave kernels (a lone wave pays ~110
    // cycles per LDS round trip, and a substep asks for them in four places), read from LDS otherwise
    bool xs = false;
    T xf = T(0), xD = T(0), xB = T(0);
    __device__ __forceinline__ void cache_xj() {
        if constexpr (XJ && REG) {
            xs = m[O_JTYPE + l8] != T(0);
            xf = m[O_FLOSS + l8];
            xD = m[O_FLOSS_D + l8];
            xB = uniform_(m[O_FLOSS_B]);
        }
    }
    __device__ __forceinline__ bool xj_slide() const { if constexpr (XJ && REG) return xs; else return XJ && m[O_JTYPE + l8] != T(0); }
    __device__ __forceinline__ T xj_floss() const { if constexpr (XJ && REG) return xf; else return m[O_FLOSS + l8]; }
    __device__ __forceinline__ T xj_floss_D() const { if constexpr (XJ && REG) return xD; else return m[O_FLOSS_D + l8]; }
    __device__ __forceinline__ T xj_floss_B() const { if constexpr (XJ && REG) return xB; else return m[O_FLOSS_B]; }
};

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

// ---- small vector helpers -------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void cross(const T* a, const T* b, T* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T>
__device__ __forceinline__ T dot(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T>
__device__ __forceinline__ void matvec(const T* R, const T* x, T* y) {
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
// symmetric 3x3 stored xx yy zz xy xz yz times vector
template <typename T>
__device__ __forceinline__ void symvec(const T* S, const T* x, T* y) {
    y[0] = S[0] * x[0] + S[3] * x[1] + S[4] * x[2];
    y[1] = S[3] * x[0] + S[1] * x[1] + S[5] * x[2];
    y[2] = S[4] * x[0] + S[5] * x[1] + S[2] * x[2];
}

// one Hillis-Steele step of the transform scan:  X_i <- X_{i-S} o X_i   (identity where i < S)
template <int S, typename T>
__device__ __forceinline__ void fk_scan_step(T* R, T* p, int l8) {
    T Ra[9], pa[3], Rn[9], t[3];
    for (int k = 0; k < 9; ++k) Ra[k] = (k % 4 == 0) ? shr<S>(R[k], T(1), l8) : shr0<S>(R[k], l8);
    for (int k = 0; k < 3; ++k) pa[k] = shr0<S>(p[k], l8);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = Ra[3 * i] * R[j] + Ra[3 * i + 1] * R[3 + j] + Ra[3 * i + 2] * R[6 + j];
    matvec(Ra, p, t);
    for (int k = 0; k < 3; ++k) p[k] = pa[k] + t[k];
    for (int k = 0; k < 9; ++k) R[k] = Rn[k];
}

// MuJoCo mj_makeImpedance + mj_referenceConstraint for one scalar row (r = pos - margin)
template <typename T, typename MT>
__device__ __forceinline__ void row_params(const MT& M, T r, T diag_approx, T jv, T& D, T& aref) {
    if constexpr (MT::cached) {
        T x = fabs(r) * M.inv_width, y;
        x = x > T(1) ? T(1) : x;
        if (M.ipower == 2) {
            T om = T(1) - x;
            y = x <= M.mid ? x * x * M.inv_mid : T(1) - om * om * M.inv_omid;
        } else if (M.ipower == 1) {
            y = x;
        } else {
            const bool lo = x <= M.mid;
            const T xa = lo ? x : T(1) - x, ia = lo ? M.inv_mid : M.inv_omid;
            T qq = xa;
            for (int k = 1; k < M.ipower; ++k) qq *= xa * ia;
            y = lo ? qq : T(1) - qq;
        }
        const T imp = M.dmin + y * M.ddiff;
        T Rr = (T(1) - imp) * rcp_(imp) * diag_approx;
        Rr = Rr < T(1e-15) ? T(1e-15) : Rr;
        D = rcp_(Rr);
        aref = -M.solB * jv - M.solK * imp * r;
        return;
    }
    const T dmin = M.glob(O_SOL_DMIN), dmax = M.glob(O_SOL_DMAX), width = M.glob(O_SOL_WIDTH);
    const T mid = M.glob(O_SOL_MID), power = M.glob(O_SOL_POWER);
    T x = fabs(r) * rcp_(width), y;
    x = x > T(1) ? T(1) : x;
    if (power == T(2)) {
        T om = T(1) - x;
        y = x <= mid ? x * x * rcp_(mid) : T(1) - om * om * rcp_(T(1) - mid);
    } else if (power == T(1)) {
        y = x;
    } else {
        // integer power > 2 (the model compiler rejects anything else): x^p / mid^(p-1) by repeated multiplication -
        // the library pow() would cost the whole kernel ~20 VGPRs and ~60 spilled SGPRs for a path no shipped model takes
        const int n = (int)power;
        const bool lo = x <= mid;
        const T xa = lo ? x : T(1) - x, ma = lo ? mid : T(1) - mid;
        T num = xa, den = T(1);
        for (int k = 1; k < n; ++k) {
            num *= xa;
            den *= ma;
        }
        const T qq = num * rcp_(den);
        y = lo ? qq : T(1) - qq;
    }
    T imp = dmin + y * (dmax - dmin);
    T Rr = (T(1) - imp) * rcp_(imp) * diag_approx;
    Rr = Rr < T(1e-15) ? T(1e-15) : Rr;
    D = rcp_(Rr);
    aref = -M.glob(O_SOL_B) * jv - M.glob(O_SOL_K) * imp * r;
}

// Dense LDL^T of one particle's 7x7 matrix held ENTIRELY by one lane (28 values), no cross-lane traffic.
// The lanes of a particle do not share the factorisation - a DPP broadcast inside an 8-lane group costs 3 DPP
// moves per 32 bits in the interleaved layout, which made the row-per-lane factorisation ~80 % data movement.
// Instead the matrix goes through the particle's LDS tile anyway (diagonal-major -> row-major), every solver
// lane reads all of it, and the lanes split by ROLE: links 0-3 factor the Newton matrix H = M + J'DJ, links 4-7
// the Euler matrix M + h B - the same instruction stream on different data, so both factors cost one pass.
template <typename T>
struct Dense {
    T a[21];            // strictly lower triangle, row-major: (1,0) (2,0) (2,1) (3,0) ...; l_ij after factor()
    T d[MAX_LINKS];     // diagonal; 1 / pivot after factor()
    static __device__ __forceinline__ constexpr int ix(int i, int j) { return i * (i - 1) / 2 + j; }

    __device__ __forceinline__ void load(const T* tile, const T* diag) {
#pragma unroll
        for (int i = 1; i < MAX_LINKS; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j) a[ix(i, j)] = tile[i * LANES + j];
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) d[i] = diag[i];
    }
    // += w * jc jc'
    __device__ __forceinline__ void add_rank1(T w, const T* jc) {
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) {
            const T wi = w * jc[i];
            d[i] += wi * jc[i];
#pragma unroll
            for (int j = 0; j < i; ++j) a[ix(i, j)] += wi * jc[j];
        }
    }
    __device__ __forceinline__ void factor() {
#pragma unroll
        for (int k = 0; k < MAX_LINKS; ++k) {
            const T inv = rcp_fast(d[k]);
            d[k] = inv;
#pragma unroll
            for (int i = k + 1; i < MAX_LINKS; ++i) {
                const T aik = a[ix(i, k)];              // unscaled a_ik = l_ik d_k
                const T lik = aik * inv;
#pragma unroll
                for (int j = k + 1; j < i; ++j) a[ix(i, j)] -= aik * a[ix(j, k)];   // a_jk already holds l_jk
                d[i] -= aik * lik;
                a[ix(i, k)] = lik;
            }
        }
    }
    // b <- (L D L')^-1 b
    __device__ __forceinline__ void solve(T* b) const {
#pragma unroll
        for (int i = 1; i < MAX_LINKS; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j) b[i] -= a[ix(i, j)] * b[j];
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) b[i] *= d[i];
#pragma unroll
        for (int i = MAX_LINKS - 2; i >= 0; --i)
#pragma unroll
            for (int j = i + 1; j < MAX_LINKS; ++j) b[i] -= a[ix(j, i)] * b[j];
    }
};

// Per-particle LDS block: the 8x8 mass-matrix tile followed by the vectors the lanes hand to the solver lanes
// and back.  The stride keeps 16-byte alignment and shifts consecutive particles by 4 (f32) / 8 (f64) banks.
constexpr int V_DH = LANES * LANES, V_RH = V_DH + LANES, V_DE = V_RH + LANES, V_RE = V_DE + LANES, V_XH = V_RE + LANES,
              V_XE = V_XH + LANES, V_JC = V_XE + LANES, V_TAU = V_JC + LANES, V_EI = V_TAU + LANES,
              PSTRIDE = V_EI + LANES * LANES + 4,      // V_EI: (M + h B)^-1, DUO only
              // the limit rows the DYN wave hands to the SOLVE wave at E1 travel in vectors the SOLVE wave only
              // (re)writes after it has taken them, and nobody touches between E3 and E1
              V_LS = V_XH, V_LD = V_DH, V_LA = V_RH;

// u_i += sum_k W[k][i] q_k + W[nv+k][i] v_k   (the joint part of clw^T obs), link values by DPP broadcast
template <int K, typename T>
__device__ __forceinline__ void cl_accumulate(const double* __restrict__ W, int A, int nv, int l8, T cq, T cv, double& u) {
    const double qk = (double)bcast<K>(cq), vk = (double)bcast<K>(cv);
    if (K < nv) u += W[K * A + l8] * qk + W[(nv + K) * A + l8] * vk;
    if constexpr (K + 1 < MAX_LINKS) cl_accumulate<K + 1, T>(W, A, nv, l8, cq, cv, u);
}

template <int S, typename T>
__device__ __forceinline__ void mass_diagonals(const T* sw, const T* sv, const T* Fn, const T* Ff, T* d) {
    if constexpr (S == 0) {
        d[0] = dot(sw, Fn) + dot(sv, Ff);
    } else {
        T fn[3], ff[3];
        for (int c = 0; c < 3; ++c) {
            fn[c] = shl_raw<S>(Fn[c]);
            ff[c] = shl_raw<S>(Ff[c]);
        }
        d[S] = dot(sw, fn) + dot(sv, ff);
    }
    if constexpr (S + 1 < MAX_LINKS) mass_diagonals<S + 1, T>(sw, sv, Fn, Ff, d);
}

struct ArmInts {            // wave-uniform integers (SGPRs)
    int site_link, n_sphere, sph_link, frame_skip, nv;
};

// ---- one mj_step ------------------------------------------------------------------------------
// Quantities of my link after forward kinematics, everything in world coordinates about the WORLD ORIGIN.
template <typename T>
struct LinkFrame {
    T R[9], p[3], ax[3];        // link frame (rotation, origin), joint axis in link coordinates
    T mass, a[3], cw[3];        // joint axis, centre of mass
    T Ib[6], hm[3];             // rotational inertia about the origin, first moment m c
    T sw[3], sv[3];             // motion axis S = (a, p x a)
};

// 1. forward kinematics: local transform (Rodrigues, frames are world-aligned at qpos0) + scan
template <typename T, typename MT>
__device__ __forceinline__ void kinematics(const MT& M, T sq, T cq, int l8, LinkFrame<T>& L, T q = T(0)) {
    T* R = L.R;
    T* p = L.p;
    T* ax = L.ax;
    const T s = sq, c = cq, tt = T(1) - cq;
    for (int k = 0; k < 3; ++k) { ax[k] = M.link(O_AXIS, k); p[k] = M.link(O_OFF, k); }
    if constexpr (XJ) {             // a slide joint: (sin, cos) stay (0, 1) - the rotation below is the identity - and q moves the origin
        const T qs = M.xj_slide() ? q : T(0);
        for (int k = 0; k < 3; ++k) p[k] += qs * ax[k];
    }
    R[0] = c + tt * ax[0] * ax[0];
    R[1] = tt * ax[0] * ax[1] - s * ax[2];
    R[2] = tt * ax[0] * ax[2] + s * ax[1];
    R[3] = tt * ax[0] * ax[1] + s * ax[2];
    R[4] = c + tt * ax[1] * ax[1];
    R[5] = tt * ax[1] * ax[2] - s * ax[0];
    R[6] = tt * ax[0] * ax[2] - s * ax[1];
    R[7] = tt * ax[1] * ax[2] + s * ax[0];
    R[8] = c + tt * ax[2] * ax[2];
    fk_scan_step<1>(R, p, l8);
    fk_scan_step<2>(R, p, l8);
    fk_scan_step<4>(R, p, l8);
}

// 2. world-frame quantities of my link
template <typename T, typename MT>
__device__ __forceinline__ void link_frames(const MT& M, LinkFrame<T>& L) {
    const T* R = L.R;
    const T* p = L.p;
    const T mass = L.mass = M.link(O_MASS);
    T t3[3];
    matvec(R, L.ax, L.a);                   // joint axis
    {
        const T com[3] = {M.link(O_COM, 0), M.link(O_COM, 1), M.link(O_COM, 2)};
        matvec(R, com, t3);
    }
    T* cw = L.cw;
    for (int k = 0; k < 3; ++k) cw[k] = p[k] + t3[k];
    T* Ib = L.Ib;                           // rotational inertia about the origin: R I R^T + m(|c|^2 - c c^T)
    {
        T Il[6], RI[9];
        for (int k = 0; k < 6; ++k) Il[k] = M.link(O_INERTIA, k);
        for (int i = 0; i < 3; ++i) {
            const T* r = R + 3 * i;
            RI[3 * i + 0] = r[0] * Il[0] + r[1] * Il[3] + r[2] * Il[4];
            RI[3 * i + 1] = r[0] * Il[3] + r[1] * Il[1] + r[2] * Il[5];
            RI[3 * i + 2] = r[0] * Il[4] + r[1] * Il[5] + r[2] * Il[2];
        }
        T cc = dot(cw, cw);
        Ib[0] = dot(RI + 0, R + 0) + mass * (cc - cw[0] * cw[0]);
        Ib[1] = dot(RI + 3, R + 3) + mass * (cc - cw[1] * cw[1]);
        Ib[2] = dot(RI + 6, R + 6) + mass * (cc - cw[2] * cw[2]);
        Ib[3] = dot(RI + 0, R + 3) - mass * cw[0] * cw[1];
        Ib[4] = dot(RI + 0, R + 6) - mass * cw[0] * cw[2];
        Ib[5] = dot(RI + 3, R + 6) - mass * cw[1] * cw[2];
    }
    for (int k = 0; k < 3; ++k) { L.hm[k] = mass * cw[k]; L.sw[k] = L.a[k]; }
    cross(p, L.a, L.sv);
    if constexpr (XJ) {             // a slide joint's motion axis: S = (0, a)
        if (M.xj_slide())
            for (int k = 0; k < 3; ++k) { L.sw[k] = T(0); L.sv[k] = L.a[k]; }
    }
}

// 3. joint-space bias force c(q, v) (Newton-Euler with zero joint acceleration, base acceleration -g)
template <typename T, typename MT>
__device__ __forceinline__ T bias_force(const MT& M, const LinkFrame<T>& L, T v, int l8) {
    const T *sw = L.sw, *sv = L.sv, *Ib = L.Ib, *hm = L.hm;
    const T mass = L.mass;
    // spatial velocity V_i = sum_{k<=i} S_k qd_k and velocity-product acceleration
    T Vw[3], Vv[3];
    for (int k = 0; k < 3; ++k) {
        Vw[k] = psum(sw[k] * v, l8);
        Vv[k] = psum(sv[k] * v, l8);
    }
    T Aw[3], Av[3];
    {
        T xw[3] = {sw[0] * v, sw[1] * v, sw[2] * v}, xv[3] = {sv[0] * v, sv[1] * v, sv[2] * v};
        T dw[3], d1[3], d2[3];
        cross(Vw, xw, dw);
        cross(Vw, xv, d1);
        cross(Vv, xw, d2);
        for (int k = 0; k < 3; ++k) {
            Aw[k] = psum(dw[k], l8);
            Av[k] = psum(d1[k] + d2[k], l8) - M.glob(O_GRAVITY, k);     // base acceleration -g
        }
    }
    // body force  f = I A + V x* (I V),  I(w, v) = (Ib w + h x v, m v - h x w)
    T nV[3], fV[3], nA[3], fA[3], t1[3], t2[3];
    symvec(Ib, Vw, nV);
    cross(hm, Vv, t1);
    cross(hm, Vw, t2);
    for (int k = 0; k < 3; ++k) { nV[k] += t1[k]; fV[k] = mass * Vv[k] - t2[k]; }
    symvec(Ib, Aw, nA);
    cross(hm, Av, t1);
    cross(hm, Aw, t2);
    for (int k = 0; k < 3; ++k) { nA[k] += t1[k]; fA[k] = mass * Av[k] - t2[k]; }
    T c1[3], c2[3], c3[3];
    cross(Vw, nV, c1);
    cross(Vv, fV, c2);
    cross(Vw, fV, c3);
    T fn[3], ff[3];
    for (int k = 0; k < 3; ++k) {
        fn[k] = ssum(nA[k] + c1[k] + c2[k], l8);
        ff[k] = ssum(fA[k] + c3[k], l8);
    }
    return dot(sw, fn) + dot(sv, ff);
}

// 4. composite inertia (suffix sums), the mass matrix by diagonals, and diagonal-major -> row-major through this
// particle's 8x8 LDS tile: the solves read their rows from the tile instead of holding a second copy of the
// matrix in registers.  Returns M[i][i] of my dof.
template <typename T>
__device__ __forceinline__ T mass_matrix_tile(const LinkFrame<T>& L, int l8, T* ldsM) {
    const T *sw = L.sw, *sv = L.sv;
    T d[MAX_LINKS];
    {
        T mc = ssum(L.mass, l8), hc[3], Ic[6];
        for (int k = 0; k < 3; ++k) hc[k] = ssum(L.hm[k], l8);
        for (int k = 0; k < 6; ++k) Ic[k] = ssum(L.Ib[k], l8);
        T Fn[3], Ff[3], t1[3], t2[3];
        symvec(Ic, sw, Fn);
        cross(hc, sv, t1);
        cross(hc, sw, t2);
        for (int k = 0; k < 3; ++k) { Fn[k] += t1[k]; Ff[k] = mc * sv[k] - t2[k]; }
        mass_diagonals<0, T>(sw, sv, Fn, Ff, d);
    }
#pragma unroll
    for (int sft = 0; sft < MAX_LINKS; ++sft) {
        if (l8 + sft < MAX_LINKS) {
            ldsM[l8 * LANES + l8 + sft] = d[sft];
            ldsM[(l8 + sft) * LANES + l8] = d[sft];
        }
    }
    return d[0];
}

#ifndef ARM_NO_FLAGS_CODE
// The same with the diagonals S0 <= s < S1 only (the four-wave shape splits them between two wavefronts: the suffix sums and
// the composite-inertia products are repeated by both, the shifted dot products and the tile stores are shared out).
template <int S0, int S1, typename T>
__device__ __forceinline__ T mass_matrix_tile_part(const LinkFrame<T>& L, int l8, T* ldsM) {
    const T *sw = L.sw, *sv = L.sv;
    T d[MAX_LINKS];
#pragma unroll
    for (int k = 0; k < MAX_LINKS; ++k) d[k] = T(0);
    T mc = ssum(L.mass, l8), hc[3], Ic[6];
    for (int k = 0; k < 3; ++k) hc[k] = ssum(L.hm[k], l8);
    for (int k = 0; k < 6; ++k) Ic[k] = ssum(L.Ib[k], l8);
    T Fn[3], Ff[3], t1[3], t2[3];
    symvec(Ic, sw, Fn);
    cross(hc, sv, t1);
    cross(hc, sw, t2);
    for (int k = 0; k < 3; ++k) { Fn[k] += t1[k]; Ff[k] = mc * sv[k] - t2[k]; }
    auto diag_s = [&](auto SC) {
        constexpr int S = decltype(SC)::value;
        if constexpr (S >= S0 && S < S1) {
            if constexpr (S == 0) {
                d[0] = dot(sw, Fn) + dot(sv, Ff);
            } else {
                T fn[3], ff[3];
                for (int c = 0; c < 3; ++c) { fn[c] = shl_raw<S>(Fn[c]); ff[c] = shl_raw<S>(Ff[c]); }
                d[S] = dot(sw, fn) + dot(sv, ff);
            }
        }
    };
    diag_s(std::integral_constant<int, 0>()); diag_s(std::integral_constant<int, 1>()); diag_s(std::integral_constant<int, 2>());
    diag_s(std::integral_constant<int, 3>()); diag_s(std::integral_constant<int, 4>()); diag_s(std::integral_constant<int, 5>());
    diag_s(std::integral_constant<int, 6>());
#pragma unroll
    for (int sft = S0; sft < S1; ++sft) {
        if (l8 + sft < MAX_LINKS) {
            ldsM[l8 * LANES + l8 + sft] = d[sft];
            ldsM[(l8 + sft) * LANES + l8] = d[sft];
        }
    }
    return d[0];
}

#endif
// plane-sphere contact geometry (mjc_PlaneSphere): signed distance, and - if some particle of the wave is within
// the margin - my dof's entry of the contact Jacobian row and the row velocity J v
template <typename T, typename MT>
__device__ __forceinline__ void contact_geometry(const MT& M, const ArmInts& I, const LinkFrame<T>& L,
                                                 const T* ctr, T v, int l8, T& cdist, bool& cinst, T& jc, T& jv) {
    const T pn[3] = {M.glob(O_PLANE_N, 0), M.glob(O_PLANE_N, 1), M.glob(O_PLANE_N, 2)};
    const T sph_r = M.glob(O_SPH_R);
    cdist = dot(ctr, pn) - M.glob(O_PLANE_D) - sph_r;
    cinst = cdist < M.glob(O_SPH_MARGIN);
    jc = T(0);
    jv = T(0);
    if (__any(cinst)) {
        T r[3], ar[3];
        for (int k = 0; k < 3; ++k) r[k] = ctr[k] - pn[k] * (sph_r + T(0.5) * cdist) - L.p[k];
        cross(L.a, r, ar);
        T jl = dot(pn, ar);
        if constexpr (XJ) jl = M.xj_slide() ? dot(pn, L.a) : jl;
        jc = (cinst && l8 <= I.sph_link) ? jl : T(0);
        jv = gsum(jc * v);
    }
}

// q, v: state of my dof (updated).  aw: constraint-solver warm start (previous qacc).  (sq, cq) =
// (sin q, cos q), advanced by angle addition.  rows: active-set memory.  tau_act: gear * clip(ctrl).
// site: world position of the tracked site computed from the q this substep STARTED with (MuJoCo
// runs kinematics before integrating), evaluated as if it were attached to the calling lane's link: only the
// lane of the site's link holds the real one, and the caller broadcasts it when a value is consumed (once per
// env step instead of once per substep).
//
// ROLE: below ~one wave per SIMD the launch lasts as long as ONE wavefront's serial program.  Such
// launches therefore run TWO wavefronts per particle group (a 128-thread workgroup, the waves land on different SIMDs
// of one CU) that split the substep's task graph and meet at three s_barriers through the particle's LDS block:
//   DYN   kinematics, link frames, bias forces -> tau   [E1]  factor M + h B, explicit inverse    [E2]
//         env-step records (inputs, action / cost / observations)                                 [E3] integrate
//   SOLVE kinematics, link frames, mass matrix -> tile, constraint rows   [E1]  factor H = M + J'DJ, one column of
//         H^-1 per lane, Newton iteration on the active set   [E2]   qacc = (M + h B)^-1 (tau + J'f)   [E3] integrate
// Both waves integrate the same qacc with the same instructions, so their copies of (q, v, sin q, cos q) stay
// bit-identical.  SOLO = one wave does everything (launches that fill the chip anyway).
enum Role : int { SOLO = 0, DYN = 1, SOLVE = 2,
                  // round 6: the roles of the FLAG-synchronised shapes (two or four wavefronts per particle group) - see flag_front
                  QDYN = 3, QSOLVE = 4, QMASS = 5, QAUX = 6 };
// which of the particle block's four spare slots holds this role's reset flags (every wave of a group keeps its own copy)
template <int ROLE>
__device__ __forceinline__ constexpr int role_slot() {
    return (ROLE == SOLVE || ROLE == QSOLVE) ? 1 : (ROLE == QMASS ? 2 : (ROLE == QAUX ? 3 : 0));
}
// the role that counts a particle's resets (one wave of the group)
template <int ROLE>
__device__ __forceinline__ constexpr bool role_counts() { return ROLE == SOLO || ROLE == DYN || ROLE == QDYN; }

// Phase timing of one wavefront (developer builds: -DMJMPC_STAMPS, read back with tools/stamps.py).  Shader-clock
// deltas between consecutive marks are summed per phase; workgroup 0 adds its sums to diag[2 + 16 * wave + phase]
// (64-bit slots) when the rollout ends.  In product builds every member is empty.
struct Stamps {
#if defined(MJMPC_STAMPS) && defined(MJMPC_STAMPS_ABS)
    // -DMJMPC_STAMPS_ABS=n: a TIMELINE instead of sums - the shader clock at every mark of substep n alone (one wave's marks
    // against another's: who posted what when; tools/stamps.py --timeline)
    long long acc[16];
    int nsub;
    __device__ __forceinline__ void begin() { for (int k = 0; k < 16; ++k) acc[k] = 0; nsub = 0; }
    __device__ __forceinline__ void mark(int k) {
        if (nsub == MJMPC_STAMPS_ABS) acc[k] = clock64();
        if (k == 13 || k == 15) ++nsub;          // (13: integration done)
    }
    __device__ __forceinline__ void flush(unsigned* diag, int wave, int lane) {
        if (diag && blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 16; ++k) ((unsigned long long*)diag)[2 + 16 * wave + k] = (unsigned long long)acc[k];
    }
#elif defined(MJMPC_STAMPS)
    long long last, acc[16];
    __device__ __forceinline__ void begin() { for (int k = 0; k < 16; ++k) acc[k] = 0; last = clock64(); }
    // -DMJMPC_STAMPS_MASK=0x..: only the marks whose bit is set read the clock (e.g. 0x1c = the two sides of E1 alone: the
    // full set of marks costs ~500 cycles per substep, more than the waits it is meant to measure)
#ifdef MJMPC_STAMPS_MASK
    __device__ __forceinline__ void mark(int k) {
        if (!((MJMPC_STAMPS_MASK >> k) & 1)) return;
        const long long t = clock64(); acc[k] += t - last; last = t;
    }
#else
    // (MJMPC_STAMPS_FENCE: a scheduling barrier at every mark - exact attribution at the price of a slightly different schedule)
#ifdef MJMPC_STAMPS_FENCE
    __device__ __forceinline__ void mark(int k) {
        __builtin_amdgcn_sched_barrier(0);
        const long long t = clock64(); acc[k] += t - last; last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    __device__ __forceinline__ void mark(int k) { const long long t = clock64(); acc[k] += t - last; last = t; }
#endif
#endif
    __device__ __forceinline__ void flush(unsigned* diag, int wave, int lane) {
        if (diag && blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 16; ++k) ((unsigned long long*)diag)[2 + 16 * wave + k] = (unsigned long long)acc[k];
    }
#else
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(unsigned*, int, int) {}
#endif
};

// cross-wave rendezvous of a DUO workgroup: LDS traffic of this wave has landed (lgkmcnt), global stores are NOT
// waited for (__syncthreads() would drain vmcnt too and stall on the record stores of the env step)
__device__ __forceinline__ void duo_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// XJ: my dof's friction-loss row (MuJoCo mj_instantiateFriction: J = e_l at position 0, Huber cost): D = 1 / R from the
// impedance at 0 (a constant of the model, O_FLOSS_D), reference acceleration -b v.  Zone of the row at acceleration xa:
// -1: slope <= -f (force +f), 0: quadratic, +1: slope >= f (force -f); f32: a slope within rounding of a bound keeps its zone
template <typename T>
struct FlossRow {
    T f = T(0), D = T(0), aref = T(0);
    int z = 0;
    __device__ __forceinline__ int zone_at(T xa) const {
        if (!(f > T(0))) return 0;
        const T sl = D * (xa - aref);
        const T bc = sizeof(T) == 4 ? T(2e-5) * (fabs(sl) + f) : T(0);
        if (z < 0) return sl > -f + bc ? (sl >= f ? 1 : 0) : -1;
        if (z > 0) return sl < f - bc ? (sl <= -f ? -1 : 0) : 1;
        return sl <= -f - bc ? -1 : (sl >= f + bc ? 1 : 0);
    }
    __device__ __forceinline__ T diag() const { return (f > T(0) && z == 0) ? D : T(0); }                   // its part of H
    __device__ __forceinline__ T rhs() const { return f > T(0) ? (z == 0 ? D * aref : (z < 0 ? f : -f)) : T(0); }
    __device__ __forceinline__ T force(T xa) const { return f > T(0) ? (z == 0 ? -D * (xa - aref) : (z < 0 ? f : -f)) : T(0); }
    __device__ __forceinline__ int memory() const { return f > T(0) ? (16 | ((z + 1) << 5)) : 0; }          // bits of `rows`
};
template <typename T, typename MT>
__device__ __forceinline__ FlossRow<T> floss_row(const MT& M, T v, int rows) {
    FlossRow<T> r;
    if constexpr (XJ) {
        r.f = M.xj_floss();
        if (r.f > T(0)) {
            r.D = M.xj_floss_D();
            r.aref = -M.xj_floss_B() * v;
            r.z = (rows & 16) ? ((rows >> 5) & 3) - 1 : 0;      // the zone the previous substep ended in
        }
    }
    return r;
}

// XJ: EXACT LINE SEARCH of the constraint problem's cost along d = a_new - a_prev (MuJoCo's Newton solver does the same).
// Without friction-loss rows the active-set iteration may take full steps - with the active set fixed the cost is quadratic and
// the iteration settles in 1 - 3 solves - but a Huber row has no curvature in its linear zones, and full steps between zone
// assignments can cycle (seen: 140 cap hits in 2e5 cart-pole substeps).  With the step length that minimises the TRUE cost
// phi(alpha) = 1/2 (a - a0)' M (a - a0) + sum_rows s_i(J_i a - aref_i),  a = a_prev + alpha d,  the iteration descends and ends.
// phi' is piecewise linear and increasing: safeguarded Newton on alpha (each evaluation: the lanes' row terms, two sums over
// the particle's lanes).  Every lane of the particle returns ITS entry of a_prev + alpha* d.  V_XH / V_XE carry a_prev and d.
template <typename T, typename MT>
__device__ __forceinline__ T xj_line_search(const MT& M, T* ldsM, int l8, T a_prev, T a_new, T tau, T sig, T D, T aref,
                                            bool inst, T jc, T Dc, T arefc, bool cinst, const FlossRow<T>& fl) {
    const T d = a_new - a_prev;
    ldsM[V_XH + l8] = a_prev;
    ldsM[V_XE + l8] = d;
    LDS_WAVE_SYNC();
    T Ma = M.link(O_ARMATURE) * a_prev, Md = M.link(O_ARMATURE) * d;       // rows of M x (the tile holds M without the armature)
#pragma unroll
    for (int j = 0; j < MAX_LINKS; ++j) {
        const T m = ldsM[l8 * LANES + j];
        Ma += m * ldsM[V_XH + j];
        Md += m * ldsM[V_XE + j];
    }
    const T g0 = gsum(d * (Ma - tau)), h0 = gsum(d * Md);                   // the smooth part: phi' = g0 + alpha h0 + rows
    // my rows along the line: residual r(alpha) = r0 + alpha rd
    const T rl0 = sig * a_prev - aref, rld = sig * d;                       // limit row (cost 1/2 D min(r, 0)^2)
    const T rf0 = a_prev - fl.aref, rfd = d;                                // friction-loss row (Huber: slope clamp(D r, -f, f))
    T rc0 = T(0), rcd = T(0);
    const bool any_c = __any(cinst);
    if (any_c) { rc0 = gsum(jc * a_prev) - arefc; rcd = gsum(jc * d); }     // the contact row (one per particle: counted by lane 0's share)
    T alpha = T(1);
    T lo = T(0), hi = T(0);                                                 // bracket once phi' has been seen positive
    bool have_hi = false;
#pragma unroll 1
    for (int k = 0; k < 12; ++k) {
        T g = T(0), hh = T(0);
        if (inst) { const T r = rl0 + alpha * rld; if (r < T(0)) { g += D * r * rld; hh += D * rld * rld; } }
        if (fl.f > T(0)) {
            const T sl = fl.D * (rf0 + alpha * rfd);
            if (sl <= -fl.f) g -= fl.f * rfd;
            else if (sl >= fl.f) g += fl.f * rfd;
            else { g += sl * rfd; hh += fl.D * rfd * rfd; }
        }
        T gp = gsum(g), hp = gsum(hh);
        if (cinst) { const T r = rc0 + alpha * rcd; if (r < T(0)) { gp += Dc * r * rcd; hp += Dc * rcd * rcd; } }
        const T dphi = g0 + alpha * h0 + gp, ddphi = h0 + hp;               // (uniform over the particle's lanes)
        const bool pos = dphi > T(0);
        if (pos) { hi = alpha; have_hi = true; } else lo = alpha;
        const T tol = T(sizeof(T) == 4 ? 1e-6 : 1e-14) * (fabs(g0) + fabs(h0) + T(1e-30));
        const bool done = fabs(dphi) <= tol || !(ddphi > T(0));
        T next = alpha - dphi * rcp_(ddphi > T(0) ? ddphi : T(1));
        if (have_hi && !(next > lo && next < hi)) next = T(0.5) * (lo + hi); // (safeguard: stay inside the bracket)
        if (!have_hi && !(next > lo)) next = T(2) * alpha + T(1);
        if (!done) alpha = next;
        if (!__any(!done)) break;
    }
    return a_prev + alpha * d;
}

// Developer build -DARM_PER_PARTICLE (round 6, measured: profiles/r06_arm_per_particle_ab.txt): the solver's decisions per
// PARTICLE (8-lane group) instead of per wavefront, as the tree kernels take them since this round - a particle none of whose
// rows changes is frozen while its seven wave-mates iterate on, the rank-one correction is taken by the particles with exactly
// one flipped limit row, the long sine / cosine series by the lanes whose step needs it.
template <typename T>
struct ParticleFreeze {
    bool done = false, act = false, cact = false;
    T aw = T(0);
    int z = 0;
};
__device__ __forceinline__ bool gany(bool x) { return gsum(x ? 1.0f : 0.0f) > 0.5f; }

// active-set test of the constraint rows at acceleration aw (f32: a row whose residual is within rounding of zero
// keeps its state - such rows otherwise flip back and forth until the iteration cap, seen a few times per 5e8
// solves; f64 has never failed to settle and keeps the plain sign test)
template <typename T>
__device__ __forceinline__ void active_set(T aw, T sig, T aref, T jc, T arefc, bool inst, bool cinst, bool act,
                                           bool cact, bool& act2, bool& cact2) {
    if constexpr (sizeof(T) == 4) {
        const T res = sig * aw - aref, resc = gsum(jc * aw) - arefc;
        const T band = T(4e-6) * (fabs(aref) + fabs(aw) + T(1));
        const T bandc = T(4e-6) * (fabs(arefc) + fabs(resc + arefc) + T(1));
        act2 = inst && (act ? !(res > band) : (res < -band));
        cact2 = cinst && (cact ? !(resc > bandc) : (resc < -bandc));
    } else {
        act2 = inst && (sig * aw - aref < T(0));
        cact2 = cinst && (gsum(jc * aw) - arefc < T(0));
    }
}

// developer A/B switch (round 4): the two waves of a DUO group trade their chains - see arm_front
#ifdef ARM_SWAP_ROLES
constexpr bool ARM_SWAP = true;
#else
constexpr bool ARM_SWAP = false;
#endif
// developer A/B switch: the SOLVE wave takes E2 (the Euler inverse's hand-over) after its Newton iterations instead of
// inside the first one - measured (round 4, 4096 x 32 f64): rollout kernel 0.1775 -> 0.186 ms, the inverse's row then
// arrives with its LDS latency in front of the Euler product; off
#if defined(ARM_E2_LATE) || defined(ARM_SWAP_ROLES)
constexpr bool E2_LATE = true;
#else
constexpr bool E2_LATE = false;
#endif
// DUO launches: which wave evaluates the joint-limit rows (see arm_front)
template <typename T>
__device__ __forceinline__ constexpr bool rows_by_dyn() {
#ifdef ARM_ROWS_IN_SOLVE
    return false;
#else
    return sizeof(T) == 8;
#endif
}

// joint-limit row of my dof (MuJoCo mj_instantiateLimit, strict dist < margin(=0)): sig = +-1 (0: no row), and - when
// some lane of the wavefront has a row - its regulariser D = 1 / R and reference acceleration
template <typename T, typename MT>
__device__ __forceinline__ void limit_row(const MT& M, T q, T v, T& sig, T& D, T& aref) {
    T dist = T(0);
    sig = T(0);
    if (M.link(O_LIMITED) != T(0)) {
        const T dlo = q - M.link(O_RANGE_LO), dhi = M.link(O_RANGE_HI) - q;
        if (dlo < T(0)) { sig = T(1); dist = dlo; }
        else if (dhi < T(0)) { sig = T(-1); dist = dhi; }
    }
    D = T(0);
    aref = T(0);
    if (__any(sig != T(0))) {
        row_params(M, dist, M.link(O_DOF_INVW), sig * v, D, aref);
        D = sig != T(0) ? D : T(0);
        aref = sig != T(0) ? aref : T(0);
    }
}

// MuJoCo's reset on instability (mj_checkPos / mj_checkVel / mj_checkAcc -> mj_resetData [EXT]).
// ResetCtl::rst = the reset record of my model block (RolloutFusion::reset_rec), nullptr: no resets.  ONE test per substep,
// where it ends (arm_back: the acceleration, the integrated qpos and qvel): a NaN or an entry beyond mjMAXVAL = 1e10 in
//   the acceleration -> mj_checkAcc: the particle goes on from the record (the state one substep after the reset state), bit 0;
//   qpos / qvel      -> the NEXT mj_step's mj_checkPos / mj_checkVel: noted (bit 1) and applied where that substep begins
//                       (arm_front) - until then the state, e.g. in the next observation, is what MuJoCo shows too.
// Either way the particle's controls are zero until its env step ends (bit 0; mj_resetData zeroes data.ctrl, which
// do_simulation wrote once before its frame_skip calls of sim.step()).  `any` is wave-uniform (a scalar register): a wave
// none of whose particles ever reset pays the test in arm_back and two scalar branches per substep, nothing else.  Every
// role of a particle group takes the same decisions from the same (q, v, qacc), so the two waves of a group stay in step.
// Flags of rflags(): 1 zero controls | 2 reset pending | 4 the latest arm_back reset on the acceleration | RST_EVER this particle
// has reset at some point of the rollout | and two OPTIONS of the launch, parked in the same LDS word because a register of their
// own costs the fused iteration's kernel its schedule (round 6: 182.2 against 177.8 us at 4096 x 32 f64 with two runtime
// flags held beside the hot loops): RST_REAL this launch steps the REAL env (state_out / the fused iteration's env step) - its
// resets are counted a second time, in diag[2], where the reference's worker raises MujocoException; RST_INF
// RolloutFusion::inf_on_reset.  They are written once, when a launch that has them begins, and read on the rare path only.
constexpr int RST_EVER = 8, RST_REAL = 16, RST_INF = 32, RST_OPTS = RST_REAL | RST_INF;
struct ResetCtl {
    bool any = false;           // some particle of this wavefront has reset, or has a reset pending
    bool count = false;         // per lane: a live particle (its resets are counted)
};
// per particle and wavefront (the two waves of a DUO group keep their own copy: they pass the same points at their own pace):
// 1 zero controls | 2 reset pending | 4 the latest arm_back reset on the acceleration - in a spare slot of the particle's LDS
// block, read and written on the rare path only (a register of its own cost the fused iteration's kernel, the one with the
// least room, copies in its hot loops)
template <int ROLE, typename T>
__device__ __forceinline__ int rflags(const T* ldsM) { return (int)ldsM[PSTRIDE - 4 + role_slot<ROLE>()]; }
template <int ROLE, typename T>
__device__ __forceinline__ void rflags_set(T* ldsM, int f) { ldsM[PSTRIDE - 4 + role_slot<ROLE>()] = (T)f; }
// where the record lives is asked for only on the rare path (a callable): held in registers through the substep it cost
// the fused iteration's kernel - the one with the least room - scalar spills in its hot loops (measured: +4 % per launch)
struct NoResetRecord {
    __device__ __forceinline__ const double* operator()() const { return nullptr; }
};
__device__ __forceinline__ unsigned long long particle_lanes(int lane) { return 0x5555ull << ((lane & 0x30) | (lane & 1)); }
template <typename T>
__device__ __forceinline__ bool mj_is_bad(T x) { return !(fabs(x) <= T(1e10)); }       // mju_isBad
// the start state of a rollout (the first mj_step's mj_checkPos / mj_checkVel)
template <int ROLE, typename T>
__device__ __forceinline__ void reset_check_start(ResetCtl& rc, T q, T v, int lane, T* ldsM) {
    const unsigned long long bal = __ballot(mj_is_bad(q) || mj_is_bad(v));
    if (bal != 0ull) {
        rc.any = true;
        if (bal & particle_lanes(lane)) rflags_set<ROLE>(ldsM, rflags<ROLE>(ldsM) | 2);
    }
}

template <int ROLE, typename T, typename MT>
__device__ __forceinline__ void arm_front(const MT& M, const ArmInts& I, T& q, T& v, T& aw, T& sq, T& cq,
                                          int& rows, T tau_act, T* ldsM, int lane, int l8, T* site,
                                          unsigned* diag, bool& free_step, Stamps& ST, ResetCtl* rc = nullptr) {
    free_step = false;
#ifdef MJMPC_NO_RESET         // developer A/B: the reset emulation compiled out (tools/ab_build.py)
    rc = nullptr;
#endif
    if (rc && __builtin_expect(rc->any, 0)) {
        int f = rflags<ROLE>(ldsM) & ~4;
        if (f & 2) {                        // mj_checkPos / mj_checkVel of this mj_step: mj_resetData, and on from there
            q = T(0); v = T(0); sq = T(0); cq = T(1); aw = T(0);
            rows = 0;
            f = (f & RST_OPTS) | 1 | RST_EVER;
            if (role_counts<ROLE>() && diag && l8 == 0 && rc->count) { atomicAdd(diag + 1, 1u); if (f & RST_REAL) atomicAdd(diag + 2, 1u); }
        }
        rflags_set<ROLE>(ldsM, f);
        if (f & 1) tau_act = M.link(O_GEAR) * fmin(fmax(T(0), M.link(O_CTRL_LO)), M.link(O_CTRL_HI));
    }
    PHASE();
    LinkFrame<T> L;
    kinematics(M, sq, cq, l8, L, q);
    if constexpr (ROLE != SOLVE) {
        const T sp[3] = {M.glob(O_SITE_POS, 0), M.glob(O_SITE_POS, 1), M.glob(O_SITE_POS, 2)};
        T t[3];
        matvec(L.R, sp, t);
        for (int k = 0; k < 3; ++k) site[k] = L.p[k] + t[k];    // as if the site sat on MY link; the caller picks the lane
    }
    // sphere centre for the contact row (needs R of the sphere's link, so it is taken here)
    T ctr[3] = {T(0), T(0), T(0)};
    bool near_plane = false;
    if (ROLE != DYN && I.n_sphere > 0) {
        const T sp[3] = {M.glob(O_SPH_POS, 0), M.glob(O_SPH_POS, 1), M.glob(O_SPH_POS, 2)};
        T t[3];
        matvec(L.R, sp, t);
        for (int k = 0; k < 3; ++k) t[k] += L.p[k];      // the centre as if the sphere sat on MY link
        if constexpr (ROLE == SOLVE) {
            // only the sphere link's lane holds the real centre; its distance test travels to the other lanes of
            // the particle as one bit of a wave ballot, and the centre itself (6 ds_bpermute, ~21 cycles each for a
            // lone wave) only when some particle of the wave is within the margin
            const T pn[3] = {M.glob(O_PLANE_N, 0), M.glob(O_PLANE_N, 1), M.glob(O_PLANE_N, 2)};
            const bool mine = dot(t, pn) - M.glob(O_PLANE_D) - M.glob(O_SPH_R) < M.glob(O_SPH_MARGIN);
            const unsigned long long bal = __ballot(mine);
            near_plane = (bal >> lane_of_link(lane, I.sph_link)) & 1ull;
            if (__any(near_plane))
                for (int k = 0; k < 3; ++k) ctr[k] = __shfl(t[k], lane_of_link(lane, I.sph_link));
        } else {
            for (int k = 0; k < 3; ++k) ctr[k] = __shfl(t[k], lane_of_link(lane, I.sph_link));
        }
    }
    ST.mark(0);         // kinematics
    PHASE();
    link_frames(M, L);
    ST.mark(1);         // world-frame link quantities

    if constexpr (ROLE == DYN && ARM_SWAP) {
        // (ARM_SWAP) this wave - the one that also reads the inputs, draws the samples and writes the records - takes the
        // LIGHT chain: mass matrix, limit rows, then the Euler inverse; the other wave the heavy one: bias forces, constraint
        // solve, Euler product.  The motor torque travels to it through LDS (read after E1).
        ldsM[V_TAU + l8] = tau_act;
        T dg = mass_matrix_tile(L, l8, ldsM);
        dg += M.link(O_ARMATURE);
        ldsM[V_JC + l8] = dg;                           // (the Newton matrix's diagonal starts from it; the slot is the other wave's own after E1)
        ldsM[V_DE + l8] = dg + M.glob(O_TIMESTEP) * M.link(O_DAMPING);
        ST.mark(4);
        duo_barrier();                                  // E1: tile, Euler diagonal, limit rows, motor torque out
        ST.mark(3);
        Dense<T> F;
        F.load(ldsM, ldsM + V_DE);
        F.factor();
        T col[MAX_LINKS];
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) col[i] = (i == l8) ? T(1) : T(0);
        F.solve(col);
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) ldsM[V_EI + l8 * LANES + i] = col[i];
        ST.mark(6);
        return;                                         // (E2 is taken in arm_back, behind this wave's records and sampler)
    }
    if constexpr (ROLE == DYN) {
        // 5. smooth force: -bias + passive damping + motor, handed to the SOLVE wave
        const T bias = bias_force(M, L, v, l8);
        ldsM[V_TAU + l8] = -bias - M.link(O_DAMPING) * v + tau_act;
        // the joint-limit rows too (a function of q and v alone): this wave reaches E1 ~700 cycles before the SOLVE
        // wave, which used to spend ~800 on them right after it
        // (f64 only: 181 -> 176.5 us per 4096 x 32 launch; in f32, whose scans are single instructions, the DYN wave has
        // no such slack and the move costs 3 %.  ARM_ROWS_IN_SOLVE: developer A/B switch back to the round-2 split)
        if constexpr (rows_by_dyn<T>()) {
            T sg, Dl, al;
            limit_row(M, q, v, sg, Dl, al);
            ldsM[V_LS + l8] = sg;
            ldsM[V_LD + l8] = Dl;
            ldsM[V_LA + l8] = al;
        }
        ST.mark(2);     // velocities, bias forces, limit rows
        duo_barrier();                                  // E1: tau out; mass-matrix tile and Euler diagonal in
        ST.mark(3);
        // 8'. explicit inverse of the Euler matrix M + h B while the SOLVE wave works on the constraints: every lane
        // holds the whole factorisation (struct Dense) and substitutes ITS unit vector - the same instruction stream
        // as one solve yields all columns; the SOLVE wave then finishes the substep with a matrix-vector product
        Dense<T> F;
        F.load(ldsM, ldsM + V_DE);
        F.factor();
        T col[MAX_LINKS];
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) col[i] = (i == l8) ? T(1) : T(0);
        F.solve(col);
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) ldsM[V_EI + l8 * LANES + i] = col[i];
        ST.mark(6);     // Euler matrix: factorisation, inverse
        duo_barrier();                                  // E2: inverse out
        ST.mark(7);
        return;
    }

    T bias = T(0);
    if constexpr (ROLE == SOLO || (ROLE == SOLVE && ARM_SWAP)) bias = bias_force(M, L, v, l8);
    T dgM = T(0);
    const T damping = M.link(O_DAMPING), h = M.glob(O_TIMESTEP);
    if constexpr (!(ROLE == SOLVE && ARM_SWAP)) {
        dgM = mass_matrix_tile(L, l8, ldsM);
        dgM += M.link(O_ARMATURE);
        ldsM[V_DE + l8] = dgM + h * damping;
    }
    if constexpr (ROLE == SOLO) LDS_WAVE_SYNC();
    ST.mark(4);         // (bias forces,) composite inertia, mass matrix, tile

    // 5. smooth force: -bias + passive damping + motor (SOLVE: arrives from the DYN wave at E1)
    T tau = -bias - damping * v + tau_act;
    T sw_sg = T(0), sw_D = T(0), sw_a = T(0);           // (ARM_SWAP: my limit row, evaluated in front of E1 - this wave gets there first)
    if constexpr (ROLE == SOLVE && ARM_SWAP) limit_row(M, q, v, sw_sg, sw_D, sw_a);
    if constexpr (ROLE == SOLVE) {
        duo_barrier();                                  // E1: tau in; tile and Euler diagonal out
        ST.mark(3);
        if constexpr (ARM_SWAP) {
            tau += ldsM[V_TAU + l8];                    // (the motor torque; bias and damping are this wave's own)
            dgM = ldsM[V_JC + l8];
        } else {
            tau = ldsM[V_TAU + l8];
        }
    }
    T ei[MAX_LINKS];            // SOLVE: my row of (M + h B)^-1 (read once the DYN wave has published it)

    // 6. constraint rows.  Limits (SOLVE: evaluated by the DYN wave before E1)
    T sig = T(0), dist = T(0), D = T(0), aref = T(0);
    bool inst = false;
    constexpr bool rows_from_dyn = ROLE == SOLVE && (rows_by_dyn<T>() || ARM_SWAP);
    if constexpr (ROLE == SOLVE && ARM_SWAP) {
        sig = sw_sg;
        D = sw_D;
        aref = sw_a;
        inst = sig != T(0);
    } else if constexpr (rows_from_dyn) {
        sig = ldsM[V_LS + l8];
        D = ldsM[V_LD + l8];
        aref = ldsM[V_LA + l8];
        inst = sig != T(0);
    } else if (M.link(O_LIMITED) != T(0)) {
        T dlo = q - M.link(O_RANGE_LO), dhi = M.link(O_RANGE_HI) - q;
        if (dlo < T(0)) { sig = T(1); dist = dlo; inst = true; }
        else if (dhi < T(0)) { sig = T(-1); dist = dhi; inst = true; }
    }
    // plane-sphere contact (condim 1): mjc_PlaneSphere + mj_instantiateContact
    bool cinst = false;
    T jc = T(0), Dc = T(0), arefc = T(0);
    if (I.n_sphere > 0 && (ROLE != SOLVE || __any(near_plane))) {
        T cdist, jv;
        contact_geometry(M, I, L, ctr, v, l8, cdist, cinst, jc, jv);
        if constexpr (ROLE == SOLVE) {      // ctr is only valid where the ballot said so (same test, same operands)
            cinst = cinst && near_plane;
            jc = near_plane ? jc : T(0);
        }
        if (__any(cinst)) {
            row_params(M, cdist - M.glob(O_SPH_MARGIN), M.glob(O_SPH_INVW), jv, Dc, arefc);
            Dc = cinst ? Dc : T(0);
            arefc = cinst ? arefc : T(0);
        }
    }
    FlossRow<T> fl = floss_row(M, v, rows);             // (XJ: my dof's friction-loss row; otherwise empty)
    const bool any_rows = __any(inst || cinst || (XJ && fl.f > T(0)));
    if (!any_rows) rows = 0;
    if constexpr (!rows_from_dyn) {
        if (any_rows) {
            row_params(M, dist, M.link(O_DOF_INVW), sig * v, D, aref);
            D = inst ? D : T(0);
            aref = inst ? aref : T(0);
        }
    }
    ST.mark(5);         // constraint rows
    // initial active set: a row that existed in the previous substep keeps its state, a new row is assumed
    // active (it appears because the joint moves into its limit); `rows` = inst | act<<1 | cinst<<2 | cact<<3
    // of the previous substep
    bool act = inst && ((rows & 1) ? (rows & 2) != 0 : true);
    bool cact = cinst && ((rows & 4) ? (rows & 8) != 0 : true);
    bool changed = false;
    const bool any_c = __any(cinst);
    T qfrc_c = T(0);

    // 7. primal active-set Newton on  1/2 a'Ma - tau'a + sum_active 1/2 D (J a - aref)^2, and
    // 8. mj_Euler with implicit joint damping:  (M + h B) qacc = qfrc_smooth + qfrc_constraint.
    if constexpr (ROLE == SOLVE) {
        // DUO: every lane factors the Newton matrix H = M + J'DJ and substitutes its own unit vector, i.e. holds one
        // column (= row) of H^-1; the acceleration is then a dot product with the right-hand side, and the Euler
        // solve a dot product with the row of (M + h B)^-1 the DYN wave has prepared meanwhile.
        if (any_rows) {
            if (any_c) ldsM[V_JC + l8] = jc;
            changed = true;
#ifdef ARM_PER_PARTICLE
            ParticleFreeze<T> fz;
#endif
            for (int it = 0; it < newton_maxit<T>(); ++it) {
                T rhs = tau + (act ? D * sig * aref : T(0));
                if (any_c) rhs += cact ? Dc * jc * arefc : T(0);
                T dh = dgM + (act ? D : T(0));
                if constexpr (XJ) { rhs += fl.rhs(); dh += fl.diag(); }
                ldsM[V_DH + l8] = dh;
                ldsM[V_RH + l8] = rhs;
                LDS_WAVE_SYNC();
                Dense<T> F;
                F.load(ldsM, ldsM + V_DH);
                T rh[MAX_LINKS];            // the right-hand side, fetched with the matrix: its LDS latency hides behind the
#pragma unroll                              // factorisation instead of standing behind the barrier
                for (int i = 0; i < MAX_LINKS; ++i) rh[i] = ldsM[V_RH + i];
                if (any_c && __any(cact)) {
                    T jv[MAX_LINKS];
#pragma unroll
                    for (int i = 0; i < MAX_LINKS; ++i) jv[i] = ldsM[V_JC + i];
                    F.add_rank1(cact ? Dc : T(0), jv);
                }
                F.factor();
                T col[MAX_LINKS];
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) col[i] = (i == l8) ? T(1) : T(0);
                F.solve(col);
                ST.mark(it == 0 ? 6 : 9);               // factorisation + inverse / further iterations
                if (it == 0 && !(rows_by_dyn<T>() || ARM_SWAP) && !E2_LATE) {     // (E2 where round 2 had it: this wave arrives ~2000 cycles after E1)
                    duo_barrier();
#pragma unroll
                    for (int i = 0; i < MAX_LINKS; ++i) ei[i] = ldsM[V_EI + l8 * LANES + i];
                }
                T acc = T(0);
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) acc += col[i] * rh[i];
                if constexpr (XJ) {         // from the second solve on: an exact line search from the previous iterate
                    if (it > 0) acc = xj_line_search(M, ldsM, l8, aw, acc, tau, sig, D, aref, inst, jc, Dc, arefc, cinst, fl);
                }
                aw = acc;
                bool act2, cact2;
                active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
                bool flip = act2 != act, cflip = cact2 != cact;
                changed = flip || cflip;
                act = act2;
                cact = cact2;
                bool fflip = false;         // XJ: my friction-loss row changed zone (the next iteration refactors)
                if constexpr (XJ) {
                    const int z2 = fl.zone_at(aw);
                    fflip = z2 != fl.z;
                    fl.z = z2;
                    changed = changed || fflip;
                }
#ifdef ARM_PER_PARTICLE
                if (fz.done) { aw = fz.aw; act = fz.act; cact = fz.cact; if constexpr (XJ) fl.z = fz.z; changed = flip = cflip = fflip = false; }
#endif
                // E2: the DYN wave needs ~1000 cycles after E1 for (M + h B)^-1; with the limit rows arriving from it
                // this wave gets HERE in about as many (after the first factorisation it would still wait ~400).  Taking
                // the rendezvous inside the first iteration rather than at the end of the substep leaves only E3
                // between the last Newton iteration and the integration.
                if (it == 0 && (rows_by_dyn<T>() || ARM_SWAP) && !E2_LATE) {
                    duo_barrier();
#pragma unroll
                    for (int i = 0; i < MAX_LINKS; ++i) ei[i] = ldsM[V_EI + l8 * LANES + i];   // my row of (M + h B)^-1, early
                }
                ST.mark(it == 0 ? 8 : 9);               // acceleration + active-set check / further iterations
                if (!__any(changed)) break;
                // One limit row j of a particle changed state (the usual case): H' = H + c e_j e_j', c = +-D_j, and the
                // right-hand side moves by d e_j, d = c sig_j aref_j.  With z = H^-1 e_j (lane i holds z_i = its
                // column's entry j) Sherman-Morrison gives  a' = y - c z y_j / (1 + c z_j),  y = a + d z  - a few
                // dozen instructions instead of a second factorisation.  Several flips in one particle, or a flip of
                // the contact row, take the general path (next iteration refactors).
                const float nflip = gsum(flip ? 1.0f : 0.0f);
#ifdef ARM_PER_PARTICLE
                if (!fz.done && !gany(changed)) { fz.done = true; fz.aw = aw; fz.act = act; fz.cact = cact; fz.z = fl.z; }
                const bool single = !XJ && !fz.done && nflip > 0.5f && nflip < 1.5f && !gany(cflip);
                if (__any(single)) {
                    if (single && flip) {
                        T zjj = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) zjj = (l8 == i) ? col[i] : zjj;
                        const T c = act ? D : -D;
                        ldsM[V_XH + 0] = (T)l8;
                        ldsM[V_XH + 1] = c;
                        ldsM[V_XH + 2] = c * sig * aref;
                        ldsM[V_XH + 3] = zjj;
                        ldsM[V_XH + 4] = aw;
                    }
                    LDS_WAVE_SYNC();
                    if (single) {
                        const int j = (int)ldsM[V_XH + 0];
                        const T c = ldsM[V_XH + 1], dl = ldsM[V_XH + 2], zjj = ldsM[V_XH + 3], aj = ldsM[V_XH + 4];
                        T z = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) z = (j == i) ? col[i] : z;
                        const T yj = aj + dl * zjj;
                        aw = (aw + dl * z) - c * z * yj * rcp_(T(1) + c * zjj);
                    }
                    active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
                    if (single) {
                        changed = (act2 != act) || (cact2 != cact);
                        act = act2;
                        cact = cact2;
                    }
                    if (!fz.done && !gany(changed)) { fz.done = true; fz.aw = aw; fz.act = act; fz.cact = cact; fz.z = fl.z; }
                    ST.mark(9);
                    if (!__any(changed)) break;
                }
#else
                if (!XJ && !__any(cflip || nflip > 1.5f || fflip)) {   // (XJ: every change refactors and goes through the line search)
                    if (flip) {
                        T zjj = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) zjj = (l8 == i) ? col[i] : zjj;
                        const T c = act ? D : -D;
                        ldsM[V_XH + 0] = (T)l8;
                        ldsM[V_XH + 1] = c;
                        ldsM[V_XH + 2] = c * sig * aref;
                        ldsM[V_XH + 3] = zjj;
                        ldsM[V_XH + 4] = aw;
                    }
                    LDS_WAVE_SYNC();
                    if (nflip > 0.5f) {
                        const int j = (int)ldsM[V_XH + 0];
                        const T c = ldsM[V_XH + 1], dl = ldsM[V_XH + 2], zjj = ldsM[V_XH + 3], aj = ldsM[V_XH + 4];
                        T z = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) z = (j == i) ? col[i] : z;
                        const T yj = aj + dl * zjj;
                        aw = (aw + dl * z) - c * z * yj * rcp_(T(1) + c * zjj);
                    }
                    active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
                    changed = (act2 != act) || (cact2 != cact);
                    act = act2;
                    cact = cact2;
                    if constexpr (XJ) {
                        const int z2 = fl.zone_at(aw);
                        changed = changed || z2 != fl.z;
                        fl.z = z2;
                    }
                    ST.mark(9);
                    if (!__any(changed)) break;
                }
#endif
                LDS_WAVE_SYNC();                        // V_RH is rewritten
            }
            if (changed && diag) atomicAdd(diag, 1u);
            rows = (inst ? 1 : 0) | (act ? 2 : 0) | (cinst ? 4 : 0) | (cact ? 8 : 0) | fl.memory();
            // qfrc_constraint = J^T f,  f = -D (J a - aref) on active rows
            qfrc_c = act ? -D * (sig * aw - aref) * sig : T(0);
            if constexpr (XJ) qfrc_c += fl.force(aw);
            if (__any(cact)) {
                T fcn = cact ? -Dc * (gsum(jc * aw) - arefc) : T(0);
                qfrc_c += jc * fcn;
            }
        }
        ldsM[V_RE + l8] = tau + qfrc_c;
        if (!any_rows || E2_LATE) {
            duo_barrier();                              // E2 (substeps without rows; otherwise taken inside the loop)
#pragma unroll
            for (int i = 0; i < MAX_LINKS; ++i) ei[i] = ldsM[V_EI + l8 * LANES + i];
        } else {
            LDS_WAVE_SYNC();
        }
        ST.mark(7);
        T x = T(0);
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) x += ei[i] * ldsM[V_RE + i];
        ldsM[V_XE + l8] = x;
        free_step = !any_rows;
        ST.mark(10);    // constraint force, Euler product
        return;
    }

    //    SOLO solver lanes (struct Dense): links 0-3 of a particle hold the Newton factor, links 4-7 the Euler
    //    factor; right-hand sides and solutions travel through the particle's LDS vectors.
    const bool roleH = l8 < 4;
    Dense<T> F;
    if (any_rows) {
        changed = true;
        if (any_c) ldsM[V_JC + l8] = jc;
#ifdef ARM_PER_PARTICLE
        ParticleFreeze<T> fz;
#endif
        for (int it = 0; it < newton_maxit<T>(); ++it) {
            T rhs = tau + (act ? D * sig * aref : T(0));
            if (any_c) rhs += cact ? Dc * jc * arefc : T(0);
            T dh = dgM + (act ? D : T(0));
            if constexpr (XJ) { rhs += fl.rhs(); dh += fl.diag(); }
            ldsM[V_DH + l8] = dh;
            ldsM[V_RH + l8] = rhs;
            LDS_WAVE_SYNC();
            if (it == 0 || roleH) {
                F.load(ldsM, ldsM + (roleH ? V_DH : V_DE));
                if (any_c && __any(cact)) {
                    T jv[MAX_LINKS];
#pragma unroll
                    for (int i = 0; i < MAX_LINKS; ++i) jv[i] = ldsM[V_JC + i];
                    F.add_rank1((roleH && cact) ? Dc : T(0), jv);
                }
                F.factor();
            }
            ST.mark(it == 0 ? 6 : 9);                   // first factorisation / further iterations
            if (roleH) {
                T b[MAX_LINKS];
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) b[i] = ldsM[V_RH + i];
                F.solve(b);
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) ldsM[V_XH + i] = b[i];
            }
            LDS_WAVE_SYNC();
            if constexpr (XJ) {
                const T a_new = ldsM[V_XH + l8];
                LDS_WAVE_SYNC();            // (the line search reuses V_XH)
                aw = it > 0 ? xj_line_search(M, ldsM, l8, aw, a_new, tau, sig, D, aref, inst, jc, Dc, arefc, cinst, fl) : a_new;
            } else {
                aw = ldsM[V_XH + l8];
            }
            bool act2, cact2;
            active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
            changed = (act2 != act) || (cact2 != cact);
            act = act2;
            cact = cact2;
            if constexpr (XJ) {
                const int z2 = fl.zone_at(aw);
                changed = changed || z2 != fl.z;
                fl.z = z2;
            }
#ifdef ARM_PER_PARTICLE
            if (fz.done) { aw = fz.aw; act = fz.act; cact = fz.cact; if constexpr (XJ) fl.z = fz.z; changed = false; }
            else if (!gany(changed)) { fz.done = true; fz.aw = aw; fz.act = act; fz.cact = cact; fz.z = fl.z; }
#endif
            ST.mark(it == 0 ? 8 : 9);                   // first solve + active-set check / further iterations
            if (!__any(changed)) break;
        }
        if (changed && diag) atomicAdd(diag, 1u);
        rows = (inst ? 1 : 0) | (act ? 2 : 0) | (cinst ? 4 : 0) | (cact ? 8 : 0) | fl.memory();
        // qfrc_constraint = J^T f,  f = -D (J a - aref) on active rows
        qfrc_c = act ? -D * (sig * aw - aref) * sig : T(0);
        if constexpr (XJ) qfrc_c += fl.force(aw);
        if (__any(cact)) {
            T fcn = cact ? -Dc * (gsum(jc * aw) - arefc) : T(0);
            qfrc_c += jc * fcn;
        }
        ldsM[V_RE + l8] = tau + qfrc_c;
        LDS_WAVE_SYNC();
    } else {
        ldsM[V_RE + l8] = tau;
        LDS_WAVE_SYNC();
        F.load(ldsM, ldsM + V_DE);
        F.factor();
    }
    if (!roleH) {
        T b[MAX_LINKS];
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) b[i] = ldsM[V_RE + i];
        F.solve(b);
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) ldsM[V_XE + i] = b[i];
    }
    free_step = !any_rows;
    ST.mark(10);        // constraint force, Euler solve
}

// ---- FLAG-synchronised shapes (round 6): two or four wavefronts per particle group, no s_barrier ---------------------------
// What round 5's phase clocks said about DUO (profiles/r04_arm_phase_clocks.txt): 13 % of the critical wave's cycles are waits
// at the three barriers - a barrier makes a wave wait for the slowest wave even when what it needs next is long there.  And
// what round 6's first attempts said (profiles/r06_arm_shapes.txt): cutting the critical chain
//   integrate -> frames -> mass matrix -> factor H -> Newton -> Euler product
// itself across wavefronts buys nothing - every cut costs a hand-over (~150 cycles) and repeats the frames.
// So: the chain stays on ONE wavefront (QSOLVE), which never waits at a barrier; everything else moves to the other waves,
// which hand their results over through LDS behind FLAGS - a flag carries the substep's sequence number, the DS unit executes a
// wave's LDS instructions in order, so whoever sees the flag sees the data written before it:
//   NW = 2   QSOLVE  integrate; frames, mass matrix -> tile, Euler diagonal [TILE]; contact row; takes [ROWS]; factors H, one
//                    column of H^-1 per lane; takes [TAU] only now; Newton iteration on the active set; takes [EI];
//                    qacc = (M + h B)^-1 (tau + J'f)  [X]
//            QDYN    takes [X], integrates; limit rows (a function of q, v alone) [ROWS]; frames, bias forces -> tau [TAU];
//                    takes [TILE], factors M + h B, one column of its inverse per lane [EI]; env-step records, sampler
//   NW = 4   (launches of at most a quarter wave per SIMD, P <= 2048 on 256 CUs: the helpers have SIMDs of their own)
//            QDYN    as above without the limit rows and the inverse
//            QAUX    takes [X], integrates (q, v only: it needs no frames); limit rows [ROWS]; takes [TILE] [TILE2], the
//                    inverse of M + h B [EI]
//            QMASS   takes [X], integrates; frames, composite inertia, the FAR diagonals of the mass matrix [TILE2]
//                    (QSOLVE computes the near ones)
// Every wave integrates the same qacc with the same explicitly rounded operations (bit-identical copies of the state, as in
// DUO).  Nothing is reused before its readers are done: every buffer of substep n is read before X_n is posted, and written
// again only by waves that have taken X_n.
// (an explicit LDS pointer: a generic `volatile int*` compiles to flat loads / stores with a vmcnt(0) drain each; and relaxed
// workgroup-scope atomics instead of `volatile`: a volatile load is followed by an s_waitcnt of its own, which would put the
// flag's round trip IN FRONT of the data reads instead of beside them)
typedef __attribute__((address_space(3))) int* qflag_ptr;
__device__ __forceinline__ int q_read(qflag_ptr qf, int which) {
    return __hip_atomic_load(qf + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void q_write(qflag_ptr qf, int which, int x) {
    __hip_atomic_store(qf + which, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
enum QFlag : int { QF_TAU = 0, QF_TILE = 1, QF_TILE2 = 2, QF_EI = 3, QF_X = 4, QF_ROWS = 5, QF_CROW = 6, QF_STUCK = 7, QF_COUNT = 8 };
__device__ __forceinline__ void q_post(qflag_ptr qf, int which, int seq, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) q_write(qf, which, seq);
    asm volatile("" ::: "memory");
}
// (the wait is bounded: a producer that never posts - a bug, or a wave of the group that could not be scheduled - ends
// the wait after 2^21 polls (~0.1 s) with the sticky flag QF_STUCK, which the launch reports through the solver-failure counter
// instead of hanging the GPU)
__device__ __forceinline__ void q_wait(qflag_ptr qf, int which, int seq) {
#ifdef ARM_QWAIT_NOP            // developer timing experiment: no waiting at all (the results are void)
    return;
#endif
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(q_read(qf, which)) < seq) {
#ifdef ARM_QUAD_SLEEP           // developer A/B: sleep between polls (s_sleep n = 64 n + 1..64 clocks)
        __builtin_amdgcn_s_sleep(ARM_QUAD_SLEEP);
#endif
        if (++spins > (1 << 21)) { q_write(qf, QF_STUCK, 1); break; }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Take a hand-over: the flag(s) and, IN THE SAME BATCH of LDS instructions, the data behind them (`loads`, a callable) - the
// flag reads are issued first, the DS unit executes in order, so a flag that reads as posted vouches for the data read after
// it; a flag that does not sends the batch round again.  One LDS round trip (~110 cycles for a lone wave) where a poll
// followed by the reads costs two - and a substep has half a dozen hand-overs on its critical chain.
template <typename F>
__device__ __forceinline__ void q_take(qflag_ptr qf, int which, int seq, F&& loads, int which2 = -1) {
    // (the first attempt is straight-line code of its own: inside a loop the compiler puts an s_waitcnt lgkmcnt(0) between the
    // flag read and the data reads - the write-after-write hazard against the previous iteration's loads into the same
    // registers - which is exactly the second round trip this is here to avoid)
    int tag = q_read(qf, which);
    int tag2 = which2 >= 0 ? q_read(qf, which2) : seq;
    asm volatile("" ::: "memory");          // (the data loads stay behind the flag loads)
    loads();
    asm volatile("" ::: "memory");
#ifndef ARM_QWAIT_NOP
    if (__builtin_expect(!(__builtin_amdgcn_readfirstlane(tag) >= seq && __builtin_amdgcn_readfirstlane(tag2) >= seq), 0)) {
        int spins = 0;
        for (;;) {
            tag = q_read(qf, which);
            tag2 = which2 >= 0 ? q_read(qf, which2) : seq;
            asm volatile("" ::: "memory");
            loads();
            asm volatile("" ::: "memory");
            if (__builtin_amdgcn_readfirstlane(tag) >= seq && __builtin_amdgcn_readfirstlane(tag2) >= seq) break;
            if (++spins > (1 << 21)) { q_write(qf, QF_STUCK, 1); break; }
        }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}

#ifndef ARM_NO_FLAGS_CODE
// NW = 4: the diagonals 0 ... ARM_NEAR_DIAGS - 1 of the mass matrix stay with the solving wave, the others go to QMASS (which
// starts a hand-over later but has nothing else to do).  Measured at 1024 x 32 f64: 1: 176.0, 2: 174.4, 3: 171.1, 4: 172.7 us
#ifndef ARM_NEAR_DIAGS
#define ARM_NEAR_DIAGS 3
#endif
template <int ROLE, int NW, typename T, typename MT>
__device__ __forceinline__ void flag_front(const MT& M, const ArmInts& I, T& q, T& v, T& aw, T& sq, T& cq, int& rows,
                                           T tau_act, T* ldsM, qflag_ptr qf, int seq, int lane, int l8, T* site,
                                           unsigned* diag, bool& free_step, Stamps& ST, ResetCtl* rc = nullptr) {
    constexpr bool ROWS_MINE = ROLE == QAUX || (ROLE == QDYN && NW == 2);       // who evaluates the limit rows and inverts M + h B
    constexpr bool SPLIT = NW == 4;                                             // the mass matrix's diagonals on two waves
#ifdef ARM_CONTACT_IN_SOLVE     // developer A/B
    constexpr bool CONTACT_BY_AUX = false;
#else
    constexpr bool CONTACT_BY_AUX = NW == 4;        // the contact row comes from QAUX (which waits ~2900 cycles per substep for the tile)
#endif
    free_step = false;
#ifdef MJMPC_NO_RESET
    rc = nullptr;
#endif
    if (rc && __builtin_expect(rc->any, 0)) {
        int f = rflags<ROLE>(ldsM) & ~4;
        if (f & 2) {                        // mj_checkPos / mj_checkVel of this mj_step: mj_resetData, and on from there
            q = T(0); v = T(0); sq = T(0); cq = T(1); aw = T(0);
            rows = 0;
            f = (f & RST_OPTS) | 1 | RST_EVER;
            if (role_counts<ROLE>() && diag && l8 == 0 && rc->count) { atomicAdd(diag + 1, 1u); if (f & RST_REAL) atomicAdd(diag + 2, 1u); }
        }
        rflags_set<ROLE>(ldsM, f);
        if (f & 1) tau_act = M.link(O_GEAR) * fmin(fmax(T(0), M.link(O_CTRL_LO)), M.link(O_CTRL_HI));
    }
    PHASE();
    const T damping = M.link(O_DAMPING), h = M.glob(O_TIMESTEP);
    auto limit_rows = [&]() {
        // the joint-limit rows: a function of (q, v) alone
        T sg, Dl, al;
        limit_row(M, q, v, sg, Dl, al);
        ldsM[V_LS + l8] = sg;
        ldsM[V_LD + l8] = Dl;
        ldsM[V_LA + l8] = al;
        q_post(qf, QF_ROWS, seq, lane);
        ST.mark(5);
    };
    if constexpr (ROLE == QAUX) limit_rows();
    LinkFrame<T> L;
    T ctr[3] = {T(0), T(0), T(0)};
    bool near_plane = false;
    if constexpr (ROLE != QAUX || CONTACT_BY_AUX) {
        kinematics(M, sq, cq, l8, L, q);
        if constexpr (ROLE == QDYN) {
            const T sp[3] = {M.glob(O_SITE_POS, 0), M.glob(O_SITE_POS, 1), M.glob(O_SITE_POS, 2)};
            T t[3];
            matvec(L.R, sp, t);
            for (int k = 0; k < 3; ++k) site[k] = L.p[k] + t[k];
        }
        if (ROLE == QAUX && CONTACT_BY_AUX && I.n_sphere > 0) {
            const T sp[3] = {M.glob(O_SPH_POS, 0), M.glob(O_SPH_POS, 1), M.glob(O_SPH_POS, 2)};
            T t[3];
            matvec(L.R, sp, t);
            for (int k = 0; k < 3; ++k) ctr[k] = __shfl(t[k] + L.p[k], lane_of_link(lane, I.sph_link));
        }
        if (ROLE == QSOLVE && !CONTACT_BY_AUX && I.n_sphere > 0) {
            const T sp[3] = {M.glob(O_SPH_POS, 0), M.glob(O_SPH_POS, 1), M.glob(O_SPH_POS, 2)};
            T t[3];
            matvec(L.R, sp, t);
            for (int k = 0; k < 3; ++k) t[k] += L.p[k];
            const T pn[3] = {M.glob(O_PLANE_N, 0), M.glob(O_PLANE_N, 1), M.glob(O_PLANE_N, 2)};
            const bool mine = dot(t, pn) - M.glob(O_PLANE_D) - M.glob(O_SPH_R) < M.glob(O_SPH_MARGIN);
            const unsigned long long bal = __ballot(mine);
            near_plane = (bal >> lane_of_link(lane, I.sph_link)) & 1ull;
            if (__any(near_plane))
                for (int k = 0; k < 3; ++k) ctr[k] = __shfl(t[k], lane_of_link(lane, I.sph_link));
        }
        ST.mark(0);         // kinematics
        PHASE();
        link_frames(M, L);  // (inlined: what a role does not use is dropped)
        ST.mark(1);
    }
    if constexpr (ROLE == QDYN) {
        // (NW = 2: the solving wave asks for the rows when its mass matrix is done and for tau a factorisation later - the rows
        // go out first, but behind the frames: the start of the substep belongs to what the bias forces need)
        if constexpr (ROWS_MINE) limit_rows();
        const T bias = bias_force(M, L, v, l8);
        ldsM[V_TAU + l8] = -bias - damping * v + tau_act;
        q_post(qf, QF_TAU, seq, lane);
        ST.mark(2);
    }
    if constexpr (ROLE == QAUX && CONTACT_BY_AUX) {
        // the contact row for the solving wave (plane / sphere: distance, my dof's Jacobian entry, impedance, reference): jc ->
        // V_JC, {instantiated, D, aref} -> the head of V_RE (the solving wave writes V_RE only at the end of its substep, after
        // it has taken this)
        bool cinst = false;
        T jc = T(0), Dc = T(0), arefc = T(0);
        if (I.n_sphere > 0) {
            T cdist, jv;
            contact_geometry(M, I, L, ctr, v, l8, cdist, cinst, jc, jv);
            if (__any(cinst)) {
                row_params(M, cdist - M.glob(O_SPH_MARGIN), M.glob(O_SPH_INVW), jv, Dc, arefc);
                Dc = cinst ? Dc : T(0);
                arefc = cinst ? arefc : T(0);
            }
        }
        ldsM[V_JC + l8] = jc;
        if (l8 == 0) { ldsM[V_RE + 0] = cinst ? T(1) : T(0); ldsM[V_RE + 1] = Dc; ldsM[V_RE + 2] = arefc; }
        q_post(qf, QF_CROW, seq, lane);
        ST.mark(1);
    }
    if constexpr (ROLE == QMASS) {
        mass_matrix_tile_part<ARM_NEAR_DIAGS, MAX_LINKS>(L, l8, ldsM);
        q_post(qf, QF_TILE2, seq, lane);
        ST.mark(4);
        return;
    }
    if constexpr (ROWS_MINE) {
        Dense<T> F;
        q_take(qf, QF_TILE, seq, [&]() { F.load(ldsM, ldsM + V_DE); }, SPLIT ? QF_TILE2 : -1);
        ST.mark(3);
        F.factor();
        T col[MAX_LINKS];
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) col[i] = (i == l8) ? T(1) : T(0);
        F.solve(col);
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) ldsM[V_EI + l8 * LANES + i] = col[i];
        q_post(qf, QF_EI, seq, lane);
        ST.mark(6);
    }
    if constexpr (ROLE == QSOLVE) {
        T dgM = SPLIT ? mass_matrix_tile_part<0, ARM_NEAR_DIAGS>(L, l8, ldsM) : mass_matrix_tile(L, l8, ldsM);
        dgM += M.link(O_ARMATURE);
        ldsM[V_DE + l8] = dgM + h * damping;
        q_post(qf, QF_TILE, seq, lane);
        ST.mark(4);         // composite inertia, mass matrix, tile
        // the contact row
        bool cinst = false;
        T jc = T(0), Dc = T(0), arefc = T(0);
        if (!CONTACT_BY_AUX && I.n_sphere > 0 && __any(near_plane)) {
            T cdist, jv;
            contact_geometry(M, I, L, ctr, v, l8, cdist, cinst, jc, jv);
            cinst = cinst && near_plane;
            jc = near_plane ? jc : T(0);
            if (__any(cinst)) {
                row_params(M, cdist - M.glob(O_SPH_MARGIN), M.glob(O_SPH_INVW), jv, Dc, arefc);
                Dc = cinst ? Dc : T(0);
                arefc = cinst ? arefc : T(0);
            }
        }
        T sig, D, aref;
        if constexpr (CONTACT_BY_AUX) {
            T cin;
            q_take(qf, QF_CROW, seq, [&]() {        // (the limit rows were posted before the contact row: one flag vouches for both)
                sig = ldsM[V_LS + l8]; D = ldsM[V_LD + l8]; aref = ldsM[V_LA + l8];
                jc = ldsM[V_JC + l8]; cin = ldsM[V_RE + 0]; Dc = ldsM[V_RE + 1]; arefc = ldsM[V_RE + 2];
            }, QF_TILE2);
            cinst = cin != T(0);
        } else {
            q_take(qf, QF_ROWS, seq, [&]() { sig = ldsM[V_LS + l8]; D = ldsM[V_LD + l8]; aref = ldsM[V_LA + l8]; },
                   SPLIT ? QF_TILE2 : -1);
        }
        const bool inst = sig != T(0);
        const bool any_rows = __any(inst || cinst);
        if (!any_rows) rows = 0;
        bool act = inst && ((rows & 1) ? (rows & 2) != 0 : true);
        bool cact = cinst && ((rows & 4) ? (rows & 8) != 0 : true);
        bool changed = false;
        const bool any_c = __any(cinst);
        T qfrc_c = T(0);
        ST.mark(5);         // constraint rows
        T tau = T(0);
        T ei[MAX_LINKS];
        if (any_rows) {
            if (!CONTACT_BY_AUX && any_c) ldsM[V_JC + l8] = jc;
            changed = true;
#ifdef ARM_PER_PARTICLE
            ParticleFreeze<T> fz;
#endif
            for (int it = 0; it < newton_maxit<T>(); ++it) {
                // the rows' parts of the right-hand side; tau joins them when it has arrived:  rhs = (tau + limit) + contact
                ldsM[V_DH + l8] = dgM + (act ? D : T(0));
                ldsM[V_RH + l8] = act ? D * sig * aref : T(0);
                ldsM[V_XH + l8] = (any_c && cact) ? Dc * jc * arefc : T(0);
                LDS_WAVE_SYNC();
                Dense<T> F;
                F.load(ldsM, ldsM + V_DH);
                // (every LDS read of a phase in ONE batch, no branch between them: a lone wave pays the full ~110 cycles for
                // each dependent round trip - seven conditional reads in a row were 700 cycles of this chain)
                T rh[MAX_LINKS], tv[MAX_LINKS], rr[MAX_LINKS], xh[MAX_LINKS];
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) { rr[i] = ldsM[V_RH + i]; xh[i] = ldsM[V_XH + i]; }
                if (it > 0) {
#pragma unroll
                    for (int i = 0; i < MAX_LINKS; ++i) tv[i] = ldsM[V_TAU + i];
                }
                if (any_c && __any(cact)) {
                    T jv[MAX_LINKS];
#pragma unroll
                    for (int i = 0; i < MAX_LINKS; ++i) jv[i] = ldsM[V_JC + i];
                    F.add_rank1(cact ? Dc : T(0), jv);
                }
                F.factor();
                T col[MAX_LINKS];
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) col[i] = (i == l8) ? T(1) : T(0);
                F.solve(col);
                ST.mark(it == 0 ? 6 : 9);
                if (it == 0) {
                    q_take(qf, QF_TAU, seq, [&]() {         // the smooth force: needed only now
                        tau = ldsM[V_TAU + l8];
#pragma unroll
                        for (int i = 0; i < MAX_LINKS; ++i) tv[i] = ldsM[V_TAU + i];
                    });
                    ST.mark(7);
                }
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) rh[i] = (tv[i] + rr[i]) + xh[i];       // rhs = (tau + limit row) + contact row
                T acc = T(0);
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) acc += col[i] * rh[i];
                aw = acc;
                bool act2, cact2;
                active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
                bool flip = act2 != act, cflip = cact2 != cact;
                changed = flip || cflip;
                act = act2;
                cact = cact2;
#ifdef ARM_PER_PARTICLE
                if (fz.done) { aw = fz.aw; act = fz.act; cact = fz.cact; changed = flip = cflip = false; }
#endif
                if (it == 0) {
                    ST.mark(8);
                    q_take(qf, QF_EI, seq, [&]() {          // my row of (M + h B)^-1
#pragma unroll
                        for (int i = 0; i < MAX_LINKS; ++i) ei[i] = ldsM[V_EI + l8 * LANES + i];
                    });
                    ST.mark(14);    // substeps with rows: the wait for (M + h B)^-1
                }
                ST.mark(it == 0 ? 8 : 9);
                if (!__any(changed)) break;
                // one limit row of a particle changed state: Sherman-Morrison instead of a second factorisation (arm_front)
                const float nflip = gsum(flip ? 1.0f : 0.0f);
#ifdef ARM_PER_PARTICLE
                if (!fz.done && !gany(changed)) { fz.done = true; fz.aw = aw; fz.act = act; fz.cact = cact; }
                const bool single = !fz.done && nflip > 0.5f && nflip < 1.5f && !gany(cflip);
                if (__any(single)) {
                    LDS_WAVE_SYNC();                    // (V_XH: the contact parts have been read)
                    if (single && flip) {
                        T zjj = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) zjj = (l8 == i) ? col[i] : zjj;
                        const T c = act ? D : -D;
                        ldsM[V_XH + 0] = (T)l8;
                        ldsM[V_XH + 1] = c;
                        ldsM[V_XH + 2] = c * sig * aref;
                        ldsM[V_XH + 3] = zjj;
                        ldsM[V_XH + 4] = aw;
                    }
                    LDS_WAVE_SYNC();
                    if (single) {
                        const int j = (int)ldsM[V_XH + 0];
                        const T c = ldsM[V_XH + 1], dl = ldsM[V_XH + 2], zjj = ldsM[V_XH + 3], aj = ldsM[V_XH + 4];
                        T z = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) z = (j == i) ? col[i] : z;
                        const T yj = aj + dl * zjj;
                        aw = (aw + dl * z) - c * z * yj * rcp_(T(1) + c * zjj);
                    }
                    active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
                    if (single) {
                        changed = (act2 != act) || (cact2 != cact);
                        act = act2;
                        cact = cact2;
                    }
                    if (!fz.done && !gany(changed)) { fz.done = true; fz.aw = aw; fz.act = act; fz.cact = cact; }
                    ST.mark(9);
                    if (!__any(changed)) break;
                }
#else
                if (!__any(cflip || nflip > 1.5f)) {
                    LDS_WAVE_SYNC();                    // (V_XH: the contact parts have been read)
                    if (flip) {
                        T zjj = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) zjj = (l8 == i) ? col[i] : zjj;
                        const T c = act ? D : -D;
                        ldsM[V_XH + 0] = (T)l8;
                        ldsM[V_XH + 1] = c;
                        ldsM[V_XH + 2] = c * sig * aref;
                        ldsM[V_XH + 3] = zjj;
                        ldsM[V_XH + 4] = aw;
                    }
                    LDS_WAVE_SYNC();
                    if (nflip > 0.5f) {
                        const int j = (int)ldsM[V_XH + 0];
                        const T c = ldsM[V_XH + 1], dl = ldsM[V_XH + 2], zjj = ldsM[V_XH + 3], aj = ldsM[V_XH + 4];
                        T z = col[0];
#pragma unroll
                        for (int i = 1; i < MAX_LINKS; ++i) z = (j == i) ? col[i] : z;
                        const T yj = aj + dl * zjj;
                        aw = (aw + dl * z) - c * z * yj * rcp_(T(1) + c * zjj);
                    }
                    active_set(aw, sig, aref, jc, arefc, inst, cinst, act, cact, act2, cact2);
                    changed = (act2 != act) || (cact2 != cact);
                    act = act2;
                    cact = cact2;
                    ST.mark(9);
                    if (!__any(changed)) break;
                }
#endif
                LDS_WAVE_SYNC();                        // V_RH / V_XH are rewritten
            }
            if (changed && diag) atomicAdd(diag, 1u);
            rows = (inst ? 1 : 0) | (act ? 2 : 0) | (cinst ? 4 : 0) | (cact ? 8 : 0);
            qfrc_c = act ? -D * (sig * aw - aref) * sig : T(0);
            if (__any(cact)) {
                T fcn = cact ? -Dc * (gsum(jc * aw) - arefc) : T(0);
                qfrc_c += jc * fcn;
            }
            ldsM[V_RE + l8] = tau + qfrc_c;
            LDS_WAVE_SYNC();
        }
        T re[MAX_LINKS];
        if (any_rows) {
#pragma unroll
            for (int i = 0; i < MAX_LINKS; ++i) re[i] = ldsM[V_RE + i];
        } else {
            // no rows: qacc = (M + h B)^-1 tau straight from the two hand-overs, one batch
            q_take(qf, QF_TAU, seq, [&]() {
#pragma unroll
                for (int i = 0; i < MAX_LINKS; ++i) { re[i] = ldsM[V_TAU + i]; ei[i] = ldsM[V_EI + l8 * LANES + i]; }
            }, QF_EI);
            ST.mark(15);    // substeps without rows: the wait for tau and (M + h B)^-1
        }
        T x = T(0);
#pragma unroll
        for (int i = 0; i < MAX_LINKS; ++i) x += ei[i] * re[i];
        ldsM[V_XE + l8] = x;
        q_post(qf, QF_X, seq, lane);
        free_step = !any_rows;
        ST.mark(10);
    }
}

#endif
// second half of a substep: take delivery of qacc and integrate (every role runs the same instructions)
template <int ROLE, typename T, typename MT, typename RST = NoResetRecord>
__device__ __forceinline__ void arm_back(const MT& M, T& q, T& v, T& aw, T& sq, T& cq, T* ldsM, int l8,
                                         bool free_step, Stamps& ST, int lane = 0, ResetCtl* rc = nullptr,
                                         unsigned* diag = nullptr, RST record = RST(), qflag_ptr qf = nullptr,
                                         int seq = 0) {
    ST.mark(11);        // DYN: env-step records
    if constexpr (ROLE == DYN && ARM_SWAP) { duo_barrier(); ST.mark(7); }      // E2: inverse out (the other wave takes it after its Newton iterations)
#ifdef ARM_AB_BACK
    if constexpr (ROLE == SOLO) LDS_WAVE_SYNC();
    else duo_barrier();
    ST.mark(12);
    const T h = M.glob(O_TIMESTEP);
    {
        T x = ldsM[V_XE + l8];
#else
    T x;
    if constexpr (ROLE == QDYN || ROLE == QMASS || ROLE == QAUX) {
        q_take(qf, QF_X, seq, [&]() { x = ldsM[V_XE + l8]; });          // flag shapes: qacc is out
    } else {
        if constexpr (ROLE == SOLO || ROLE == QSOLVE) LDS_WAVE_SYNC();
        else duo_barrier();                             // E3
        x = ldsM[V_XE + l8];
    }
    ST.mark(12);        // wait at B
    const T h = M.glob(O_TIMESTEP);
    {
#endif
        if (free_step) aw = x;
        // explicitly rounded products and sums: the two waves of a DUO group must compute bit-identical (q, v), so the
        // compiler may not contract these differently in the two instantiations
        v = add_rn(v, mul_rn(h, x));
        const T dq = mul_rn(h, v);
        q = add_rn(q, dq);
        const bool slide = M.xj_slide();                // (XJ: a slide joint advances q alone - its (sin, cos) stay (0, 1))
        const T dqa = slide ? T(0) : dq;
        // advance (sin q, cos q) by dq: angle addition with a short series, one Newton step of renormalisation.
        // Large steps (|dq| > 0.25 rad per substep, never seen with h = 0.01): series at dq / 256, doubled back up.
        // (ONE wave-level test guards everything rare about the integration - this, and MuJoCo's reset on instability below: a
        // NaN or an entry beyond mjMAXVAL = 1e10 in the acceleration or the integrated state makes |dq| = h |v| huge or NaN)
        T sd, cd;
        const bool big = __any(!(fabs(dq) <= (slide ? T(1e5) : T(0.25))));
        if (__builtin_expect(big, 0)) {
            sincos_small(dqa * T(1.0 / 256.0), sd, cd);
            for (int k = 0; k < 8; ++k) {
                const T s2 = T(2) * sd * cd;
                cd = T(1) - T(2) * sd * sd;
                sd = s2;
            }
#ifdef ARM_PER_PARTICLE
            if (fabs(dq) <= (slide ? T(1e5) : T(0.25))) sincos_small(dqa, sd, cd);     // (the long series in the lanes that need it)
#endif
        } else {
            sincos_small(dqa, sd, cd);
        }
        const T s1 = sq * cd + cq * sd, c1 = cq * cd - sq * sd;
        const T k = T(1.5) - T(0.5) * (s1 * s1 + c1 * c1);
        sq = s1 * k;
        cq = c1 * k;
#ifdef MJMPC_NO_RESET
        rc = nullptr;
#endif
        if (rc && __builtin_expect(big, 0)) {
            // the substep's test (ResetCtl).  The acceleration every role holds is the Euler solve's (M + h B)^-1
            // (qfrc_smooth + qfrc_constraint), which stands in for mj_forward's qacc
            const unsigned long long bal = __ballot(mj_is_bad(x) || mj_is_bad(q) || mj_is_bad(v));
            if (bal != 0ull) {
                rc->any = true;
                const unsigned long long mine = particle_lanes(lane);
                const double* rst = record();
                if (rst == nullptr) {
                    // (no record: the launch that makes it)
                } else if (__ballot(mj_is_bad(x)) & mine) {
                    // mj_checkAcc: mj_resetData, mj_forward again, mj_Euler from the reset state = the record
                    q = (T)rst[l8];
                    v = (T)rst[LANES + l8];
                    sq = (T)rst[2 * LANES + 3 + l8];
                    cq = (T)rst[3 * LANES + 3 + l8];
                    aw = T(0);
                    const int f0 = rflags<ROLE>(ldsM);
                    rflags_set<ROLE>(ldsM, f0 | 1 | 4 | RST_EVER);
                    if (role_counts<ROLE>() && diag && l8 == 0 && rc->count) { atomicAdd(diag + 1, 1u); if (f0 & RST_REAL) atomicAdd(diag + 2, 1u); }
                } else if (bal & mine) {
                    rflags_set<ROLE>(ldsM, rflags<ROLE>(ldsM) | 2);
                }
            }
        }
    }
    LDS_WAVE_SYNC();            // the tile is rewritten by the next substep
    ST.mark(13);        // integration
}

// ---- the rest of the control iteration (MONO launches) -----------------------------------------------
// Dynamic LDS of a MONO rollout workgroup, in doubles: MONO_RED scratch (cost-to-go of its 8 particles, their weights) |
// the action tile T[8][H A].
constexpr int MONO_RED = 32;

// One env step of the real arm (Reacher7DOFEnv.step, reacher_env.py:29-39) by this workgroup, from (q, v, tgt) - my
// lane's entries of the f64 state vector `st` (qpos[8] | qvel[8] | target[3]), read by the caller - with the action in
// `action` (LDS): the same substeps as a particle of the rollout - all eight particle slots carry the one state, slot 0
// writes it back.  DUO: wave 0 = DYN, wave 1 = SOLVE as in the rollout.  The particle blocks of `lds` must be zero.
template <typename T, bool DUO, typename MT>
__device__ __forceinline__ void real_env_step(const MT& M, const ArmInts& I, const MonoStep& mo, double* st, T q, T v,
                                              const T* tgt, const double* action, int A, T* lds, int lane, int wave, int l8,
                                              int g, unsigned* diag) {
    const int nv = I.nv;
    T* ldsM = lds + g * PSTRIDE;
    T aw = T(0);
    if (l8 >= nv) { q = T(0); v = T(0); }
    T sinq, cosq;
    sincos_(q, sinq, cosq);
    if constexpr (XJ) { if (M.xj_slide()) { sinq = T(0); cosq = T(1); } }
    int rows = 0;
    Stamps ST;
    ST.begin();
    bool fs = false;
    ResetCtl rc;
    rc.count = g == 0;              // (all eight particle slots carry the one state: they reset together; slot 0 counts)
    // (the real env: its resets are counted apart - the option bit goes into this wave's flags word, blocks zeroed by the caller)
    if (DUO && wave == 1) rflags_set<SOLVE>(ldsM, RST_REAL);
    else rflags_set<SOLO>(ldsM, RST_REAL);
    auto record = [&mo]() -> const double* { return mo.reset_rec; };
    if (mo.reset_rec) {
        if (DUO && wave == 1) reset_check_start<SOLVE>(rc, q, v, lane, ldsM);
        else reset_check_start<SOLO>(rc, q, v, lane, ldsM);
    }
    if (DUO && wave == 1) {
        if constexpr (DUO) {
            T nosite[3];
            for (int sub = 0; sub < I.frame_skip; ++sub) {
                arm_front<SOLVE>(M, I, q, v, aw, sinq, cosq, rows, T(0), ldsM, lane, l8, nosite, diag, fs, ST, &rc);
                arm_back<SOLVE>(M, q, v, aw, sinq, cosq, ldsM, l8, fs, ST, lane, &rc, nullptr, record);
            }
        }
        return;
    }
    constexpr int R = DUO ? DYN : SOLO;
    const T u = l8 < A ? (T)action[l8] : T(0);
    const T tau_act = M.link(O_GEAR) * fmin(fmax(u, M.link(O_CTRL_LO)), M.link(O_CTRL_HI));
    T site[3];
    unsigned* dcount = diag;
    for (int sub = 0; sub < I.frame_skip; ++sub) {
        if (R == DYN && sub > 0) arm_back<R>(M, q, v, aw, sinq, cosq, ldsM, l8, fs, ST, lane, &rc, dcount, record);
        arm_front<R>(M, I, q, v, aw, sinq, cosq, rows, tau_act, ldsM, lane, l8, site, dcount, fs, ST, &rc);
        if constexpr (R == SOLO) arm_back<R>(M, q, v, aw, sinq, cosq, ldsM, l8, fs, ST, lane, &rc, dcount, record);
    }
    const int site_lane = lane_of_link(lane, I.site_link);
    for (int k = 0; k < 3; ++k) site[k] = __shfl(site[k], site_lane);
    if constexpr (R == DYN) arm_back<R>(M, q, v, aw, sinq, cosq, ldsM, l8, fs, ST, lane, &rc, dcount, record);
    if (rc.any && (rflags<R>(ldsM) & 4))    // the last substep ended in mj_checkAcc's reset: site_xpos is the reset state's
        for (int k = 0; k < 3; ++k) site[k] = (T)mo.reset_rec[2 * LANES + k];
    if (g == 0) {
        const T dx = site[0] - tgt[0], dy = site[1] - tgt[1], dz = site[2] - tgt[2];
        if (mo.step_cost && l8 == 0)
            ((T*)mo.step_cost)[0] = fabs(dx) + fabs(dy) + fabs(dz) + T(5) * sqrt_(dx * dx + dy * dy + dz * dz);
        if (mo.step_nobs) {
            T* o = (T*)mo.step_nobs;
            if (l8 < nv) { o[l8] = q; o[nv + l8] = v; }
            if (l8 < 3) {
                const T hh = l8 == 0 ? site[0] : (l8 == 1 ? site[1] : site[2]);
                const T gg = l8 == 0 ? tgt[0] : (l8 == 1 ? tgt[1] : tgt[2]);
                o[2 * nv + l8] = hh;
                o[2 * nv + 3 + l8] = hh - gg;
            }
        }
        if (l8 < nv) {
            st[l8] = (double)q;
            st[LANES + l8] = (double)v;
        }
    }
}

// Softmax statistics of this workgroup's particles -> record {max, S, W[H A]} with weights exp(x_p - max),
// x_p = -q0_p / lam (mppi.py:84-97), left in global memory for the finish kernel.
template <typename T, int NT>
__device__ __forceinline__ void mono_record(double lam, double* __restrict__ rec_out, int HA, double* red, const T* actT) {
    const int tid = threadIdx.x;
    __syncthreads();                // q0 of the particles and the action tile are complete
    if (tid < LANES) {              // eight lanes: max, weights, their sum (8-lane butterfly inside one DPP row)
        const double q = red[tid], x = q == INFINITY ? -INFINITY : (-1.0 / lam) * q;
        double m = x;
        for (int o = 1; o < LANES; o <<= 1) m = fmax(m, __shfl_xor(m, o));
        const double e = x == -INFINITY ? 0.0 : exp(x - m);
        double S = e;
        for (int o = 1; o < LANES; o <<= 1) S += __shfl_xor(S, o);
        red[LANES + tid] = e;
        if (tid == 0) { rec_out[0] = m; rec_out[1] = S; }
    }
    __syncthreads();
    double e[LANES];
#pragma unroll
    for (int k = 0; k < LANES; ++k) e[k] = red[LANES + k];
    for (int j = tid; j < HA; j += NT) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < LANES; ++k) acc += e[k] * (double)actT[k * HA + j];
        rec_out[2 + j] = acc;
    }
}

// The second (and last) launch of a fused control iteration: H workgroups, one per horizon row.  Workgroup t merges the
// records of all rollout workgroups for ITS row of the weighted action sum (thread tid walks the records tid, tid + 128, ...
// with a running maximum; the threads' partials are then rescaled to the workgroup's maximum and added in a fixed order:
// the same result whoever runs where), forms the new mean row (mppi.py:69-82) and writes it where the shift puts it
// (olgaussian_mpc.py:116-129) - into mean_out, a buffer of its own, because other workgroups still read mean_in.
// Workgroup 0's row is the action: it publishes it (device copy, mapped pinned host slot + completion flag), advances
// the step counter and then steps the device-resident real env with its two wavefronts in the DYN / SOLVE roles of the
// rollout; what that step needs from global memory (model block, state) is fetched at the top of the kernel, under the
// latency of the record loads.  Sharded runs (mo.record): the rows of this GPU's record {max, S, W} instead.
constexpr int MAX_A = 8;
constexpr int FIN_CHUNK = 4;        // records per thread in flight
template <typename T>
__global__ __launch_bounds__(128) void arm_mppi_finish_kernel(const T* __restrict__ model, const double* __restrict__ recs, long n_rec,
                                                              int H, int A, const double* __restrict__ mean_in,
                                                              double* __restrict__ mean_out, const MonoStep mo,
                                                              int env_step, unsigned* diag) {
    __shared__ __attribute__((aligned(16))) T lds[LANES * PSTRIDE + ARM_BLOB_LEN + 3];
    __shared__ double sh[2 * (2 + MAX_A)];
    const int tid = threadIdx.x, t = blockIdx.x, HA = H * A, rec = 2 + HA;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stepper = t == 0 && env_step && mo.state_io && !mo.record;
    // ---- loads first: the records of my first chunk, my mean row, and - the workgroup that steps the env - model and state
    double rm[FIN_CHUNK], rs[FIN_CHUNK], rw[FIN_CHUNK][MAX_A];
    auto fetch = [&](long base) {
#pragma unroll
        for (int k = 0; k < FIN_CHUNK; ++k) {
            const long b = base + tid + 128L * k;
            const double* r = recs + (b < n_rec ? b : 0) * rec;
            rm[k] = b < n_rec ? r[0] : -INFINITY;
            rs[k] = r[1];
#pragma unroll
            for (int a = 0; a < MAX_A; ++a) rw[k][a] = a < A ? r[2 + t * A + a] : 0.0;
        }
    };
    fetch(0);
    const double mean_old = (tid < A && !mo.record) ? mean_in[t * A + tid] : 0.0;
    const long long count = (t == 0 && mo.step_counter) ? *mo.step_counter : 0;
    const int l8 = lane_link(lane), g = lane_slot(lane);
    T q0 = T(0), v0 = T(0), tgt[3] = {T(0), T(0), T(0)};
    if (stepper) {
        for (int k = tid; k < ARM_BLOB_LEN; k += 128) lds[LANES * PSTRIDE + k] = model[k];
        for (int k = tid; k < LANES * PSTRIDE; k += 128) lds[k] = T(0);
        q0 = (T)mo.state_io[l8];
        v0 = (T)mo.state_io[LANES + l8];
        for (int k = 0; k < 3; ++k) tgt[k] = (T)mo.state_io[2 * LANES + k];
    }
    // ---- my share of the records, with a running maximum
    double m = -INFINITY, S = 0.0, W[MAX_A];
#pragma unroll
    for (int a = 0; a < MAX_A; ++a) W[a] = 0.0;
    for (long base = 0; base < n_rec; base += 128L * FIN_CHUNK) {
        if (base > 0) fetch(base);
        double mn = m;
#pragma unroll
        for (int k = 0; k < FIN_CHUNK; ++k) mn = fmax(mn, rm[k]);
        const double sc = m == -INFINITY ? 0.0 : exp(m - mn);
        S *= sc;
#pragma unroll
        for (int a = 0; a < MAX_A; ++a) W[a] *= sc;
#pragma unroll
        for (int k = 0; k < FIN_CHUNK; ++k) {
            const double c = rm[k] == -INFINITY ? 0.0 : exp(rm[k] - mn);
            S += c * rs[k];
#pragma unroll
            for (int a = 0; a < MAX_A; ++a) W[a] += c * rw[k][a];
        }
        m = mn;
    }
    // ---- the workgroup's maximum, then the rescaled partials added up (wave butterflies, one LDS hop)
    double M = m;
    for (int o = 32; o > 0; o >>= 1) M = fmax(M, __shfl_xor(M, o));
    if (lane == 0) sh[wave] = M;
    __syncthreads();
    M = fmax(sh[0], sh[1]);
    const double f = m == -INFINITY ? 0.0 : exp(m - M);
    S *= f;
    for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o);
#pragma unroll
    for (int a = 0; a < MAX_A; ++a) {
        W[a] *= f;
        for (int o = 32; o > 0; o >>= 1) W[a] += __shfl_xor(W[a], o);
    }
    __syncthreads();
    if (lane == 0) {
        sh[wave * (2 + MAX_A) + 1] = S;
#pragma unroll
        for (int a = 0; a < MAX_A; ++a) sh[wave * (2 + MAX_A) + 2 + a] = W[a];
    }
    __syncthreads();
    S = sh[1] + sh[(2 + MAX_A) + 1];
    const double wsum = tid < A ? sh[2 + tid] + sh[(2 + MAX_A) + 2 + tid] : 0.0;
    if (mo.record) {
        if (tid < A) mo.record[2 + t * A + tid] = wsum;
        if (t == 0 && tid == 0) { mo.record[0] = M; mo.record[1] = S; }
        return;
    }
    double nm = 0.0;
    if (tid < A) {
        nm = (1.0 - mo.step_size) * mean_old + mo.step_size * (wsum / S);
        if (mo.shift_mode < 0) {
            mean_out[t * A + tid] = nm;
        } else {
            if (t > 0) mean_out[(t - 1) * A + tid] = nm;
            if (t == H - 1) mean_out[t * A + tid] = mo.shift_mode == 0 ? 0.0 : nm;       // 'null' / 'repeat'
        }
    }
    if (t != 0) return;
    // ---- the action: publish, count the step, step the real env
    __syncthreads();
    if (tid < A) sh[tid] = nm;
    double* slot = mo.action_host ? mo.action_host + (count & 1) * (A + 1) : nullptr;
    if (tid < A) {
        if (mo.action_out) mo.action_out[tid] = nm;
        // mapped pinned host memory, system-scope stores: the action, then - once they are acknowledged - the new step
        // count as the completion flag (two slots: a launch enqueued ahead cannot overwrite an action not yet read)
        if (slot) __hip_atomic_store(slot + tid, nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (wave == 0) {                // (the wave that wrote the action waits for the acknowledgement; the other goes on)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            if (mo.step_counter) *mo.step_counter = count + 1;              // noise stream of the next step
            if (slot) __hip_atomic_store(slot + A, (double)(count + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!stepper) return;
    __syncthreads();                // action in LDS, model block staged, particle blocks zeroed
    Model<T, true> Mv{lds + LANES * PSTRIDE, l8};
    Mv.cache();
    ArmInts I;
    I.site_link = (int)lds[LANES * PSTRIDE + O_SITE_LINK];
    I.n_sphere = (int)lds[LANES * PSTRIDE + O_N_SPHERE];
    I.sph_link = (int)lds[LANES * PSTRIDE + O_SPH_LINK];
    I.frame_skip = (int)lds[LANES * PSTRIDE + O_FRAME_SKIP];
    I.nv = (int)lds[LANES * PSTRIDE + O_NV];
    I.site_link = __builtin_amdgcn_readfirstlane(I.site_link);
    I.n_sphere = __builtin_amdgcn_readfirstlane(I.n_sphere);
    I.sph_link = __builtin_amdgcn_readfirstlane(I.sph_link);
    I.frame_skip = __builtin_amdgcn_readfirstlane(I.frame_skip);
    I.nv = __builtin_amdgcn_readfirstlane(I.nv);
    real_env_step<T, true>(Mv, I, mo, mo.state_io, q0, v0, tgt, sh, A, lds, lane, wave, l8, g, diag);
}

// ---- the rollout kernel -------------------------------------------------------------------------
// state: f64 [qpos(8) | qvel(8) | target(3)]; mean: f64 [H][A]; noise/cost/act/obs/next_obs: T, in the
// reference's C-order layouts (P,H,A) / (P,H) / (P,H,2nv+6).  noise, act, obs, next_obs may be null.
// STEP = true is the same code instantiated under its own name for the single-particle "real env" step
// (mjmpc_arm_step_state), so that profiler statistics of the P-particle rollout are not diluted by it.
// Register budget: f64 needs ~250 VGPRs (two waves per SIMD), f32 ~140 (three; a budget of 128 spills) -
// occupancy carries the kernel once P exceeds ~8k particles
// CL = true: the closed-loop-linear policy variant, its own instantiation so that the open-loop kernel does
// not carry its registers (136 -> 178 VGPRs in f32 when both lived in one kernel)
// WAVES = cap on resident waves per SIMD.  The f32 build fits three, but when a launch has no more than two waves
// per SIMD to offer the dispatcher packs some SIMDs with three and leaves others with one, and the launch lasts as
// long as the crowded ones (measured inside the control loop at 16 384 particles: 0.42 ms against 0.31 ms) - so
// such launches use the instantiation capped at two.
// DUO = two wavefronts per particle group (roles DYN / SOLVE above), for launches of at most half a wave per SIMD.
// MONO = launch 1 of the two-launch control iteration (struct MonoStep, arm_rollout.h, by value): samples drawn in the
// kernel, actions kept in LDS, ONE softmax record {max, S, W[H A]} per workgroup left in the engine's record buffer; the
// merge, mean update, action, shift and real-env step are launch 2 (arm_mppi_finish_kernel).  (An in-kernel merge behind
// arrival counters was measured and dropped: agent-scope fences / dependent L2 round trips cost more than a launch.)
// (the kernels' body: arm_rollout_body.inc, #included into the two __global__ functions below)

template <typename T, bool STEP, bool CL, int WAVES, bool DUO, bool MONO>
__global__ __launch_bounds__(DUO ? 128 : 64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void arm_rollout_kernel(const T* __restrict__ model, const double* state,
                                                         long P, int H, int A, const double* mean,
                                                         const T* __restrict__ noise, T* __restrict__ cost,
                                                         T* __restrict__ act, T* __restrict__ obs,
                                                         T* __restrict__ nobs, double* state_out, unsigned* diag,
                                                         RolloutFusion fuse, const MonoStep mono_arg) {
    const MonoStep* mop = &mono_arg;      // (a kernel argument: its fields arrive with the other arguments, no pointer chase)
    constexpr int SHAPE = DUO ? 2 : 1, NW = SHAPE;
    constexpr bool QUAD = false;
    const qflag_ptr qf = nullptr;
    int seq = 0;
#include "arm_rollout_body.inc"
}
#ifndef ARM_NO_FLAGS_CODE
// The flag-synchronised shapes: NW = 2 (a 128-thread workgroup, as DUO) and NW = 4 (P <= 2048 on 256 CUs: a 256-thread
// workgroup = the four waves of a particle group, one per SIMD of a CU)
template <typename T, int NW, bool MONO>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void arm_rollout_flags_kernel(const T* __restrict__ model, const double* state,
                                                         long P, int H, int A, const double* mean,
                                                         const T* __restrict__ noise, T* __restrict__ cost,
                                                         T* __restrict__ act, T* __restrict__ obs,
                                                         T* __restrict__ nobs, unsigned* diag,
                                                         RolloutFusion fuse, const MonoStep mono_arg) {
    const MonoStep* mop = &mono_arg;
    constexpr int SHAPE = 10 + NW;
    constexpr bool STEP = false, CL = false, DUO = false, QUAD = true;
    double* const state_out = nullptr;
    __shared__ int qflags[QF_COUNT];        // hand-over flags of the workgroup's waves (flag_front)
    qflag_ptr qf = (qflag_ptr)qflags;
    int seq = 0;                            // substep sequence number
#include "arm_rollout_body.inc"
}

#endif
}  // namespace

// (the extended-joint build exports the same two entry points under its own names: arm_rollout.h)
#ifdef MJMPC_ARM_XJ
#define launch_arm_mppi_finish launch_arm_mppi_finish_xj
#define launch_arm_rollout launch_arm_rollout_xj
#else
long arm_rollout_groups(long P) { return (P + LANES - 1) / LANES; }
long mono_record_doubles(long groups, int H, int A) { return groups * (2 + (long)H * A); }
#endif

template <typename T>
hipError_t launch_arm_mppi_finish(const T* model, const double* records, long n_rec, int H, int A, const double* mean_in,
                                  double* mean_out, const MonoStep& mono, int env_step, unsigned* diag, hipStream_t stream) {
    if (A > MAX_A || H < 1 || n_rec < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(arm_mppi_finish_kernel<T>, dim3((unsigned)H), dim3(128), 0, stream, model, records, n_rec, H, A, mean_in,
                       mean_out, mono, env_step, diag);
    return hipGetLastError();
}
template hipError_t launch_arm_mppi_finish<float>(const float*, const double*, long, int, int, const double*, double*,
                                                  const MonoStep&, int, unsigned*, hipStream_t);
template hipError_t launch_arm_mppi_finish<double>(const double*, const double*, long, int, int, const double*, double*,
                                                   const MonoStep&, int, unsigned*, hipStream_t);

template <typename T>
hipError_t launch_arm_rollout(const T* model, const double* state, long P, int H, int A, const double* mean,
                              const T* noise, T* cost, T* act, T* obs, T* nobs, double* state_out,
                              unsigned* diag, hipStream_t stream, RolloutFusion fuse,
                              const MonoStep* mono) {
    if (P <= 0 || H <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((P + LANES - 1) / LANES);
    // Cap the resident waves per SIMD at what this launch needs (2, or - f32 only, f64 does not fit - 3): the
    // dispatcher then has to spread the workgroups evenly instead of crowding some SIMDs.  (A cap of 1 for launches
    // of at most one wave per SIMD measured 3 % slower than 2.)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long simds = 4L * cus;
    const int cap = ((long)grid <= 2 * simds || sizeof(T) == 8) ? 2 : 3;
    // two wavefronts per particle group while that still leaves every wave a SIMD of its own (P <= 4096 on 256 CUs);
    // those kernels are capped at ONE resident wave per SIMD, which makes the dispatcher spread the 2 x 2 waves of
    // the two workgroups a CU receives over its four SIMDs (0.253 -> 0.227 ms at 4096 particles with the cap).
    // MJMPC_ARM_DUO=0/1 overrides the choice (developer switch for A/B timing).
    static const int duo_env = [] { const char* e = getenv("MJMPC_ARM_DUO"); return e ? atoi(e) : -1; }();
    const bool duo = !fuse.clw && (duo_env >= 0 ? duo_env != 0 : 2L * grid <= simds);
    // between half a wave and one wave per SIMD (4096 < P <= 8192 on 256 CUs) the one-wave kernel is capped at ONE
    // resident wave per SIMD as well: inside the replayed control iteration the dispatcher otherwise doubles waves up on
    // some SIMDs while others idle (8192 particles: 361 us per launch in the loop against 280 us back to back)
    static const int one_env = [] { const char* e = getenv("MJMPC_ARM_ONE"); return e ? atoi(e) : -1; }();
    const bool one_per_simd = !duo && !fuse.clw && (one_env >= 0 ? one_env != 0 : (long)grid <= simds);
    const MonoStep mono_arg = mono ? *mono : MonoStep();
    const size_t dyn = mono ? sizeof(double) * MONO_RED + sizeof(T) * LANES * (size_t)H * A : 0;
#ifndef ARM_NO_FLAGS_CODE
    // Round 6: the flag-synchronised shape with four wavefronts per particle group while every one of them still has a SIMD
    // of its own (P <= 2048 on 256 CUs) - the launches that are BASELINE config 2 and the shards of a strong-scaling run.
    // Above that DUO stays (the two-wave flag shape measures 4 % slower than DUO at 4096 particles: profiles/r06_arm_shapes.txt).
    // The real-env step (state_out) keeps DUO.  MJMPC_ARM_FLAGS=0 (DUO) / 2 / 4 overrides the choice (developer A/B switch).
    static const int flags_env = [] { const char* e = getenv("MJMPC_ARM_FLAGS"); return e ? atoi(e) : -1; }();
    int nw = 0;
    if (duo && !state_out && duo_env < 0) nw = flags_env >= 0 ? flags_env : (4L * grid <= simds ? 4 : 0);
    if (nw == 2 || nw == 4) {
        if (mono) {
            if (obs || nobs || !fuse.gseq || !mono->chol || !mono->tree) return hipErrorInvalidValue;
            if (dyn + sizeof(T) * (LANES * PSTRIDE + ARM_BLOB_LEN + 3) > 64 * 1024) return hipErrorInvalidValue;
        }
#define MJMPC_LAUNCH_F(NW_, MONO_)                                                                                        \
        hipLaunchKernelGGL((arm_rollout_flags_kernel<T, NW_, MONO_>), dim3(grid), dim3(64 * NW_), dyn, stream, model, state, P, H, A, \
                           mean, noise, cost, act, obs, nobs, diag, fuse, mono_arg)
        if (nw == 4) { if (mono) MJMPC_LAUNCH_F(4, true); else MJMPC_LAUNCH_F(4, false); }
        else { if (mono) MJMPC_LAUNCH_F(2, true); else MJMPC_LAUNCH_F(2, false); }
#undef MJMPC_LAUNCH_F
        return hipGetLastError();
    }
#endif
#define MJMPC_LAUNCH_W(STEP_, CL_, W_, DUO_, MONO_)                                                                   \
    hipLaunchKernelGGL((arm_rollout_kernel<T, STEP_, CL_, W_, DUO_, MONO_>), dim3(grid), dim3(DUO_ ? 128 : 64), dyn,  \
                       stream, model, state, P, H, A, mean, noise, cost, act, obs, nobs, state_out, diag, fuse, mono_arg)
#define MJMPC_LAUNCH(STEP_, CL_, MONO_)                                 \
    do {                                                                \
        if (one_per_simd) {                                             \
            MJMPC_LAUNCH_W(STEP_, CL_, 1, false, MONO_);                \
        } else if constexpr (sizeof(T) == 8) {                          \
            MJMPC_LAUNCH_W(STEP_, CL_, 2, false, MONO_);                \
        } else {                                                        \
            if (cap == 2) MJMPC_LAUNCH_W(STEP_, CL_, 2, false, MONO_);  \
            else MJMPC_LAUNCH_W(STEP_, CL_, 3, false, MONO_);           \
        }                                                               \
    } while (0)
    if (mono) {
        if (fuse.clw || state_out || obs || nobs || !fuse.gseq || !mono->chol || !mono->tree)
            return hipErrorInvalidValue;
        if (dyn + sizeof(T) * (LANES * PSTRIDE + ARM_BLOB_LEN + 3) > 64 * 1024) return hipErrorInvalidValue;     // H A too large for the LDS tile
        if (duo) MJMPC_LAUNCH_W(false, false, 1, true, true);
        else MJMPC_LAUNCH(false, false, true);
    }
    else if (fuse.clw) MJMPC_LAUNCH(false, true, false);
    else if (duo && state_out) MJMPC_LAUNCH_W(true, false, 1, true, false);
    else if (duo) MJMPC_LAUNCH_W(false, false, 1, true, false);
    else if (state_out) MJMPC_LAUNCH(true, false, false);
    else MJMPC_LAUNCH(false, false, false);
#undef MJMPC_LAUNCH
#undef MJMPC_LAUNCH_W
    return hipGetLastError();
}

template hipError_t launch_arm_rollout<float>(const float*, const double*, long, int, int, const double*,
                                              const float*, float*, float*, float*, float*, double*, unsigned*,
                                              hipStream_t, RolloutFusion, const MonoStep*);
template hipError_t launch_arm_rollout<double>(const double*, const double*, long, int, int, const double*,
                                               const double*, double*, double*, double*, double*, double*, unsigned*,
                                               hipStream_t, RolloutFusion, const MonoStep*);

}  // namespace mjmpc

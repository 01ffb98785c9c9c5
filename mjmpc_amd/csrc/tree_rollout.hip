// Fused rollout kernel for a kinematic TREE of hinge / slide links: the same
//   for b in particles: for t in horizon: env.step(mean[t] + noise[b,t])
// double loop as arm_rollout.hip (reference mjmpc/envs/gym_env_wrapper.py:125-153, MuJoCo mj_step inlined), for
// models the 8-lane serial-chain kernel cannot hold: up to 32 dofs on a branching tree (a hand on an arm; the
// reference's vendored swimmer.xml and half_cheetah.xml with their slide/slide/hinge floating roots), gravity, joint
// limits and springs, motors or position servos on a subset of the joints, the inertia-box fluid model, up to 16 contact
// points - sphere / capsule end against the plane, or sphere / capsule geom-geom pairs (an object in a hand; the swimmer's
// self-collision) - with frictionless rows or pyramidal friction cones.  Cost and observation follow the model's task: the
// reacher's (reacher_env.py:29-47), forward progress (swimmer.py:10-24, half_cheetah.py:10-25) or the shape of pen-v0's
// reorientation reward (examples/configs/hand/pen-v0.yml:8).
//
// Execution model: ONE PARTICLE = 32 LANES (lane = link = dof, links numbered depth-first), two particles per
// wavefront - or 16 lanes (one DPP row) and four particles per wavefront for models of up to 16 dofs, which also factor
// densely in registers (tree_rollout_dense.hip) - and up to four wavefronts per workgroup sharing one LDS copy of the
// model's constants.  Lanes talk through a small per-particle LDS area (a wavefront owns its area: in-order LDS, no
// s_barrier).  Everything is expressed in world
// coordinates about the world origin, so the tree recursions become
//   root-to-link accumulations  (forward kinematics, spatial velocity, velocity-product acceleration)
//       = pointer jumping over the ancestor tables (ceil(log2 depth) rounds of "read my 2^k-th ancestor"),
//   link-to-leaves accumulations (Newton-Euler forces, composite inertias)
//       = range sums over the depth-first numbering: a doubling table T_k[i] = x_i + ... + x_{i+2^k-1} is built in
//         log2(32) rounds and every link adds the blocks that tile [i, i + subtree size) - exact (no cancelling
//         differences of prefix sums) and the same instruction stream for every topology.
// Linear algebra: the mass matrix of a tree is sparse - M[i][j] = 0 unless j lies on i's path to the root (or the
// other way round) - and MuJoCo's L'DL factorisation in leaves-first order creates no fill-in.  Lane i keeps its row
// PATH-INDEXED in registers, r[c] = M[i][ancestor at distance c] (c < DP, the longest path; 8 for the 24-dof hand
// instead of 24 columns), and links of equal HEIGHT above their deepest leaf, which are mutually unrelated, are
// eliminated TOGETHER: one round per height (8 rounds, not 24 pivots), every lane pulling the rows of its descendants
// of that height from LDS (a host-built list per lane).  H = M + J'DJ keeps the pattern (a contact row couples only
// dofs on one path), so the constraint solver and the Euler solve use the same factorisation; the two triangular
// solves run leaves-first (L') and root-first (L), one LDS round per height / depth - except for the TRUNK (the chain of
// single-child links from the root), which is factored and solved in registers with DPP broadcasts.  A geom-geom contact
// between two trees is kept inside that pattern by an ELIMINATION tree that hangs the manipulator under the object
// (compile_tree.py).  The soft-constraint problem is the arm kernel's primal active-set Newton iteration with several
// contact rows; in the friction instantiation an exact line search takes over when it does not settle (MuJoCo's Newton).
// Round 4: the GENERAL instantiation (GEN: ball / free joints as quaternion links, friction-loss rows, boxes, static geoms,
// equalities, tendon limits, solver parameters per row, affine actuators) and a DENSE factorisation over the 32 lanes of
// a particle (DN = 32: models of 17 .. 32 dofs whose elimination paths are longer than 8 links) - see dense32_factor.
// Round 6: every decision of the solver is taken PER PARTICLE (a converged particle is frozen while its wave-mates iterate on;
// rank-one correction or refactorisation by the particle's own changes; the long sine / cosine path per lane), so a particle's
// trajectory does not depend on who shares its wavefront (tests/test_wave_mates_gpu.py); the exact line search finds its root
// by false position and takes over earlier in models with friction-loss rows.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <type_traits>

#include "lanegroup.h"
#include "tree_model.h"
#include "tree_rollout.h"

namespace mjmpc {
namespace {

constexpr int TREE_MAXIT = 16;
// the friction instantiations' safeguarded iteration (exact line search from iteration LS_START on: MuJoCo's Newton method
// proper, which terminates - a finite number of active sets, a strictly decreasing convex cost) may run on to MuJoCo's own
// cap: round 3 stopped it at 16 and counted a handful of particle-substeps per 10^8 on the pen-in-hand model as failures
constexpr int TREE_MAXIT_LS = 100;
#ifndef TREE_DPP_SUBTREE
#define TREE_DPP_SUBTREE 1
#endif
// developer switch for phase timing (tools/tree_time.py with a build -DTREE_SKIP=bits): 1 = no Newton iteration,
// 2 = no Euler factor/solve, 4 = no mass-matrix assembly, 8 = no bias forces, 16 = the constraint stage without its iterations,
// 32 = with exactly one, 64 = no cone line search.  Product builds: 0.
#ifndef TREE_SKIP
#define TREE_SKIP 0
#endif
// developer builds (-DTREE_STATS, tools/tree_stats.py): shader-clock per phase and Newton iteration counts of the first
// particle of the launch, accumulated behind the failure counter (diag + 2 ... as 64-bit words)
#ifdef TREE_STATS
struct TreeClock {
    unsigned long long t;
    unsigned long long* out;
    __device__ void start(unsigned* diag, bool on) { out = on ? (unsigned long long*)(diag + 2) : nullptr; t = __builtin_readcyclecounter(); }
    __device__ void mark(int slot) {
        const unsigned long long n = __builtin_readcyclecounter();
        if (out) atomicAdd(out + slot, n - t);
        t = n;
    }
    __device__ void count(int slot, unsigned v) { if (out) atomicAdd(out + slot, (unsigned long long)v); }
    unsigned long long t2;      // a second clock for the parts of the Newton iteration (slots 12 ...): lap(-1) starts it
    __device__ void lap(int slot) {
        const unsigned long long n = __builtin_readcyclecounter();
        if (out && slot >= 0) atomicAdd(out + slot, n - t2);
        t2 = n;
    }
};
#else
struct TreeClock {
    __device__ void start(unsigned*, bool) {}
    __device__ void mark(int) {}
    __device__ void count(int, unsigned) {}
    __device__ void lap(int) {}
};
#endif

// laps inside a Newton iteration (tools/tree_stats.py): slots 15 / 22 / 23 = next active set, line search, rank-one correction;
// with -DTREE_STATS_FINE the walk of the owners' residuals, the sets made from them, and everything after
#ifdef TREE_STATS_FINE
#define TREE_LAP_A clk.lap(15)
#define TREE_LAP_B clk.lap(22)
#define TREE_LAP_C clk.lap(26)
#define TREE_FLAP(slot_) clk.lap(slot_)         /* slots 24 ...: see tools/tree_stats.py --fine */
#else
#define TREE_LAP_A
#define TREE_LAP_B clk.lap(15)
#define TREE_LAP_C clk.lap(22)
#define TREE_FLAP(slot_)
#endif
constexpr int wg_waves(int DP, bool fric, int scalar_bytes = 8, int PL = 32) {
    return DP <= 8 ? 4 : (DP <= 16 ? (PL == 16 && scalar_bytes == 4 ? 4 : 2) : 1);
}

#define TSYNC()                                                \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

// per-particle LDS area, in scalars (DP = compile-time bound on the links of a root-to-leaf path)
constexpr int A_X = 0;                      // 12 x 32 exchange (kinematics, path sums, doubling tables)
constexpr int a_sf(int PL) { return A_X + 6 * PL; }     // S[6][PL], beside the first half of the exchange area (used apart from it)
constexpr int A_ROW = 0;                    // path-indexed rows [32][row_stride]; overlays A_X / A_SF
// row stride: 2 DP entries (so that dist + c never leaves the row) + 1, an odd number of
// doubles: 32 lanes reading 32 different rows at the same offset then hit 32 different bank pairs (a stride of 16
// doubles = 128 B put them on two: measured 16-way conflicts)
constexpr int row_stride(int DP) { return DP + 1; }
constexpr int TILE_STRIDE = 17;     // the dense 16 x 16 tile of a 16-lane particle (odd stride: conflict-free columns)
constexpr int a_max(int a, int b) { return a > b ? a : b; }
// the row area also holds the exchange buffers (12 x PL) and, for 16-lane particles, the dense tile; a row read may
// run up to DP - 1 entries past the last row (see tree_factor): the broadcast vector behind it is finite data too
constexpr int a_vec(int DP, int PL) {
    return (a_max(a_max(row_stride(DP) * PL, 12 * PL), PL == 16 ? TILE_STRIDE * 16 : 0) + 3) & ~3;     // broadcast vector [PL]
}
// contact Jacobian rows [NS][NJ][DP], PATH-INDEXED like the matrix rows: a contact point on link L moves only with the
// dofs on L's path to the root, entry c belongs to L's ancestor at distance c (a quarter of a [32]-lane row at DP = 8)
constexpr int a_jc(int DP, int PL) { return a_vec(DP, PL) + PL; }
constexpr int CS_BASE = 15; // per contact point: sphere centre (lean instantiation) / contact point (full)[3], dist, D, aref
                            // (normal part), mu B Jt1.v, mu B Jt2.v, [8:11] contact normal (geom-geom; after the Newton
                            // iteration: the point's force sums on Jn, mu Jt1, mu Jt2), [11:14] first tangent of the
                            // contact frame (mju_makeFrame from the normal and the capsule axis), -
// general instantiation: five more scalars per record - [15:18] the three rows' D of a connect equality (their reference
// accelerations in [5:8]; the two anchors in [0:3] and [11:14]), [18:20] spare
constexpr int CS_GEN = 20;
// elliptic friction cones (GEN = 3, round 5): once the rows are built, a contact record's [11:20] hold the cone's quadratic
// model at the solver's base point - [14:20] the symmetric 3 x 3 weight W over (normal, tangent 1, tangent 2) as 00 01 02 11
// 12 22, [11:14] the vector b of  H = M + sum J' W J,  rhs = tau + sum J' b  (see cone_model; the tangent in [11:14] has
// served by then, the geometry stage writes it afresh every substep)
constexpr int CS_W = 14, CS_B = 11;
constexpr int a_cs(int DP, int NS, int NJ, int PL) { return a_jc(DP, PL) + NS * NJ * DP; }
constexpr int a_misc(int DP, int NS, int NJ, int PL, int CSZ = CS_BASE) { return a_cs(DP, NS, NJ, PL) + NS * CSZ; }   // site[3]
constexpr int a_row2(int DP, int NS, int NJ, int PL, int CSZ = CS_BASE) { return a_misc(DP, NS, NJ, PL, CSZ) + 8; }     // the Euler matrix's factor
// the full instantiation with rows of up to 16 entries factors the Euler matrix beside the first Newton matrix (two
// 32-entry rows do not fit the register file; the lean instantiation runs the large launches, where the second row area
// would cost a resident workgroup per CU: 65536 x 64 on the hand 77 -> 129 ms - as it would in the f32 launches with 16
// lanes per particle, which are the large ones: 32768 x 32 on the cheetah 18 -> 35 ms; f64 rows of 16: three such rows
// at once spill - 256 VGPRs + 153 AGPRs + 80 bytes of scratch - and the pen-in-hand launch goes from 17.0 to 31.7 ms)
// (DN > 0: 16-lane particles factor densely in registers - see dense_factor - and have nothing to merge)
constexpr bool merge_factor(int DP, bool fric, int scalar_bytes, int PL, int DN = 0) {
    return DN == 0 && (fric || DP <= 8) && DP <= 16 && !(scalar_bytes == 4 && PL == 16) && !(scalar_bytes == 8 && DP > 8);
}
constexpr int a_len(int DP, int NS, int NJ, int PL, int scalar_bytes, int DN = 0, int CSZ = CS_BASE) {
    return a_row2(DP, NS, NJ, PL, CSZ) + (merge_factor(DP, NJ == 3, scalar_bytes, PL, DN) ? row_stride(DP) * PL : 0);
}
static_assert(TILE_STRIDE * 16 <= a_vec(8, 16), "the dense tile lives in the (otherwise unused) row area");

template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T>
__device__ __forceinline__ T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T>
__device__ __forceinline__ void mv3(const T* R, const T* x, T* y) {
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
template <typename T>
__device__ __forceinline__ void symv3(const T* S, const T* x, T* y) {      // xx yy zz xy xz yz
    y[0] = S[0] * x[0] + S[3] * x[1] + S[4] * x[2];
    y[1] = S[3] * x[0] + S[1] * x[1] + S[5] * x[2];
    y[2] = S[4] * x[0] + S[5] * x[1] + S[2] * x[2];
}

// the two 16-lane rows of a particle's 32 lanes added up, result in both: v_permlane16_swap (gfx950) exchanges the odd
// rows of its first operand with the even rows of its second, so (x, x) comes back as (even-row value, odd-row value)
__device__ __forceinline__ float add_rows(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double add_rows(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// the value the same lane of the particle's other 16-lane row holds
__device__ __forceinline__ float swap_rows(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);       // (r = {even-row value, odd-row value} at my position)
}
__device__ __forceinline__ double swap_rows(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = threadIdx.x & 16;
    return __hiloint2double((int)(odd ? b[0] : b[1]), (int)(odd ? a[0] : a[1]));
}
// sum over the 32 lanes of a particle, result in every lane: four DPP steps inside the 16-lane rows and one row swap
// (no LDS crossbar: a ds_bpermute butterfly costs five times as much)
template <int PL, typename T>
__device__ __forceinline__ T sum_lanes(T x) {
    x += dpp_all<0xB1>(x);          // quad_perm [1,0,3,2]
    x += dpp_all<0x4E>(x);          // quad_perm [2,3,0,1]
    x += dpp_all<0x141>(x);         // row_half_mirror: i <-> 7 - i
    x += dpp_all<0x140>(x);         // row_mirror: i <-> 15 - i
    return PL == 32 ? add_rows(x) : x;      // (16 lanes per particle: a particle is one DPP row)
}

// bitwise OR over the lanes of a particle, result in every lane (same butterfly as sum_lanes)
template <int PL>
__device__ __forceinline__ unsigned or_lanes(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);
    if (PL == 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        x = r[0] | r[1];
    }
    return x;
}

// ---- DPP row broadcasts (a 16-lane row: a whole 16-lane particle, or the first half of a 32-lane one) -----------
template <int K>
__device__ __forceinline__ float bcast_row(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xF, 0xF, false));
}
template <int K>
__device__ __forceinline__ double bcast_row(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + K, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + K, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// acc += (lane K's x) * m  in ONE instruction: v_fmac with a DPP row_newbcast source (f64 DPP knows no other control).
// The two wait states a DPP read wants after a VALU write of the same register (GFX9 hazard; the compiler cannot see
// into the asm) are spelled out.
template <int K>
__device__ __forceinline__ void fma_bcast(double& acc, double x, double m) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(K));
}
template <int K>
__device__ __forceinline__ void fma_bcast(float& acc, float x, float m) {
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(K));
}
// GROUPS of such broadcast FMAs as ONE asm statement each: the wait states once, in front, and nothing the compiler could
// put between the instructions.  (Dropping the wait states from single-instruction statements whose sources "are only
// read in the surrounding sequence" measured 10 % faster on the cheetah and was withdrawn: the compiler may reload such
// a register from an AGPR, or finish computing it, right before the statement - f32 HalfCheetah failed its parity test
// that way.  Inside one statement that cannot happen; a source written by an EARLIER instruction of the group is a
// plain VALU dependency, which the hardware interlocks - only the DPP read of a freshly written register is not.)
#define MJMPC_BC " row_newbcast:%"
#define MJMPC_BCT " row_mask:0xf bank_mask:0xf\n\t"
#define MJMPC_DPP_GROUPS(T_, SFX_)                                                                                          \
    /* acc += (lane K's x0) * m0 + (lane K's x1) * m1 + (lane K's x2) * m2 */                                               \
    template <int K>                                                                                                        \
    __device__ __forceinline__ void fma_bcast_3(T_& acc, T_ x0, T_ x1, T_ x2, T_ m0, T_ m1, T_ m2) {                       \
        asm volatile("s_nop 1\n\t"                                                                                        \
                     "v_fmac_" SFX_ "_dpp %0, %1, %4" MJMPC_BC "7" MJMPC_BCT                                                \
                     "v_fmac_" SFX_ "_dpp %0, %2, %5" MJMPC_BC "7" MJMPC_BCT                                                \
                     "v_fmac_" SFX_ "_dpp %0, %3, %6" MJMPC_BC "7 row_mask:0xf bank_mask:0xf"                               \
                     : "+v"(acc) : "v"(x0), "v"(x1), "v"(x2), "v"(m0), "v"(m1), "v"(m2), "n"(K));                          \
    }                                                                                                                       \
    /* a_i += (lane K's a_i) * m, four accumulators */                                                                      \
    template <int K>                                                                                                        \
    __device__ __forceinline__ void fma_bcast_self4(T_& a0, T_& a1, T_& a2, T_& a3, T_ m) {                                \
        asm volatile("s_nop 1\n\t"                                                                                        \
                     "v_fmac_" SFX_ "_dpp %0, %0, %4" MJMPC_BC "5" MJMPC_BCT                                                \
                     "v_fmac_" SFX_ "_dpp %1, %1, %4" MJMPC_BC "5" MJMPC_BCT                                                \
                     "v_fmac_" SFX_ "_dpp %2, %2, %4" MJMPC_BC "5" MJMPC_BCT                                                \
                     "v_fmac_" SFX_ "_dpp %3, %3, %4" MJMPC_BC "5 row_mask:0xf bank_mask:0xf"                               \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "n"(K));                                           \
    }                                                                                                                       \
    /* acc[c] += (lane K's x[c]) * m, six components */                                                                     \
    template <int K>                                                                                                        \
    __device__ __forceinline__ void fma_bcast_vec6(T_* acc, const T_* x, T_ m) {                                           \
        asm volatile("s_nop 1\n\t"                                                                                        \
                     "v_fmac_" SFX_ "_dpp %0, %6, %12" MJMPC_BC "13" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %7, %12" MJMPC_BC "13" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %2, %8, %12" MJMPC_BC "13" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %3, %9, %12" MJMPC_BC "13" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %4, %10, %12" MJMPC_BC "13" MJMPC_BCT                                             \
                     "v_fmac_" SFX_ "_dpp %5, %11, %12" MJMPC_BC "13 row_mask:0xf bank_mask:0xf"                            \
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])                  \
                     : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(m), "n"(K));                 \
    }                                                                                                                       \
    /* up += (lane K's s) . f,  dn += (lane K's f) . s   over six components */                                             \
    template <int K>                                                                                                        \
    __device__ __forceinline__ void fma_bcast_dots6(T_& up, T_& dn, const T_* s, const T_* f) {                            \
        asm volatile("s_nop 1\n\t"                                                                                        \
                     "v_fmac_" SFX_ "_dpp %0, %2, %8" MJMPC_BC "14" MJMPC_BCT                                               \
                     "v_fmac_" SFX_ "_dpp %1, %8, %2" MJMPC_BC "14" MJMPC_BCT                                               \
                     "v_fmac_" SFX_ "_dpp %0, %3, %9" MJMPC_BC "14" MJMPC_BCT                                               \
                     "v_fmac_" SFX_ "_dpp %1, %9, %3" MJMPC_BC "14" MJMPC_BCT                                               \
                     "v_fmac_" SFX_ "_dpp %0, %4, %10" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %10, %4" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %0, %5, %11" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %11, %5" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %0, %6, %12" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %12, %6" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %0, %7, %13" MJMPC_BC "14" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %13, %7" MJMPC_BC "14 row_mask:0xf bank_mask:0xf"                             \
                     : "+v"(up), "+v"(dn)                                                                                   \
                     : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(f[0]), "v"(f[1]), "v"(f[2]),  \
                       "v"(f[3]), "v"(f[4]), "v"(f[5]), "n"(K));                                                           \
    }                                                                                                                       \
    /* up += (lane K's sb) . f,  dn += (lane K's fb) . s   - the broadcast sources apart from the multipliers */            \
    template <int K>                                                                                                        \
    __device__ __forceinline__ void fma_bcast_dots6x(T_& up, T_& dn, const T_* sb, const T_* fb, const T_* s, const T_* f) { \
        asm volatile("s_nop 1\n\t"                                                                                        \
                     "v_fmac_" SFX_ "_dpp %0, %2, %20" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %8, %14" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %0, %3, %21" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %9, %15" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %0, %4, %22" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %10, %16" MJMPC_BC "26" MJMPC_BCT                                             \
                     "v_fmac_" SFX_ "_dpp %0, %5, %23" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %11, %17" MJMPC_BC "26" MJMPC_BCT                                             \
                     "v_fmac_" SFX_ "_dpp %0, %6, %24" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %12, %18" MJMPC_BC "26" MJMPC_BCT                                             \
                     "v_fmac_" SFX_ "_dpp %0, %7, %25" MJMPC_BC "26" MJMPC_BCT                                              \
                     "v_fmac_" SFX_ "_dpp %1, %13, %19" MJMPC_BC "26 row_mask:0xf bank_mask:0xf"                            \
                     : "+v"(up), "+v"(dn)                                                                                   \
                     : "v"(sb[0]), "v"(sb[1]), "v"(sb[2]), "v"(sb[3]), "v"(sb[4]), "v"(sb[5]), "v"(fb[0]), "v"(fb[1]),     \
                       "v"(fb[2]), "v"(fb[3]), "v"(fb[4]), "v"(fb[5]), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]),        \
                       "v"(s[4]), "v"(s[5]), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "n"(K));   \
    }
MJMPC_DPP_GROUPS(double, "f64")
MJMPC_DPP_GROUPS(float, "f32")
#undef MJMPC_DPP_GROUPS

// a_j += (lane K's a_j) * m for j in [J0, J1): groups of four, then singles
template <int K, int J0, int J1, typename T>
__device__ __forceinline__ void fma_bcast_self_range(T* a, T m) {
    if constexpr (J1 - J0 >= 4) {
        fma_bcast_self4<K>(a[J0], a[J0 + 1], a[J0 + 2], a[J0 + 3], m);
        fma_bcast_self_range<K, J0 + 4, J1>(a, m);
    } else if constexpr (J1 - J0 >= 1) {
        fma_bcast<K>(a[J0], a[J0], m);
        fma_bcast_self_range<K, J0 + 1, J1>(a, m);
    }
}

struct Topo {       // my link's place in the tree (registers)
    int parent, subsize, jumps;
    int seg_end;    // one past the last link of my kinematic tree (links are numbered tree by tree, depth-first)
    int anc[5];     // my ancestor at distance 2^k (pointer jumping), -1 beyond the root; only ever indexed by an
                    // unrolled loop counter, so that it stays in registers
    unsigned ancmask;
};

// x[c] <- sum over my path to the root (myself included) of x[c]: pointer jumping
template <int J, int NC, int NCOL, typename T>
__device__ __forceinline__ void path_sum_bcast(T* acc, const T* x, unsigned ancmask) {
    const T take = ((ancmask >> J) & 1u) ? T(1) : T(0);         // lane J is me or one of my ancestors
    static_assert(NC == 6, "six components per path sum (a spatial vector)");
    fma_bcast_vec6<J>(acc, x, take);
    if constexpr (J + 1 < NCOL) path_sum_bcast<J + 1, NC, NCOL>(acc, x, ancmask);
}
template <int NC, int DP, int PL, int NCOL = 16, typename T>
__device__ __forceinline__ void path_sum(T* x, const Topo& tp, T* X, int l) {
    if constexpr (PL == 16) {       // one DPP row: every ancestor's value by broadcast, no LDS round per jump
        T acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = T(0);
        path_sum_bcast<0, NC, NCOL>(acc, x, tp.ancmask);
#pragma unroll
        for (int c = 0; c < NC; ++c) x[c] = acc[c];
        return;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (k >= tp.jumps) break;
#pragma unroll
        for (int c = 0; c < NC; ++c) X[c * PL + l] = x[c];
        TSYNC();
        const int a = tp.anc[k];
        if (a >= 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) x[c] += X[c * PL + a];
        }
        TSYNC();
    }
}

// x[c] <- sum over my subtree (myself included) of x[c]; subtree = links [l, l + subsize): doubling tables in two
// LDS buffers of NC x 32, the blocks of sizes 2^k that tile the range are added as the tables appear
template <int NC, int PL, int NL = 32, typename T>
__device__ __forceinline__ void subtree_sum(T* x, const Topo& tp, T* X, int l) {
    static_assert(NC <= 6, "two buffers of NC x PL must fit the 12 x PL exchange area");
    if constexpr (TREE_DPP_SUBTREE) {
        // The suffix sums S_l = x_l + ... + x_(end of my TREE) are zero-filling DPP row shifts (a 16-lane particle is one
        // row; a 32-lane one adds the second row's total to the first: its lane 0, by a row swap and a broadcast), and
        // the subtree [l, l + n) is S_l - S_{l+n}, one ds_bpermute per dword - no LDS round trip.  The scan is SEGMENTED at
        // the roots (tp.seg_end): a light object listed before a heavy manipulator would otherwise lose its inertia in
        // the difference.  Inside one tree the difference still cancels: relative error eps * (what follows me in my tree
        // / my subtree), a few tens on the models here - against the exact doubling tables below (TREE_DPP_SUBTREE = 0).
        const int end = l + tp.subsize;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // (round 5: shifts beyond the instantiation's link count NL move nothing - compiled out: a chain of two links takes
            // one of the four steps.  As wave-uniform run-time branches on the model's link count they cost the cheetah 1.5 %)
            T s = x[c];
            T t = dpp_zero<0x101>(s);
            s += l + 1 < tp.seg_end ? t : T(0);
            if constexpr (NL > 2) {
                t = dpp_zero<0x102>(s);
                s += l + 2 < tp.seg_end ? t : T(0);
            }
            if constexpr (NL > 4) {
                t = dpp_zero<0x104>(s);
                s += l + 4 < tp.seg_end ? t : T(0);
            }
            if constexpr (NL > 8) {
                t = dpp_zero<0x108>(s);
                s += l + 8 < tp.seg_end ? t : T(0);
            }
            if constexpr (PL == 32) {
                const T other = bcast_row<0>(swap_rows(s));     // lane 0 of the particle's OTHER row, in every lane of mine
                s += (l < 16 && tp.seg_end > 16) ? other : T(0);
            }
            const T tail = __shfl(s, end & (PL - 1), PL);
            x[c] = s - (end < tp.seg_end ? tail : T(0));
        }
        return;
    }
    constexpr int LOG = PL == 32 ? 5 : 4;
    T cur[NC], acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { cur[c] = x[c]; acc[c] = T(0); }
    int pos = l;
#pragma unroll
    for (int k = 0; k <= LOG; ++k) {
        T* buf = X + (k & 1) * NC * PL;
#pragma unroll
        for (int c = 0; c < NC; ++c) buf[c * PL + l] = cur[c];      // T_k[l]
        TSYNC();
        if ((tp.subsize >> k) & 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += buf[c * PL + pos];
            pos += 1 << k;
        }
        if (k < LOG) {
            const int nb = l + (1 << k);
            if (nb < PL) {
#pragma unroll
                for (int c = 0; c < NC; ++c) cur[c] += buf[c * PL + nb];   // T_{k+1}[l] = T_k[l] + T_k[l + 2^k]
            }
        }
    }
    TSYNC();
#pragma unroll
    for (int c = 0; c < NC; ++c) x[c] = acc[c];
}

// first tangent of the contact frame: mju_makeFrame's rule - the hint (capsule axis) made orthogonal to the normal, or,
// without a usable hint, the y (else z) axis
template <typename T>
__device__ __forceinline__ void frame_tangent(const T* nrm, const T* hint, T* t1) {
    T ax3[3] = {hint[0], hint[1], hint[2]};
    if (dot3(ax3, ax3) < T(0.25)) {
        const bool yy = nrm[1] < T(0.5) && nrm[1] > T(-0.5);
        ax3[0] = T(0);
        ax3[1] = yy ? T(1) : T(0);
        ax3[2] = yy ? T(0) : T(1);
    }
    const T pr = dot3(nrm, ax3);
    for (int k = 0; k < 3; ++k) t1[k] = ax3[k] - pr * nrm[k];
    const T nn = dot3(t1, t1);
    if (nn < T(1e-30)) { t1[0] = T(1); t1[1] = T(0); t1[2] = T(0); }
    else { const T inv = rsqrt_(nn); for (int k = 0; k < 3; ++k) t1[k] *= inv; }
}

// MuJoCo mj_makeImpedance + mj_referenceConstraint for one scalar row (r = pos - margin); constants from LDS
// sol = {K, B, dmin, dmax, width, mid, power}: one of the model's sets (T_SOLTAB)
template <typename T>
__device__ __forceinline__ void tree_row_params(const T* sol, T r, T diag_approx, T jv, T& D, T& aref) {
    const T dmin = sol[2], dmax = sol[3], width = sol[4], mid = sol[5];
    const int n = (int)sol[6];
    T x = fabs(r) * rcp_(width), y;
    x = x > T(1) ? T(1) : x;
    const bool lo = x <= mid;
    const T xa = lo ? x : T(1) - x, ma = lo ? mid : T(1) - mid;
    T num = xa, den = T(1);
    for (int k = 1; k < n; ++k) {           // integer power (the model compiler rejects anything else)
        num *= xa;
        den *= ma;
    }
    const T qq = num * rcp_(den);
    y = n == 1 ? x : (lo ? qq : T(1) - qq);
    const T imp = dmin + y * (dmax - dmin);
    T Rr = (T(1) - imp) * rcp_(imp) * diag_approx;
    Rr = Rr < T(1e-15) ? T(1e-15) : Rr;
    D = rcp_(Rr);
    aref = -sol[1] * jv - sol[0] * imp * r;
}

// ---- the TRUNK of the elimination tree --------------------------------------------------------------------------
// Links 0 .. kt-1 that form a chain from the root, every one with a single child (the arm under a hand; the object's
// chain above a manipulator): they are eliminated LAST, one per round, and in the rounds scheme each of those rounds is
// an LDS round trip with one busy lane.  Their lanes sit in one DPP row, so the trunk is done in registers instead: the
// rows go to ABSOLUTE column order (through LDS, once), a dense leaves-first L'DL with DPP broadcasts follows, and the
// triangular solves take the trunk's levels as broadcast FMAs.  kt is the same for every particle of a launch.
template <int DP>
struct Trunk { static constexpr int KT = DP < 16 ? DP : 16; };
// pivot reciprocals of the in-register factorisations (-DTREE_PIVOT_RCP_FAST: developer A/B switch, one Newton step on v_rcp_f64)
#ifdef TREE_PIVOT_RCP_FAST
#define PIVOT_RCP rcp_fast
#else
#define PIVOT_RCP rcp_
#endif

// ra[col] = A[l][col] (col <= l < kt, absolute columns) in, the factor out: ra[l] = D_l, ra[col] = L[l][col]
template <int K, int KT, typename T>
struct TrunkStep {
    static __device__ __forceinline__ void run(T* ra, int l, int kt) {
        if (K < kt) {
            asm volatile("" : "+v"(l));     // (lane masks recomputed here, not kept - and spilled - across the substep)
            const T invd = PIVOT_RCP(bcast_row<K>(ra[K]));
            T fj = T(0);                        // A[K][l] / D_K for lanes l < K: lane K's entry of MY column
#pragma unroll
            for (int col = 0; col < K; ++col) {
                const T v = bcast_row<K>(ra[col]);
                fj = (l == col) ? v : fj;
            }
            fj *= invd;
            const T nf = -fj;
            fma_bcast_self_range<K, 0, K>(ra, nf);             // A[l][col] -= f_l A[K][col]  (lanes >= K: f = 0)
#pragma unroll
            for (int col = 0; col < K; ++col) ra[col] = (l == K) ? ra[col] * invd : ra[col];  // row K itself becomes L[K][.]
        }
        if constexpr (K > 1) TrunkStep<K - 1, KT, T>::run(ra, l, kt);
    }
};

// trunk lanes: path-indexed registers -> absolute columns, through the published rows
template <int DP, int PL, typename T>
__device__ __forceinline__ void trunk_load(T* ra, const T* ROW, int l, int kt) {
    constexpr int KT = Trunk<DP>::KT;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int col = 0; col < KT; ++col) ra[col] = (l < kt && col <= l) ? ROW[l * row_stride(DP) + (l - col)] : T(0);
}
template <int DP, int PL, typename T>
__device__ __forceinline__ void trunk_store(const T* ra, T* ROW, int l, int kt) {
    constexpr int KT = Trunk<DP>::KT;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int col = 0; col < KT; ++col)
        if (l < kt && col <= l) ROW[l * row_stride(DP) + (l - col)] = ra[col];
}

template <int K, int KT, typename T>
struct TrunkFwd {       // L' w = b on the trunk, leaves first: b_l -= L[K][l] b_K, K descending
    static __device__ __forceinline__ void run(const T* cK, T& b, int kt) {
        if (K < kt) fma_bcast<K>(b, b, -cK[K]);
        if constexpr (K > 1) TrunkFwd<K - 1, KT, T>::run(cK, b, kt);
    }
};
template <int A, int KT, typename T>
struct TrunkBwd {       // L x = u on the trunk, root first: x_l -= L[l][a] x_a, a ascending
    static __device__ __forceinline__ void run(const T* La, T& b, int kt) {
        if (A + 1 < kt) fma_bcast<A>(b, b, -La[A]);
        if constexpr (A + 2 < KT) TrunkBwd<A + 1, KT, T>::run(La, b, kt);
    }
};

// The updates of the trunk rows by ALL the links below the trunk (their Schur complement), taken at once when those rows
// are final: S[j][c] = sum over links k below the trunk of L[k][j] D_k L[k][c] (j, c trunk links, c <= j).  Link k reads
// its own entries for the trunk links back from its published row (trunk link c is depth_k - c links above it), forms
// the products in registers, and ONE lane sum per (j, c) gives every lane the total; trunk lane j subtracts S[j][.] from
// its row (absolute column order, see trunk_load).  In the rounds these were pulls by the trunk lanes - the arm under a
// hand pulled one row per finger per round, and every round waited for it.
constexpr int TRUNK_SCHUR_MAX = 6;
template <int J, int C, int KT, int PL, typename T>
struct TrunkSchur {
    static __device__ __forceinline__ void run(T* ra, const T* tk, T dk, int l, int kt) {
        if (J < kt) {
            const T s = sum_lanes<PL>(tk[J] * dk * tk[C]);
            ra[C] = (l == J) ? ra[C] - s : ra[C];
        }
        if constexpr (C < J) TrunkSchur<J, C + 1, KT, PL, T>::run(ra, tk, dk, l, kt);
        else if constexpr (J + 1 < KT && J + 1 < TRUNK_SCHUR_MAX) TrunkSchur<J + 1, 0, KT, PL, T>::run(ra, tk, dk, l, kt);
    }
};
template <int DP, int PL, typename T>
__device__ __forceinline__ void trunk_schur(T* ra, const T* ROW, int l, int depth, int kt) {
    constexpr int KT = Trunk<DP>::KT;
    asm volatile("" : "+v"(l));
    T tk[KT];
    const bool below = l >= kt;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        const int d = depth - c;                                    // trunk link c is d links above me
        tk[c] = (below && c < kt && d >= 1 && d < DP) ? ROW[l * row_stride(DP) + d] : T(0);
    }
    const T dk = below ? ROW[l * row_stride(DP)] : T(0);
    TrunkSchur<0, 0, KT, PL, T>::run(ra, tk, dk, l, kt);
}

// Tree-sparse L'DL, in place: in  r[c] = A[l][ancestor at distance c]  (c < DP, zero beyond the root),
// out r[0] = D_l, r[c] = L[l][ancestor at distance c] (c >= 1).  One round per height: every lane publishes its row,
// then pulls the rows of its descendants of that height (elimination list ELIM[e * 32 + l], sorted by height:
// k | dist << 8 | height << 16, -1 ends it):  r[c] -= (r_k[dist] / r_k[0]) r_k[dist + c].  Rows are DP + 1 long (slot DP
// carries 1 / D_k); dist + c may run past a row's own path (dist + c > depth of k) and even past the row, into the next
// lane's - published, finite - data: what is read there only ever lands in entries of r past MY path (c > my depth),
// which nothing consumes (they are published with the row and read again only into such entries).
template <int DP, int PL, typename T>
__device__ __forceinline__ void tree_factor(T* r, const int* ELIM, T* ROW, int l, int n_rounds, int kt, int depth) {
    constexpr int KT = Trunk<DP>::KT;
    int e = 0, ent = ELIM[l];
    const int lds_rounds = n_rounds - kt;                   // heights below the trunk's lowest link (kt = 1: all but the root's)
    // short trunks (an arm: k (k + 1) / 2 lane sums) take the updates from below as ONE Schur complement after the rounds
    // (trunk_schur); long ones (an object's six joints above an arm: 55 sums) keep pulling rows in the rounds
    const bool schur = DP <= 8 && kt >= 2 && kt <= TRUNK_SCHUR_MAX;     // (compiled for the short-path instantiations only)
    const bool trunk = kt >= 2 && l < kt;
    const bool defer = schur && trunk;
    for (int hgt = 0; hgt < lds_rounds; ++hgt) {
#pragma unroll
        for (int c = 0; c < DP; ++c) ROW[l * row_stride(DP) + c] = r[c];
        ROW[l * row_stride(DP) + DP] = rcp_(r[0]);                 // 1 / D of a row that is final; read by its ancestors
        TSYNC();
        while (__any(!defer && ent >= 0 && (ent >> 16) == hgt)) {
            if (!defer && ent >= 0 && (ent >> 16) == hgt) {
                const T* rk = ROW + (ent & 255) * row_stride(DP);
                const int a = (ent >> 8) & 255;
                const T f = rk[a] * rk[DP];
#pragma unroll
                for (int c = 0; c < DP; ++c) r[c] -= f * rk[a + c];
                ++e;
                ent = e < PL - 1 ? ELIM[e * PL + l] : -1;
            }
        }
        TSYNC();
    }
    const T invd = rcp_(r[0]);
    const bool below = !trunk;                              // my row is final: scale it (trunk rows: by the trunk steps)
#pragma unroll
    for (int c = 1; c < DP; ++c) r[c] = below ? r[c] * invd : r[c];
#pragma unroll
    for (int c = 0; c < DP; ++c) ROW[l * row_stride(DP) + c] = r[c];       // the solves read L from here
    TSYNC();
    if (kt >= 2) {
        T ra[KT];
        trunk_load<DP, PL>(ra, ROW, l, kt);
        if constexpr (DP <= 8) { if (schur) trunk_schur<DP, PL>(ra, ROW, l, depth, kt); }
        TrunkStep<KT - 1, KT, T>::run(ra, l, kt);
        TSYNC();
        trunk_store<DP, PL>(ra, ROW, l, kt);
        TSYNC();
        if (trunk) {
#pragma unroll
            for (int c = 0; c < DP; ++c) r[c] = ROW[l * row_stride(DP) + c];
        }
    }
}

// The same for TWO matrices of the tree's pattern in one pass over the rounds (H = M + J'DJ of the first Newton iteration
// and the Euler matrix M + hB): one set of LDS round trips and list walks instead of two.
template <int DP, int PL, typename T>
__device__ __forceinline__ void tree_factor2(T* r, T* q, const int* ELIM, T* ROW, T* ROW2, int l, int n_rounds, int kt, int depth) {
    constexpr int KT = Trunk<DP>::KT;
    int e = 0, ent = ELIM[l];
    const int lds_rounds = n_rounds - kt;
    const bool schur = DP <= 8 && kt >= 2 && kt <= TRUNK_SCHUR_MAX;     // (compiled for the short-path instantiations only)
    const bool trunk = kt >= 2 && l < kt;
    const bool defer = schur && trunk;
    for (int hgt = 0; hgt < lds_rounds; ++hgt) {
#pragma unroll
        for (int c = 0; c < DP; ++c) { ROW[l * row_stride(DP) + c] = r[c]; ROW2[l * row_stride(DP) + c] = q[c]; }
        ROW[l * row_stride(DP) + DP] = rcp_(r[0]);
        ROW2[l * row_stride(DP) + DP] = rcp_(q[0]);
        TSYNC();
        while (__any(!defer && ent >= 0 && (ent >> 16) == hgt)) {
            if (!defer && ent >= 0 && (ent >> 16) == hgt) {
                const T* rk = ROW + (ent & 255) * row_stride(DP);
                const T* qk = ROW2 + (ent & 255) * row_stride(DP);
                const int a = (ent >> 8) & 255;
                const T f = rk[a] * rk[DP], g = qk[a] * qk[DP];
#pragma unroll
                for (int c = 0; c < DP; ++c) { r[c] -= f * rk[a + c]; q[c] -= g * qk[a + c]; }
                ++e;
                ent = e < PL - 1 ? ELIM[e * PL + l] : -1;
            }
        }
        TSYNC();
    }
    const T invd = rcp_(r[0]), invq = rcp_(q[0]);
    const bool below = !trunk;
#pragma unroll
    for (int c = 1; c < DP; ++c) { r[c] = below ? r[c] * invd : r[c]; q[c] = below ? q[c] * invq : q[c]; }
#pragma unroll
    for (int c = 0; c < DP; ++c) { ROW[l * row_stride(DP) + c] = r[c]; ROW2[l * row_stride(DP) + c] = q[c]; }
    TSYNC();
    if (kt >= 2) {
        T ra[KT];
        trunk_load<DP, PL>(ra, ROW, l, kt);
        if constexpr (DP <= 8) { if (schur) trunk_schur<DP, PL>(ra, ROW, l, depth, kt); }
        TrunkStep<KT - 1, KT, T>::run(ra, l, kt);
        TSYNC();
        trunk_store<DP, PL>(ra, ROW, l, kt);
        trunk_load<DP, PL>(ra, ROW2, l, kt);
        if constexpr (DP <= 8) { if (schur) trunk_schur<DP, PL>(ra, ROW2, l, depth, kt); }
        TrunkStep<KT - 1, KT, T>::run(ra, l, kt);
        TSYNC();
        trunk_store<DP, PL>(ra, ROW2, l, kt);
        TSYNC();
        if (trunk) {
#pragma unroll
            for (int c = 0; c < DP; ++c) { r[c] = ROW[l * row_stride(DP) + c]; q[c] = ROW2[l * row_stride(DP) + c]; }
        }
    }
}

// x <- (L' D L)^-1 b, one entry per lane; ROW holds the factor (tree_factor), AT[c * 32 + l] = my ancestor at distance c
template <int DP, int PL, typename T>
__device__ __forceinline__ T tree_solve(const T* r, T b, const int* ELIM, const int* AT, const T* ROW, T* VEC, int l,
                                        int n_rounds, int depth, int max_depth, int kt) {
    constexpr int KT = Trunk<DP>::KT;
    // L' w = b, leaves first:  w_i = b_i - sum over descendants k of L[k][i] w_k
    int e = 0, ent = ELIM[l];
    const int lds_rounds = n_rounds - kt;
    const bool trunk = kt >= 2 && l < kt;       // trunk lanes take everything below the trunk AFTER the rounds, in one go
    for (int hgt = 0; hgt < lds_rounds; ++hgt) {
        VEC[l] = b;
        TSYNC();
        while (__any(!trunk && ent >= 0 && (ent >> 16) == hgt)) {
            if (!trunk && ent >= 0 && (ent >> 16) == hgt) {
                const int k = ent & 255, a = (ent >> 8) & 255;
                b -= ROW[k * row_stride(DP) + a] * VEC[k];
                ++e;
                ent = e < PL - 1 ? ELIM[e * PL + l] : -1;
            }
        }
        TSYNC();
    }
    if (kt >= 2) {
        // w_k of every link below the trunk is final.  Short trunks on short paths: trunk lane j needs sum_k L[k][j] w_k over
        // all the links below - every such link reads its entries for the trunk links back from its own row and ONE lane
        // sum per trunk link hands the total over (as trunk_schur does for the factorisation).  Otherwise the trunk lanes
        // pull, but independently of each other now: four list entries per trip, eight loads in flight.
        constexpr int KS = DP <= 8 ? TRUNK_SCHUR_MAX : 12;     // trunk links the lane-sum path of the SOLVE takes (one sum each)
        if (DP <= 16 && kt <= KS) {
            if constexpr (DP <= 16) {
                const bool below = l >= kt;
                T tk[KS];
#pragma unroll
                for (int c = 0; c < KS; ++c) {
                    const int d = depth - c;
                    tk[c] = (below && c < kt && d >= 1 && d < DP) ? ROW[l * row_stride(DP) + d] : T(0);
                }
                const T wk = below ? b : T(0);
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    if (j < kt) {
                        const T sj = sum_lanes<PL>(tk[j] * wk);
                        b = (l == j) ? b - sj : b;
                    }
                }
            }
        } else {
        VEC[l] = b;
        TSYNC();
        for (int e0 = 0; e0 < PL - 1; e0 += 4) {
            int en[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                en[u] = (trunk && e0 + u < PL - 1) ? ELIM[(e0 + u) * PL + l] : -1;
                ok[u] = en[u] >= 0 && (en[u] >> 16) < lds_rounds;
            }
            if (!__any(ok[0])) break;           // (the lists are sorted by height: nothing below the trunk is left)
            T lv[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = ok[u] ? en[u] & 255 : 0, a = ok[u] ? (en[u] >> 8) & 255 : 0;
                lv[u] = ROW[k * row_stride(DP) + a];
                wv[u] = VEC[k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) b -= ok[u] ? lv[u] * wv[u] : T(0);
        }
        TSYNC();
        }
        // the trunk's rounds: my column of its factor from the published rows, then broadcast FMAs
        asm volatile("" : "+v"(l));
        T cK[KT];
#pragma unroll
        for (int K = 1; K < KT; ++K) cK[K] = (K < kt && l < K) ? ROW[K * row_stride(DP) + (K - l)] : T(0);
        cK[0] = T(0);
        TrunkFwd<KT - 1, KT, T>::run(cK, b, kt);
    }
    b *= rcp_(r[0]);
    // L x = u, root first:  x_i = u_i - sum over ancestors L[i][anc] x_anc; level dl finalises the links of depth dl
    int dl0 = 0;
    if (kt >= 2) {
        T La[KT];
#pragma unroll
        for (int a = 0; a < KT; ++a) La[a] = (l < kt && a < l) ? ROW[l * row_stride(DP) + (l - a)] : T(0);
        TrunkBwd<0, KT, T>::run(La, b, kt);
        // everybody below the trunk takes all its trunk ancestors (depth 0 .. kt-1: lanes 0 .. kt-1) in one round
        VEC[l] = b;
        TSYNC();
        if (l >= kt) {
            for (int dl = 0; dl < kt; ++dl) {
                if (depth > dl) b -= ROW[l * row_stride(DP) + (depth - dl)] * VEC[dl];
            }
        }
        TSYNC();
        dl0 = kt;
    }
    for (int dl = dl0; dl + 1 < max_depth; ++dl) {
        VEC[l] = b;
        TSYNC();
        if (depth > dl) {
            const int c = depth - dl;
            b -= ROW[l * row_stride(DP) + c] * VEC[AT[c * PL + l]];
        }
        TSYNC();
    }
    return b;
}

// my dense row of the mass matrix: md[j] = S_j . F_l (j an ancestor of l, or l), S_l . F_j (j in l's subtree), else 0
template <int J, int DN, typename T>
__device__ __forceinline__ void dense_mass_row(T* md, const T* S, const T* F, int l, unsigned ancmask, int subsize, bool dof) {
    T up = T(0), dn = T(0);
    fma_bcast_dots6<J>(up, dn, S, F);           // (lane J's S) . (my F),  (lane J's F) . (my S)
    const bool anc = (ancmask >> J) & 1u, sub = J > l && J < l + subsize;
    md[J] = dof ? (anc ? up : (sub ? dn : T(0))) : T(0);
    if constexpr (J + 1 < DN) dense_mass_row<J + 1, DN>(md, S, F, l, ancmask, subsize, dof);
}

// r[j] += wn (lane j's jn) + w1 (lane j's j1) + w2 (lane j's j2): the contribution of one contact point to my dense row
template <int J, int DN, bool FRIC, typename T>
__device__ __forceinline__ void dense_contact(T* r, T jn, T j1, T j2, T wn, T w1, T w2) {
    if constexpr (FRIC) fma_bcast_3<J>(r[J], jn, j1, j2, wn, w1, w2);
    else fma_bcast<J>(r[J], jn, wn);
    if constexpr (J + 1 < DN) dense_contact<J + 1, DN, FRIC>(r, jn, j1, j2, wn, w1, w2);
}

// in: r[j] = H[l][j].  out: r[k] = L[l][k] for k < l, r[j] = D_l L[j][l] for j > l (what the backward solve wants),
// dinv = 1 / D_l.  Lanes past the matrix (l >= DN, or unit rows of spare lanes) ride along untouched.
template <int K, int DN, typename T>
struct DenseStep {
    static __device__ __forceinline__ void run(T* r, T& dinv, int l) {
        const T inv = PIVOT_RCP(bcast_row<K>(r[K]));
        dinv = l == K ? inv : dinv;
        const T lik = l > K ? r[K] * inv : T(0), nlik = -lik;
        fma_bcast_self_range<K, K + 1, DN>(r, nlik);        // r[j] -= lik * (lane K's r[j]), j > K
        r[K] = l > K ? lik : r[K];
        if constexpr (K + 1 < DN) DenseStep<K + 1, DN, T>::run(r, dinv, l);
    }
};
template <int DN, typename T>
__device__ __forceinline__ void dense_factor(T* r, T& dinv, int l) {
    asm volatile("" : "+v"(l));     // (the lane masks below are recomputed here, not kept - and spilled - across the substep)
    dinv = T(1);
    DenseStep<0, DN, T>::run(r, dinv, l);
}

// x <- (L D L')^-1 b, one entry per lane
template <int K, int DN, typename T>
struct DenseFwd {
    static __device__ __forceinline__ void run(const T* r, T& x, int l) {
        fma_bcast<K>(x, x, l > K ? -r[K] : T(0));       // x -= L[l][K] * (lane K's x, final by now)
        if constexpr (K + 2 < DN) DenseFwd<K + 1, DN, T>::run(r, x, l);
    }
};
template <int J, int DN, typename T>
struct DenseBwd {
    static __device__ __forceinline__ void run(const T* r, T dinv, T z, T& acc, int l) {
        const T xj = z - dinv * acc;                    // final in lane J (its acc is complete)
        fma_bcast<J>(acc, xj, l < J ? r[J] : T(0));
        if constexpr (J > 1) DenseBwd<J - 1, DN, T>::run(r, dinv, z, acc, l);
    }
};
template <int DN, typename T>
__device__ __forceinline__ T dense_solve(const T* r, T dinv, T b, int l) {
    asm volatile("" : "+v"(l));
    T x = b;
    DenseFwd<0, DN, T>::run(r, x, l);
    const T z = x * dinv;                               // (backward: x_i = z_i - dinv_i sum_{j > i} r_i[j] x_j)
    T acc = T(0);
    DenseBwd<DN - 1, DN, T>::run(r, dinv, z, acc, l);
    return z - dinv * acc;
}

// ---- dense factorisation for 32-lane particles (DN = 32): the particle's two 16-lane DPP rows hold dofs 0..15 (the EVEN
// row) and 16..31 (the ODD row); v_permlane16_swap hands every lane the value its partner lane of the other row holds.
// Lane l keeps row l of the matrix: even-row lanes columns 0..15 only (A11), odd-row lanes all 32 (A21 | A22).
// Right-looking L D L' - see Dense32Step for how a step gets by with one row exchange.  At the end the odd row's
// L[j][0..15] are transposed through LDS into the even-row lanes' entries 16..31 (as D_i L[j][i], what the backward solve
// reads).  Measured on the pen-in-hand model (30 dofs, elimination paths of 16 links), cycles per factorisation / solve:
// tree-sparse 20.4 k / 10.6 k, this 8.6 k / 3.2 k.
__device__ __forceinline__ void row_pair(float x, float& even, float& odd) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    even = __uint_as_float(r[0]);
    odd = __uint_as_float(r[1]);
}
__device__ __forceinline__ void row_pair(double x, double& even, double& odd) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    even = __hiloint2double((int)b[0], (int)a[0]);
    odd = __hiloint2double((int)b[1], (int)a[1]);
}
template <typename T>
__device__ __forceinline__ T row_even(T x) { T e, o; row_pair(x, e, o); return e; }
template <typename T>
__device__ __forceinline__ T row_odd(T x) { T e, o; row_pair(x, e, o); return o; }

// r[J0 + c] += (lane (J0 + c) % 16's x) * m for c < N: up to four broadcast FMAs per asm statement (one set of wait states)
#define MJMPC_COLS(T_, SFX_)                                                                                                 \
    template <int L0, int L1, int L2, int L3>                                                                               \
    __device__ __forceinline__ void fma_bcast_cols4(T_& a0, T_& a1, T_& a2, T_& a3, T_ x, T_ m) {                          \
        asm volatile("s_nop 1\n\t"                                                                                        \
                     "v_fmac_" SFX_ "_dpp %0, %4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"                      \
                     "v_fmac_" SFX_ "_dpp %1, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"                      \
                     "v_fmac_" SFX_ "_dpp %2, %4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"                      \
                     "v_fmac_" SFX_ "_dpp %3, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"                            \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(m), "n"(L0), "n"(L1), "n"(L2), "n"(L3));     \
    }
MJMPC_COLS(double, "f64")
MJMPC_COLS(float, "f32")
#undef MJMPC_COLS
// columns [J0, J1) of a step: r[j] += (lane j % 16 of my row's x) * m
template <int J0, int J1, typename T>
__device__ __forceinline__ void dense32_cols(T* r, T x, T m) {
    if constexpr (J1 - J0 >= 4) {
        fma_bcast_cols4<J0 % 16, (J0 + 1) % 16, (J0 + 2) % 16, (J0 + 3) % 16>(r[J0], r[J0 + 1], r[J0 + 2], r[J0 + 3], x, m);
        dense32_cols<J0 + 4, J1>(r, x, m);
    } else if constexpr (J1 - J0 >= 1) {
        fma_bcast<J0 % 16>(r[J0], x, m);
        dense32_cols<J0 + 1, J1>(r, x, m);
    }
}
// the reciprocal of pivot K (the diagonal entry of lane K), in every lane that will use it
template <int K, typename T>
__device__ __forceinline__ T dense32_pivot_inv(const T* r) {
    if constexpr (K < 16) return PIVOT_RCP(bcast_row<K>(row_even(r[K])));
    else return PIVOT_RCP(bcast_row<K - 16>(r[K]));          // (only the odd-row lanes use it: theirs is the right one)
}
// Step K.  A[K][j] = A[j][K] (the matrix is symmetric and lane j > K has not scaled its entry K yet), so the pivot row is
// never sent anywhere: every lane broadcasts-reads "entry K of lane j" - from its own row for columns of its own half,
// through ONE row exchange of entry K per step for the even-row columns the odd-row lanes need.
template <int K, typename T>
struct Dense32Step {
    static __device__ __forceinline__ void run(T* r, T& dinv, int l, T inv) {
        dinv = l == K ? inv : dinv;
        const T rk = r[K];
        const T lik = l > K ? rk * inv : T(0), nlik = -lik;
        T inv_next = T(0);
        if constexpr (K < 16) {
            const T rke = row_even(rk);                     // entry K of the even-row partner (even-row lanes: my own)
            const T nlik_odd = l >= 16 ? nlik : T(0);      // (even-row lanes keep no columns 16 .. 31)
            if constexpr (K + 1 < 16) {
                dense32_cols<K + 1, K + 2>(r, rke, nlik);
                inv_next = dense32_pivot_inv<K + 1>(r);     // (the next pivot is final: its reciprocal overlaps the rest of the step)
                dense32_cols<K + 2, 16>(r, rke, nlik);
                dense32_cols<16, 32>(r, rk, nlik_odd);
            } else {
                dense32_cols<16, 17>(r, rk, nlik_odd);
                inv_next = dense32_pivot_inv<16>(r);
                dense32_cols<17, 32>(r, rk, nlik_odd);
            }
        } else if constexpr (K + 1 < 32) {
            const T nlik_odd = l >= 16 ? nlik : T(0);
            dense32_cols<K + 1, K + 2>(r, rk, nlik_odd);
            inv_next = dense32_pivot_inv<K + 1>(r);
            dense32_cols<K + 2, 32>(r, rk, nlik_odd);
        }
        r[K] = l > K ? lik : rk;
        if constexpr (K + 1 < 32) Dense32Step<K + 1, T>::run(r, dinv, l, inv_next);
    }
};
// TR: 16 x 17 scalars of LDS (the row area, idle in this instantiation)
template <typename T>
__device__ __forceinline__ void dense32_factor(T* r, T& dinv, int l, T* TR) {
    asm volatile("" : "+v"(l));
    dinv = T(1);
    Dense32Step<0, T>::run(r, dinv, l, dense32_pivot_inv<0>(r));
    if (l >= 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) TR[(l - 16) * 17 + k] = r[k];
    }
    TSYNC();
    if (l < 16) {
        const T d = rcp_(dinv);
#pragma unroll
        for (int j = 0; j < 16; ++j) r[16 + j] = TR[j * 17 + l] * d;
    }
    TSYNC();
}
// L y = b in three legs: the even row's triangle (odd-row lanes ride along with zero multipliers), the odd-row lanes'
// 16 products with the even row's y (one row exchange), the odd row's triangle; L' x = z the other way round.
template <int K, int K1, typename T>
__device__ __forceinline__ void dense32_fwd_tri(const T* r, T& x, int l, bool mine) {
    if constexpr (K + 1 < K1) {
        fma_bcast<K % 16>(x, x, (mine && l > K) ? -r[K] : T(0));
        dense32_fwd_tri<K + 1, K1>(r, x, l, mine);
    }
}
template <int J, int J0, typename T>
__device__ __forceinline__ void dense32_bwd_tri(const T* r, T dinv, T z, T& acc, int l, bool mine) {
    if constexpr (J > J0) {
        const T xj = z - dinv * acc;                    // final in lane J
        fma_bcast<J % 16>(acc, xj, (mine && l < J) ? r[J] : T(0));
        dense32_bwd_tri<J - 1, J0>(r, dinv, z, acc, l, mine);
    }
}
template <int C, typename T>
__device__ __forceinline__ void dense32_cross(const T* r, T& acc, T xo, T sgn, bool mine) {       // acc += sgn sum_c r[C0 + c] (lane c's xo)
    if constexpr (C < 16) {
        fma_bcast<C>(acc, xo, mine ? sgn * r[C] : T(0));
        dense32_cross<C + 1>(r, acc, xo, sgn, mine);
    }
}
template <typename T>
__device__ __forceinline__ T dense32_solve(const T* r, T dinv, T b, int l) {
    asm volatile("" : "+v"(l));
    const bool even = l < 16;
    T x = b;
    dense32_fwd_tri<0, 16>(r, x, l, even);                      // y_0 .. y_15
    dense32_cross<0>(r, x, row_even(x), T(-1), !even);          // odd-row lanes: x -= sum_{K < 16} L[l][K] y_K
    dense32_fwd_tri<16, 32>(r, x, l, !even);
    const T z = x * dinv;
    T acc = T(0);
    dense32_bwd_tri<31, 16>(r, dinv, z, acc, l, !even);         // x_31 .. x_17 (x_16: its sum is complete)
    dense32_cross<0>(r + 16, acc, row_odd(z - dinv * acc), T(1), even);     // even-row lanes: acc += sum_{J >= 16} (D_l L[J][l]) x_J
    dense32_bwd_tri<15, 0>(r, dinv, z, acc, l, even);
    return z - dinv * acc;
}
// my dense row of the mass matrix (see dense_mass_row): Se / Fe = the even-row partner's S and F, So / Fo the odd-row one's
template <int J, typename T>
__device__ __forceinline__ void dense32_mass_row(T* md, const T* Se, const T* Fe, const T* So, const T* Fo, const T* S, const T* F,
                                                 int l, unsigned ancmask, int subsize, bool dof) {
    T up = T(0), dn = T(0);
    if constexpr (J < 16) fma_bcast_dots6x<J>(up, dn, Se, Fe, S, F);
    else fma_bcast_dots6x<J - 16>(up, dn, So, Fo, S, F);
    const bool anc = (ancmask >> J) & 1u, sub = J > l && J < l + subsize;
    md[J] = dof ? (anc ? up : (sub ? dn : T(0))) : T(0);
    if constexpr (J + 1 < 32) dense32_mass_row<J + 1>(md, Se, Fe, So, Fo, S, F, l, ancmask, subsize, dof);
}
// r[j] += wn (lane j's jn) + w1 (lane j's j1) + w2 (lane j's j2) over the 32 lanes; je / jo = the Jacobians' entries as
// the even-row / odd-row partner holds them
template <int J, bool FRIC, typename T>
__device__ __forceinline__ void dense32_contact(T* r, const T* je, const T* jo, T wn, T w1, T w2) {
    if constexpr (J < 16) {
        if constexpr (FRIC) fma_bcast_3<J>(r[J], je[0], je[1], je[2], wn, w1, w2);
        else fma_bcast<J>(r[J], je[0], wn);
    } else {
        if constexpr (FRIC) fma_bcast_3<J - 16>(r[J], jo[0], jo[1], jo[2], wn, w1, w2);
        else fma_bcast<J - 16>(r[J], jo[0], wn);
    }
    if constexpr (J + 1 < 32) dense32_contact<J + 1, FRIC>(r, je, jo, wn, w1, w2);
}
// one name for both dense schemes
template <int DN, typename T>
__device__ __forceinline__ void dense_factor_any(T* r, T& dinv, int l, T* TR) {
    if constexpr (DN == 32) dense32_factor(r, dinv, l, TR);
    else dense_factor<DN>(r, dinv, l);
}
template <int DN, typename T>
__device__ __forceinline__ T dense_solve_any(const T* r, T dinv, T b, int l) {
    if constexpr (DN == 32) return dense32_solve(r, dinv, b, l);
    else return dense_solve<DN>(r, dinv, b, l);
}

// round 5's colliders (GEN >= 2): functions of their own or inlined into the record walk (developer A/B: -DTREE_GEOM_INLINE)
#ifdef TREE_GEOM_INLINE
#define TREE_GEOM_FN __forceinline__
#else
#define TREE_GEOM_FN __noinline__
#endif
// The surface point of a solid box (half sizes h, its own frame) nearest to `loc` (that frame): outside - the clamped point,
// normal towards `loc`; inside - the nearest face and its outward normal; len = signed distance (mjc_SphereBox's geometry;
// the oracle's box_point).  false: `loc` lies on the surface to rounding (no normal).
template <typename T>
__device__ __forceinline__ bool box_point(const T* h, const T* loc, T* cl, T* nb, T& len) {
    bool inside = true;
    for (int i = 0; i < 3; ++i) {
        cl[i] = fmin(fmax(loc[i], -h[i]), h[i]);
        inside = inside && cl[i] == loc[i];
    }
    nb[0] = T(0); nb[1] = T(0); nb[2] = T(0);
    if (inside) {
        int kk = 0;
        T best = h[0] - fabs(loc[0]);
        for (int i = 1; i < 3; ++i) {
            const T gap = h[i] - fabs(loc[i]);
            if (gap < best) { best = gap; kk = i; }
        }
        const T sg = loc[kk] >= T(0) ? T(1) : T(-1);
        for (int i = 0; i < 3; ++i) {
            if (i == kk) { cl[i] = sg * h[i]; nb[i] = sg; }
        }
        len = -best;
        return true;
    }
    for (int i = 0; i < 3; ++i) nb[i] = loc[i] - cl[i];
    len = sqrt_(dot3(nb, nb));
    if (!(len > T(1e-14))) return false;
    const T inv = T(1) / len;
    for (int i = 0; i < 3; ++i) nb[i] *= inv;
    return true;
}

// The parameter t in [0, 1] at which the segment a + t b (box frame) comes nearest to the solid box: the squared distance is
// convex and piecewise quadratic, its pieces end where a coordinate crosses a face plane (at most six break points); every
// piece is minimised in closed form and the least minimum taken, the first where pieces tie (the oracle's seg_box_param)
// The surface point of a solid cylinder (axis from p0 along d, end to end; radius r) nearest to a point c: outside - the point of
// the solid nearest to c, normal from it to c; inside - the nearest of the side and the two caps.  len = signed distance of c
// from the surface; false when c lies on it to rounding.  (The oracle's cyl_point, operation for operation.)
template <typename T>
__device__ TREE_GEOM_FN bool cyl_point(const T* p0, const T* d, T r, const T* c, T* q, T* n, T& len) {
    const T L = sqrt_(dot3(d, d));
    T u[3], w[3], rv[3], rh[3];
    for (int i = 0; i < 3; ++i) { u[i] = d[i] / L; w[i] = c[i] - p0[i]; }
    const T z = dot3(w, u);
    for (int i = 0; i < 3; ++i) rv[i] = w[i] - z * u[i];
    const T rho = sqrt_(dot3(rv, rv));
    if (rho > T(1e-14)) {
        for (int i = 0; i < 3; ++i) rh[i] = rv[i] / rho;
    } else {                                // on the axis: a fixed direction across it
        const bool xx = u[0] < T(0.9) && u[0] > T(-0.9);
        const T e[3] = {xx ? T(1) : T(0), xx ? T(0) : T(1), T(0)};
        T pr = dot3(e, u);
        for (int i = 0; i < 3; ++i) rh[i] = e[i] - pr * u[i];
        pr = sqrt_(dot3(rh, rh));
        for (int i = 0; i < 3; ++i) rh[i] /= pr;
    }
    if (!(z > T(0) && z < L && rho < r)) {
        const T zc = z < T(0) ? T(0) : (z > L ? L : z), rc = rho < r ? rho : r;
        T diff[3];
        for (int i = 0; i < 3; ++i) { q[i] = p0[i] + zc * u[i] + rc * rh[i]; diff[i] = c[i] - q[i]; }
        len = sqrt_(dot3(diff, diff));
        if (len < T(1e-14)) return false;
        for (int i = 0; i < 3; ++i) n[i] = diff[i] / len;
        return true;
    }
    const T ds = r - rho, db = z, dt = L - z;
    if (ds <= db && ds <= dt) {
        for (int i = 0; i < 3; ++i) { n[i] = rh[i]; q[i] = p0[i] + z * u[i] + r * rh[i]; }
        len = -ds;
    } else if (db <= dt) {
        for (int i = 0; i < 3; ++i) { n[i] = -u[i]; q[i] = p0[i] + rho * rh[i]; }
        len = -db;
    } else {
        for (int i = 0; i < 3; ++i) { n[i] = u[i]; q[i] = p0[i] + L * u[i] + rho * rh[i]; }
        len = -dt;
    }
    return true;
}

template <typename T>
__device__ TREE_GEOM_FN T seg_box_param(const T* h, const T* a, const T* b) {
    T bp[8];
    bp[0] = T(0);
    bp[7] = T(1);
    for (int i = 0; i < 3; ++i)
        for (int sg = 0; sg < 2; ++sg) {
            T t = T(1);             // (a coordinate that does not move has no crossing: a duplicate of the end point)
            if (b[i] != T(0)) t = ((sg ? h[i] : -h[i]) - a[i]) / b[i];
            bp[1 + 2 * i + sg] = fmin(fmax(t, T(0)), T(1));
        }
    for (int i = 1; i < 8; ++i)
        for (int j = i; j > 0 && bp[j] < bp[j - 1]; --j) { const T t = bp[j]; bp[j] = bp[j - 1]; bp[j - 1] = t; }
    T best_f = T(1e300), best_t = T(0), zlo = T(2), zhi = T(-1);
    if (sizeof(T) == 4) best_f = T(3e38);
    for (int k = 0; k < 7; ++k) {
        const T u = bp[k], w = bp[k + 1], mid = T(0.5) * (u + w);
        T B = T(0), C = T(0), off[3];
        int act[3];
        for (int i = 0; i < 3; ++i) {
            const T si = a[i] + mid * b[i];
            act[i] = si > h[i] ? 1 : (si < -h[i] ? -1 : 0);
            off[i] = a[i] - T(act[i]) * h[i];
            if (act[i]) { B += b[i] * b[i]; C += b[i] * off[i]; }
        }
        T tc = u;
        if (B > T(0)) tc = fmin(fmax(-C / B, u), w);
        T f = T(0);
        for (int i = 0; i < 3; ++i)
            if (act[i]) { const T e = off[i] + tc * b[i]; f += e * e; }
        if (f < best_f) { best_f = f; best_t = tc; }
        if (!act[0] && !act[1] && !act[2]) {        // (no coordinate outside its faces on this piece: the axis runs INSIDE the box here)
            zlo = fmin(zlo, u);
            zhi = fmax(zhi, w);
        }
    }
    // an axis that enters the box has distance 0 on a whole stretch: a point well inside it, three eighths of the way (see the oracle)
    if (zhi >= zlo) return zlo + T(0.375) * (zhi - zlo);
    return best_t;
}

// closest points of two segments p1 + s d1, p2 + t d2 (the clamped closed form of the geom-geom records; the oracle's seg_seg)
template <typename T>
__device__ __forceinline__ void seg_seg_params(const T* o1, const T* d1, const T* o2, const T* d2, T& ss, T& tt) {
    const T r3[3] = {o1[0] - o2[0], o1[1] - o2[1], o1[2] - o2[2]};
    const T a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r3), EPS = T(1e-18);
    auto clamp01 = [](T x) { return x < T(0) ? T(0) : (x > T(1) ? T(1) : x); };
    if (a <= EPS && e <= EPS) { ss = T(0); tt = T(0); }
    else if (a <= EPS) { ss = T(0); tt = clamp01(f / e); }
    else {
        const T c = dot3(d1, r3);
        if (e <= EPS) { tt = T(0); ss = clamp01(-c / a); }
        else {
            const T b = dot3(d1, d2), denom = a * e - b * b;
            ss = denom > T(1e-12) * a * e ? clamp01((b * f - c * e) / denom) : T(0);
            tt = (b * ss + f) / e;
            if (tt < T(0)) { tt = T(0); ss = clamp01(-c / a); }
            else if (tt > T(1)) { tt = T(1); ss = clamp01((b - c) / a); }
        }
    }
}

// Two boxes (round 5): contact `want` of the up to four the oracle's box_box makes - the same separating-axis test (edge axes
// must beat the best face axis by 5 %), the same Sutherland-Hodgman clipping of the incident face against the reference
// face's side planes, the same edge-pair closest points, in the same order of operations.  R0 / R1: row-major, their COLUMNS
// are the boxes' axes in the world.  n: unit normal from box 1 to box 0.  false: no such contact.
template <typename T>
__device__ TREE_GEOM_FN bool box_box_contact(const T* c0, const T* R0, const T* h0, const T* c1, const T* R1, const T* h1, T margin,
                                             int want, T* n, T* pos, T& dist) {
    T A[3][3], B[3][3], d[3];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) { A[i][k] = R0[3 * k + i]; B[i][k] = R1[3 * k + i]; }
    for (int k = 0; k < 3; ++k) d[k] = c1[k] - c0[k];
    T best = T(sizeof(T) == 4 ? -3e38 : -1e300), bestL[3] = {T(0), T(0), T(1)};
    int code = -1;
    for (int t = 0; t < 6; ++t) {
        const T* L = t < 3 ? A[t] : B[t - 3];
        T ra = T(0), rb = T(0);
        for (int i = 0; i < 3; ++i) { ra += h0[i] * fabs(dot3(L, A[i])); rb += h1[i] * fabs(dot3(L, B[i])); }
        const T sep = fabs(dot3(L, d)) - ra - rb;
        if (sep > best) { best = sep; code = t; bestL[0] = L[0]; bestL[1] = L[1]; bestL[2] = L[2]; }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T L[3];
            cross3(A[i], B[j], L);
            const T ln = sqrt_(dot3(L, L));
            if (ln < T(1e-9)) continue;
            for (int k = 0; k < 3; ++k) L[k] /= ln;
            T ra = T(0), rb = T(0);
            for (int k = 0; k < 3; ++k) { ra += h0[k] * fabs(dot3(L, A[k])); rb += h1[k] * fabs(dot3(L, B[k])); }
            const T sep = fabs(dot3(L, d)) - ra - rb;
            if (sep > best + T(0.05) * fabs(best) + T(1e-12)) { best = sep; code = 6 + 3 * i + j; bestL[0] = L[0]; bestL[1] = L[1]; bestL[2] = L[2]; }
        }
    if (!(best < margin)) return false;
    const T sgn = dot3(bestL, d) < T(0) ? T(-1) : T(1);
    T L[3];
    for (int k = 0; k < 3; ++k) { L[k] = sgn * bestL[k]; n[k] = -L[k]; }
    if (code >= 6) {
        if (want != 0) return false;
        const int i = (code - 6) / 3, j = (code - 6) % 3;
        T pa[3], pb[3];
        for (int k = 0; k < 3; ++k) { pa[k] = c0[k]; pb[k] = c1[k]; }
        for (int a = 0; a < 3; ++a) {
            if (a != i) { const T sg = dot3(L, A[a]) > T(0) ? T(1) : T(-1); for (int k = 0; k < 3; ++k) pa[k] += sg * h0[a] * A[a][k]; }
            if (a != j) { const T sg = dot3(L, B[a]) > T(0) ? T(-1) : T(1); for (int k = 0; k < 3; ++k) pb[k] += sg * h1[a] * B[a][k]; }
        }
        T sa[3], da[3], sb[3], db[3], ta, tb;
        for (int k = 0; k < 3; ++k) {
            sa[k] = pa[k] - h0[i] * A[i][k]; da[k] = T(2) * h0[i] * A[i][k];
            sb[k] = pb[k] - h1[j] * B[j][k]; db[k] = T(2) * h1[j] * B[j][k];
        }
        seg_seg_params(sa, da, sb, db, ta, tb);
        T dd = T(0);
        for (int k = 0; k < 3; ++k) {
            const T qa = sa[k] + ta * da[k], qb = sb[k] + tb * db[k];
            dd += (qb - qa) * L[k];
            pos[k] = T(0.5) * (qa + qb);
        }
        dist = dd;
        return dd < margin;
    }
    const bool ref1 = code >= 3;
    const int fi = code % 3;
    const T (*RA)[3] = ref1 ? B : A;
    const T (*IA)[3] = ref1 ? A : B;
    const T *hr = ref1 ? h1 : h0, *hi = ref1 ? h0 : h1, *cr = ref1 ? c1 : c0, *ci = ref1 ? c0 : c1;
    T nr[3];
    for (int k = 0; k < 3; ++k) nr[k] = ref1 ? -L[k] : L[k];
    int ii = 0;
    T most = T(-1);
    for (int a = 0; a < 3; ++a) { const T c = fabs(dot3(nr, IA[a])); if (c > most) { most = c; ii = a; } }
    const T si = dot3(nr, IA[ii]) > T(0) ? T(-1) : T(1);
    const int i1 = (ii + 1) % 3, i2 = (ii + 2) % 3;
    T poly[8][3], tmp[8][3];
    int np_ = 4;
    for (int c = 0; c < 4; ++c) {
        const T s1 = (c == 0 || c == 3) ? T(1) : T(-1), s2 = c < 2 ? T(1) : T(-1);
        for (int k = 0; k < 3; ++k) poly[c][k] = ci[k] + si * hi[ii] * IA[ii][k] + s1 * hi[i1] * IA[i1][k] + s2 * hi[i2] * IA[i2][k];
    }
    const int r1 = (fi + 1) % 3, r2 = (fi + 2) % 3;
    for (int side = 0; side < 4 && np_ > 0; ++side) {
        const T* u = RA[side < 2 ? r1 : r2];
        const T sg = (side & 1) ? T(-1) : T(1), hu = hr[side < 2 ? r1 : r2];
        int nt = 0;
        for (int c = 0; c < np_; ++c) {
            const T* P = poly[c];
            const T* Q = poly[(c + 1) % np_];
            T dp = T(0), dq = T(0);
            for (int k = 0; k < 3; ++k) { dp += sg * u[k] * (P[k] - cr[k]); dq += sg * u[k] * (Q[k] - cr[k]); }
            dp -= hu; dq -= hu;
            if (dp <= T(0) && nt < 8) { for (int k = 0; k < 3; ++k) tmp[nt][k] = P[k]; ++nt; }
            if (((dp < T(0) && dq > T(0)) || (dp > T(0) && dq < T(0))) && nt < 8) {
                const T t = dp / (dp - dq);
                for (int k = 0; k < 3; ++k) tmp[nt][k] = P[k] + t * (Q[k] - P[k]);
                ++nt;
            }
        }
        np_ = nt;
        for (int c = 0; c < nt; ++c)
            for (int k = 0; k < 3; ++k) poly[c][k] = tmp[c][k];
    }
    // the clipped corners within the margin of the reference face, then corner want * nk / 4 of them (the first four as they are)
    int nk = 0;
    T kd[8];
    for (int c = 0; c < np_; ++c) {
        T dd = -hr[fi];
        for (int k = 0; k < 3; ++k) dd += nr[k] * (poly[c][k] - cr[k]);
        if (dd < margin) {
            for (int k = 0; k < 3; ++k) tmp[nk][k] = poly[c][k];
            kd[nk++] = dd;
        }
    }
    const int nout = nk < 4 ? nk : 4;
    if (want >= nout) return false;
    const int src = nk <= 4 ? want : (want * nk) / 4;
    dist = kd[src];
    for (int k = 0; k < 3; ++k) pos[k] = tmp[src][k] - T(0.5) * kd[src] * nr[k];
    return true;
}

// Round 5's contact-record kinds, evaluated by the record's lane (GEN = 2 instantiations only): link poses by value (A: the
// record's link, B: link [13] or the world), the record and its extension.
template <typename T>
struct LinkPose { T R[9], p[3]; };
template <typename T>
struct ExtraContact { T cp[3], n[3], t1[3], dist; bool hit; };
template <typename T>
__device__ __forceinline__ ExtraContact<T> extra_geometry(int kind, LinkPose<T> fa, LinkPose<T> fb, const T* sp, const T* ex, T pnx, T pny, T pnz, T plane_d) {
    ExtraContact<T> out;
    const T pn[3] = {pnx, pny, pnz}, zero3[3] = {T(0), T(0), T(0)};
    const T* Rl = fa.R;
    const T* Rb = fb.R;
    T tv[3], oA[3], oB[3];
    mv3(Rl, sp + 1, tv);
    for (int k = 0; k < 3; ++k) oA[k] = fa.p[k] + tv[k];
    out.hit = false;
    out.dist = T(0);
    for (int k = 0; k < 3; ++k) { out.cp[k] = T(0); out.n[k] = pn[k]; out.t1[k] = T(0); }
    if (kind == PT_PLANE_CYL) {
        // a cylinder on the plane (mjc_PlaneCylinder, as the oracle restates it): the lowest point of the lower cap's rim, the
        // point below it on the other cap's rim, two more points of the lower rim 120 degrees to either side - candidate
        // sp[22] of the four; the later ones count only if the first does
        T a3[3], vec[3];
        mv3(Rl, sp + 8, a3);
        T pa = dot3(pn, a3);
        if (pa > T(0)) { for (int k = 0; k < 3; ++k) a3[k] = -a3[k]; pa = -pa; }
        for (int k = 0; k < 3; ++k) vec[k] = pn[k] - pa * a3[k];
        T len = sqrt_(dot3(vec, vec));
        if (len < T(1e-12)) {
            const bool yy = a3[1] < T(0.5) && a3[1] > T(-0.5);
            const T ax3[3] = {T(0), yy ? T(1) : T(0), yy ? T(0) : T(1)};
            const T pr = dot3(a3, ax3);
            for (int k = 0; k < 3; ++k) vec[k] = ax3[k] - pr * a3[k];
            len = sqrt_(dot3(vec, vec));
        }
        const T rr = sp[4], hh = sp[14], sc0 = rr / len;
        for (int k = 0; k < 3; ++k) vec[k] *= sc0;
        const T d0 = dot3(pn, oA) - plane_d, pv = dot3(pn, vec), d1 = d0 + pa * hh - pv;
        const int kk = (int)sp[22];
        T pt3[3], cdist;
        if (kk == 0) {
            for (int k = 0; k < 3; ++k) pt3[k] = oA[k] + a3[k] * hh - vec[k];
            cdist = d1;
        } else if (kk == 1) {
            for (int k = 0; k < 3; ++k) pt3[k] = oA[k] - a3[k] * hh - vec[k];
            cdist = d0 - pa * hh - pv;
        } else {
            T v1[3];
            cross3(vec, a3, v1);
            const T sc = (kk == 2 ? T(1) : T(-1)) * T(0.86602540378443864676);
            for (int k = 0; k < 3; ++k) pt3[k] = oA[k] + a3[k] * hh + T(0.5) * vec[k] + sc * v1[k];
            cdist = d0 + pa * hh + T(0.5) * pv;
        }
        for (int k = 0; k < 3; ++k) out.cp[k] = pt3[k] - pn[k] * (T(0.5) * cdist);
        frame_tangent(pn, zero3, out.t1);
        out.dist = cdist;
        out.hit = d1 < sp[5] && cdist < sp[5];
        return out;
    }
    mv3(Rb, sp + 14, tv);
    for (int k = 0; k < 3; ++k) oB[k] = fb.p[k] + tv[k];
    if (kind == PT_BOX_BOX) {
        // two boxes: contact sp[22] of up to four (box_box_contact)
        T RwA[9], RwB[9], Q[9];
        {
            const T w_ = ex[12], x_ = ex[13], y_ = ex[14], z_ = ex[15];
            Q[0] = T(1) - T(2) * (y_ * y_ + z_ * z_); Q[1] = T(2) * (x_ * y_ - w_ * z_); Q[2] = T(2) * (x_ * z_ + w_ * y_);
            Q[3] = T(2) * (x_ * y_ + w_ * z_); Q[4] = T(1) - T(2) * (x_ * x_ + z_ * z_); Q[5] = T(2) * (y_ * z_ - w_ * x_);
            Q[6] = T(2) * (x_ * z_ - w_ * y_); Q[7] = T(2) * (y_ * z_ + w_ * x_); Q[8] = T(1) - T(2) * (x_ * x_ + y_ * y_);
        }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                RwA[3 * i + j] = Rl[3 * i] * ex[3 + j] + Rl[3 * i + 1] * ex[6 + j] + Rl[3 * i + 2] * ex[9 + j];
                RwB[3 * i + j] = Rb[3 * i] * Q[j] + Rb[3 * i + 1] * Q[3 + j] + Rb[3 * i + 2] * Q[6 + j];
            }
        T nv3[3] = {T(0), T(0), T(1)}, cp3[3] = {T(0), T(0), T(0)}, cdist = T(0);
        const bool ok = box_box_contact(oA, RwA, ex, oB, RwB, ex + 16, sp[5], (int)sp[22], nv3, cp3, cdist);
        for (int k = 0; k < 3; ++k) { out.n[k] = nv3[k]; out.cp[k] = cp3[k]; }
        frame_tangent(nv3, zero3, out.t1);
        out.dist = cdist;
        out.hit = ok && cdist < sp[5];
        return out;
    }
    if (kind == PT_SEG_CYL || kind == PT_CYL_SEG) {
        // a sphere or a capsule against a cylinder (a scheme of its own, see the oracle): candidate sp[22] of three - the point
        // of the capsule's axis nearest to the cylinder's axis, its two ends where they are not that point - against the
        // cylinder's nearest surface point (cyl_point)
        const bool cylA = kind == PT_CYL_SEG;
        T dA[3], dB[3];
        mv3(Rl, sp + 8, dA);
        mv3(Rb, sp + 18, dB);
        const T* os0 = cylA ? oB : oA;
        const T* dv = cylA ? dB : dA;
        const T* oc = cylA ? oA : oB;
        const T* dc = cylA ? dA : dB;
        const T rs = cylA ? sp[17] : sp[4], rc = cylA ? sp[4] : sp[17];
        T ts = T(0), tc = T(0);
        if (dot3(dv, dv) > T(0)) seg_seg_params(os0, dv, oc, dc, ts, tc);
        const int cand = (int)sp[22];
        const T tt = cand == 0 ? ts : (cand == 1 ? T(0) : T(1));
        bool ok = cand == 0 || tt != ts;
        T ps[3], q[3], nn[3], len = T(0);
        for (int k = 0; k < 3; ++k) ps[k] = os0[k] + tt * dv[k];
        ok = cyl_point(oc, dc, rc, ps, q, nn, len) && ok;
        const T sgn = cylA ? T(-1) : T(1);          // nn points from the cylinder to the sphere; the contact's normal from geom B to geom A
        const T cdist = len - rs;
        for (int k = 0; k < 3; ++k) {
            out.n[k] = sgn * nn[k];
            const T onB = cylA ? ps[k] + out.n[k] * rs : q[k];
            out.cp[k] = onB + out.n[k] * (T(0.5) * cdist);
        }
        frame_tangent(out.n, zero3, out.t1);
        out.dist = cdist;
        out.hit = ok && cdist < sp[5];
        return out;
    }
    // a capsule against a box (a scheme of its own, see the oracle): candidate sp[22] of three - where the capsule's axis comes
    // nearest to the box (a point of its stretch inside, should it pass through), its two ends where they are not that point
    const bool boxA = kind == PT_BOX_CAPSULE;
    const T* Rx = boxA ? Rl : Rb;                       // the box's link
    const T* ob = boxA ? oA : oB;
    const T* os0 = boxA ? oB : oA;
    const T rs = boxA ? sp[17] : sp[4];
    T Rw[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rw[3 * i + j] = Rx[3 * i] * ex[3 + j] + Rx[3 * i + 1] * ex[6 + j] + Rx[3 * i + 2] * ex[9 + j];
    const T rel[3] = {os0[0] - ob[0], os0[1] - ob[1], os0[2] - ob[2]};
    T loc[3], cl[3], os[3] = {os0[0], os0[1], os0[2]}, dv[3], bl[3];
    for (int i = 0; i < 3; ++i) loc[i] = Rw[i] * rel[0] + Rw[3 + i] * rel[1] + Rw[6 + i] * rel[2];
    if (boxA) mv3(Rb, sp + 18, dv); else mv3(Rl, sp + 8, dv);
    for (int i = 0; i < 3; ++i) bl[i] = Rw[i] * dv[0] + Rw[3 + i] * dv[1] + Rw[6 + i] * dv[2];
    const T tstar = seg_box_param(ex, loc, bl);
    const int cand = (int)sp[22];
    const T tt = cand == 0 ? tstar : (cand == 1 ? T(0) : T(1));
    bool ok = cand == 0 || tt != tstar;
    for (int i = 0; i < 3; ++i) { loc[i] += tt * bl[i]; os[i] += tt * dv[i]; }
    T nl[3], nb[3], len = T(0);
    ok = box_point(ex, loc, cl, nl, len) && ok;
    mv3(Rw, nl, nb);
    T cb[3];
    mv3(Rw, cl, cb);
    for (int k = 0; k < 3; ++k) cb[k] += ob[k];
    // nb points from the box to the capsule; the contact's normal from geom B to geom A
    const T sgn = boxA ? T(-1) : T(1);
    const T cdist = len - rs;
    for (int k = 0; k < 3; ++k) {
        out.n[k] = sgn * nb[k];
        const T onB = boxA ? os[k] + out.n[k] * rs : cb[k];      // geom B's surface point; the contact midway along the normal
        out.cp[k] = onB + out.n[k] * (T(0.5) * cdist);
    }
    frame_tangent(out.n, zero3, out.t1);
    out.dist = cdist;
    out.hit = ok && cdist < sp[5];
    return out;
}

// Exact line search of the constraint solver's safeguard (see the Newton loop): the root in [0, 1] of the increasing,
// piecewise linear  phi'(al) = g0 + al dg + sum over the particle's rows of D_r min(0, r_r + al dr_r) dr_r  - every lane
// brings its limit row (Dl, rl, drl) and, as the owner of a contact point, that point's rows (Dc, rb[], drb[]).  Bisection
// to 2^-24, then the secant between the last bracket (exact when no row switches inside it).  A function of its own,
// not inlined: it runs in a few particle-substeps per million and must not cost the kernel's hot path registers.
// General instantiation (GEN): a point's rows may carry their own D (Dk: a connect equality) and be BILATERAL (bil: cost
// 1/2 D r^2 on both sides), and every lane brings its friction-loss row (Df, bound ff, residual rf + al drf: the slope of
// the Huber cost is D r clamped to +-ff).
template <int PL, int NR, int GEN, typename T>
__device__ __noinline__ T exact_line_search(T g0, T dg, T Dl, T rl, T drl, T Dc, T rb0, T rb1, T rb2, T rb3, T drb0, T drb1, T drb2, T drb3,
                                            T Dk0, T Dk1, T Dk2, T Dk3, bool bil, T Df, T ff, T rf, T drf) {
    // (the rows arrive by value, as in cone_line_search below: arrays behind pointers would keep the caller's copies in scratch)
    const T rb[4] = {rb0, rb1, rb2, rb3}, drb[4] = {drb0, drb1, drb2, drb3}, Dk[4] = {Dk0, Dk1, Dk2, Dk3};
    auto phi = [&](T al) -> T {
        const T r = rl + al * drl;
        T tsum = Dl * (r < T(0) ? r : T(0)) * drl;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const T rr = rb[k] + al * drb[k];
            if constexpr (GEN) tsum += Dk[k] * ((bil || rr < T(0)) ? rr : T(0)) * drb[k];
            else tsum += Dc * (rr < T(0) ? rr : T(0)) * drb[k];
        }
        if constexpr (GEN) {
            const T sl = Df * (rf + al * drf);
            tsum += fmin(fmax(sl, -ff), ff) * drf;
        }
        // (the first lane's value for every lane of the particle: the bisection's decisions must not differ between lanes -
        // see cone_line_search, where they were seen to)
        return __shfl(g0 + al * dg + sum_lanes<PL>(tsum), 0, PL);
    };
    T fhi = phi(T(1));
    if (!(fhi > T(0))) return T(1);
    T lo = T(0), hi = T(1), flo = phi(T(0));
    if (!(flo < T(0))) return T(0);             // (numerically not a descent direction: the gradient at the base point is zero to rounding - stay)
#ifdef TREE_LS_BISECT           // developer A/B: rounds 3 - 5's root finder (24 bisections, then the secant of the last bracket)
    for (int b = 0; b < 24; ++b) {
        const T mid = T(0.5) * (lo + hi), fm = phi(mid);
        if (fm > T(0)) { hi = mid; fhi = fm; } else { lo = mid; flo = fm; }
    }
    const T den = fhi - flo;
    return den > T(0) ? lo - flo * (hi - lo) * rcp_(den) : lo;
#else
    // Round 6: false position with the Illinois rule (the end that has not moved twice running gives up half its value) - on a
    // piecewise LINEAR phi' the secant is the root as soon as the bracket lies on the root's piece, i.e. after a few evaluations
    // instead of 25; a bisection step whenever the secant leaves the bracket.  Particles of a wavefront leave the loop together.
    const T tol = T(sizeof(T) == 4 ? 1e-6 : 1e-14) * (fabs(flo) + fabs(fhi));
    T al = T(0);
    int side = 0;
    bool done = false;
    for (int k = 0; k < 32 && __any(!done); ++k) {
        T m = lo - flo * (hi - lo) * rcp_(fhi - flo);
        if (!(m > lo && m < hi)) m = T(0.5) * (lo + hi);
        const T fm = phi(m);
        if (!done) {
            al = m;
            if (fm > T(0)) { hi = m; fhi = fm; if (side > 0) flo *= T(0.5); side = 1; }
            else { lo = m; flo = fm; if (side < 0) fhi *= T(0.5); side = -1; }
            done = fabs(fm) <= tol || !(hi - lo > T(sizeof(T) == 4 ? 1e-7 : 1e-16));
        }
    }
    return al;
#endif
}

// The same root with ELLIPTIC cones among the particle's records (GEN = 3): phi' is increasing and continuous but no longer
// piecewise linear - the lane that owns an elliptic record (ell) adds grad s(r + al dr) . dr of the cone's cost s (zones and
// formulas: MuJoCo PrimalUpdateConstraint [EXT]; DESIGN 2 derives them).  Newton's iteration on phi' with
// phi'' alongside, kept inside the bracket (a bisection step when it leaves it); on a piecewise linear phi' it lands on the
// root once it is on the root's piece.  D0 = the normal row's D, ir = impratio (tangent rows: D0 ir), fr the friction
// coefficient, mu = fr / sqrt(ir).
template <int PL, int NR, typename T>
__device__ __noinline__ T cone_line_search(T g0, T dg, T Dl, T rl, T drl, T Dc, T rb0, T rb1, T rb2, T rb3, T drb0, T drb1, T drb2, T drb3,
                                           T Dk0, T Dk1, T Dk2, T Dk3, bool bil, T Df, T ff, T rf, T drf, bool ell, T fr, T ir, T mu) {
    // (the rows arrive BY VALUE and the floor flag leaves as a negative return: arrays behind pointers and a flag behind a
    // reference would live in scratch memory on both sides of this call)
    const T rb[4] = {rb0, rb1, rb2, rb3}, drb[4] = {drb0, drb1, drb2, drb3}, Dk[4] = {Dk0, Dk1, Dk2, Dk3};
    if (TREE_SKIP & 64) return T(1);        // (developer timing: the full step, no search)
    bool at_floor;
    // (no lane-dependent branch in here: the lane sums below are DPP exchanges, every lane of the particle must arrive at
    // them together - the cone's and the plain rows' parts are both computed and one selected)
    auto phi = [&](T al, T& curv) -> T {
        const T r = rl + al * drl;
        T tsum = Dl * (r < T(0) ? r : T(0)) * drl, csum = r < T(0) ? Dl * drl * drl : T(0);
        T te, ce;
        {
            const T r0 = rb[0] + al * drb[0], r1 = rb[1] + al * drb[1], r2 = rb[2] + al * drb[2];
            const T N = mu * r0, U1 = fr * r1, U2 = fr * r2, Tt = sqrt_(U1 * U1 + U2 * U2);
            const T Dt = Dc * ir;
            const bool top = N >= mu * Tt || (Tt <= T(0) && N >= T(0));
            const bool bottom = !top && (mu * N + Tt <= T(0) || (Tt <= T(0) && N < T(0)));
            const T tb = Dc * r0 * drb[0] + Dt * (r1 * drb[1] + r2 * drb[2]);
            const T cb = Dc * drb[0] * drb[0] + Dt * (drb[1] * drb[1] + drb[2] * drb[2]);
            const T iT = rcp_(Tt > T(0) ? Tt : T(1)), u1 = U1 * iT, u2 = U2 * iT;
            const T Dm = Dc * rcp_(mu * mu * (T(1) + mu * mu)), NmT = N - mu * Tt;
            // d/dal of (N - mu T): mu dr0 - mu fr (u . dr_t);  d2/dal2 of T: fr^2 (|dr_t|^2 - (u . dr_t)^2) / T
            const T ud = u1 * drb[1] + u2 * drb[2];
            const T dNmT = mu * drb[0] - mu * fr * ud;
            const T tm = Dm * NmT * dNmT;
            const T cm = Dm * dNmT * dNmT - Dm * NmT * mu * fr * fr * (drb[1] * drb[1] + drb[2] * drb[2] - ud * ud) * iT;
            te = top ? T(0) : (bottom ? tb : tm);
            ce = top ? T(0) : (bottom ? cb : cm);
        }
        T tr = T(0), cr = T(0);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const T rr = rb[k] + al * drb[k];
            const T Dr = Dk[k];
            const bool on = bil || rr < T(0);
            tr += Dr * (on ? rr : T(0)) * drb[k];
            cr += on ? Dr * drb[k] * drb[k] : T(0);
        }
        tsum += ell ? te : tr;
        csum += ell ? ce : cr;
        const T sl = Df * (rf + al * drf);
        tsum += fmin(fmax(sl, -ff), ff) * drf;
        csum += (sl > -ff && sl < ff) ? Df * drf * drf : T(0);
        // (the root finder's decisions must be the same in every lane of the particle: each lane evaluates its rows at ITS
        // alpha, and the sums mix them.  Lane sums were seen to differ between lanes here - two groups of lanes settling on
        // two different "roots" - so the first lane's values are the particle's)
        curv = __shfl(dg + sum_lanes<PL>(csum), 0, PL);
        return __shfl(g0 + al * dg + sum_lanes<PL>(tsum), 0, PL);
    };
    T c1, c0;
    T fhi = phi(T(1), c1);
    const T flo = phi(T(0), c0);
    // (full step: phi'(1) <= 0 - or phi'(0) >= 0: in exact arithmetic a Newton direction descends, phi'(0) = -p'Hp; M a - tau
    // at the base point is known through the solves' own equation, i.e. to eps |H| |a|, and once the gradient along p is
    // below that the sign of phi'(0) is noise: the iteration is at the floor of its arithmetic, where the Newton step itself
    // is the best correction there is - staying would stop a few digits short of it)
    bool done = !(fhi > T(0)) || !(flo < T(0));
    at_floor = !(flo < T(0));       // (... and the particle has converged: the caller stops iterating on it)
    T al = T(1);
    T lo = T(0), hi = T(1), f = fhi, fp = c1;
    const T tol = T(sizeof(T) == 4 ? 1e-6 : 1e-14) * (fabs(flo) + fabs(fhi));
    for (int k = 0; k < 40 && __any(!done); ++k) {
        T an = al - f * rcp_(fp);
        if (!(fp > T(0)) || !(an > lo) || !(an < hi)) an = T(0.5) * (lo + hi);
        T fn, cn;
        fn = phi(an, cn);
        if (!done) {
            al = an; f = fn; fp = cn;
            if (fn > T(0)) hi = an; else lo = an;
            done = fabs(fn) <= tol || !(hi - lo > T(sizeof(T) == 4 ? 1e-7 : 1e-16));
        }
    }
    return at_floor ? T(-1) : al;           // (at the floor: the full step, flagged)
}

// waves per SIMD the register allocation aims at: the lean kernels for short paths fit three (f32) / two (f64)
// workgroups' LDS on a CU
constexpr int min_waves(int scalar_bytes, int DP, bool fric, int gen = 0) {
#ifdef TREE_GEN_F32_ONE_WAVE            // developer A/B (round 4): the general f32 kernels spill 512 B per lane at two waves per SIMD; at one
    if (gen) return 1;                  // they gain 3-5 % on 4096-particle launches and lose a third at 32768 (cart-pole 2.11 -> 3.13 ms): off
#endif
    return (DP <= 8 && !fric) ? (scalar_bytes == 4 ? 3 : 2) : ((DP <= 8 && scalar_bytes == 4) ? 2 : 1);
}

// PL = lanes per particle: 32, or 16 for models of up to 16 dofs (four particles per wavefront, a particle = one DPP row)
// DN > 0 (with PL = 16): the dense in-register factorisation of a matrix of up to DN dofs instead of the tree-sparse one
// GEN: the general instantiation (round 4) - ball / free joints (quaternion links), friction-loss rows, sphere / box pairs,
// static geoms, connect / joint equalities, fixed-tendon limits.  GEN = 2 (round 5): also a cylinder on the plane, capsule / box
// and box / box contacts, mjc_PlaneBox's corner rule - instantiations of their own, so that the models of round 4 keep the
// kernels they had (inlined into the one general kernel the new geometry cost the door model 6 %, behind a call 10 %).
// Models that need none of it run GEN = 0, whose code
// is the earlier rounds' to the instruction.
template <typename T, int DP, int NS, bool FRIC, int PL, int DN, int GEN = 0>
__global__ __launch_bounds__(64 * wg_waves(DP, FRIC, sizeof(T), PL), min_waves(sizeof(T), DP, FRIC, GEN)) void tree_rollout_kernel(
    const T* __restrict__ model_all, int model_stride, const double* __restrict__ state, int state_stride, long P, long shard_size, int H,
    int A, const double* __restrict__ mean,
    const T* __restrict__ noise, T* __restrict__ cost, T* __restrict__ act, T* __restrict__ obs, T* __restrict__ nobs,
    unsigned* diag, double* state_out, const double* __restrict__ clw, double* site_out, TreeFusion fuse) {
    constexpr int WG_WAVES = wg_waves(DP, FRIC, sizeof(T), PL);
    constexpr int PPW = 64 / PL;            // particles per wavefront
    constexpr int A_SF = a_sf(PL);
    static_assert(DP <= PL && NS <= PL, "a path / the contact points must fit the lanes of a particle");
    constexpr int NJ = FRIC ? 3 : 1;        // Jacobian rows kept per contact point: normal (+ two tangents)
    constexpr int NR = FRIC ? 4 : 1;        // constraint rows per contact point: Jn (+- mu Jt_k)
    typedef typename std::conditional<FRIC, unsigned long long, unsigned>::type mask_t;    // NR bits per contact point
    static_assert(!GEN || FRIC, "the general instantiation extends the full one");
    constexpr int CS = GEN ? CS_GEN : CS_BASE;
    constexpr int A_VEC = a_vec(DP, PL), A_JC = a_jc(DP, PL), A_CS = a_cs(DP, NS, NJ, PL), A_MISC = a_misc(DP, NS, NJ, PL, CS),
                  A_ROW2 = a_row2(DP, NS, NJ, PL, CS), A_LEN = a_len(DP, NS, NJ, PL, sizeof(T), DN, CS);
    static_assert(DN == 0 || (PL == 16 && DN <= 16 && DP <= DN) || (PL == 32 && DN == 32), "dense rows: one particle = one DPP row, or two (DN = 32)");
    constexpr bool MERGE = merge_factor(DP, FRIC, sizeof(T), PL, DN);
    constexpr int NLINKS = (DN > 0 && DN < PL) ? DN : PL;       // a dense instantiation's models have at most DN links
    constexpr int NBLOB = T_TOPO;           // the constants the loop reads; topology tables are read once, from global memory
    __shared__ __attribute__((aligned(16))) T lds[NBLOB + 1 + PPW * WG_WAVES * A_LEN];
    __shared__ int ELIM[(PL - 1) * PL];     // elimination lists
    __shared__ int AT[DP * PL];             // AT[c * PL + l] = ancestor of link l at distance c IN THE ELIMINATION TREE (-1 beyond the root)
    __shared__ T PEXT[GEN ? TREE_MAX_SPHERES * TREE_PEXT_STRIDE : 1];      // (GEN) what the new record kinds need beyond [24]
    // gridDim.y shards of shard_size consecutive particles (the reference's workers); a workgroup never straddles two.
    // Dynamics randomization (SubprocVecEnv.randomize_dynamics) gives each its own model block (model_stride != 0),
    // a per-worker set_env_state (subproc_vec_env.py:242-251) its own start state (state_stride != 0).
    const T* model = model_all + (long)blockIdx.y * model_stride;
    state += (long)blockIdx.y * state_stride;
    T* M = lds;
    for (int k = threadIdx.x; k < NBLOB; k += blockDim.x) M[k] = model[k];
    for (int k = threadIdx.x; k < (PL - 1) * PL; k += blockDim.x) ELIM[k] = (int)model[T_ELIM + (k / PL) * TL + (k % PL)];
    for (int k = threadIdx.x; k < PPW * WG_WAVES * A_LEN; k += blockDim.x) lds[NBLOB + 1 + k] = T(0);
    if constexpr (GEN)
        for (int k = threadIdx.x; k < TREE_MAX_SPHERES * TREE_PEXT_STRIDE; k += blockDim.x) PEXT[k] = model[T_PEXT + k];
    if (threadIdx.x < PL) {
        int a = threadIdx.x;
        for (int c = 0; c < DP; ++c) {
            AT[c * PL + threadIdx.x] = a;
            a = a >= 0 ? (int)model[T_EPARENT + a] : -1;       // (the ELIMINATION tree: tree_model.h)
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane & (PL - 1), half = lane / PL;        // `half`: which particle of the wavefront (0 .. PPW - 1)
    const long in_shard = ((long)blockIdx.x * WG_WAVES + wave) * PPW + half;
    const long pid = (long)blockIdx.y * shard_size + in_shard;
    const bool live = in_shard < shard_size && pid < P;
    T* X = lds + NBLOB + 1 + (wave * PPW + half) * A_LEN;
    T* ROW = X + A_ROW;
    T* ROW2 = X + A_ROW2;
    T* VEC = X + A_VEC;
    const int n_rounds = __builtin_amdgcn_readfirstlane((int)model[T_N_ROUNDS]), depth = (int)model[T_DEPTH + l];
    int max_depth = 0;
    for (int c = 0; c < DP; ++c) max_depth += __any(AT[c * PL + l] >= 0) ? 1 : 0;     // links on the longest path
    // the trunk of the elimination tree (see TrunkStep): links 0 .. kt-1 chained from the single root, one child each
    int kt = 1;
    if constexpr (DN == 0) {
        const int nv_ = __builtin_amdgcn_readfirstlane((int)model[T_NV]);
        const int ep = (half == 0 && l < nv_) ? (int)model[T_EPARENT + l] : -2;
        if (__ballot(ep == -1) == 1ull) {
            while (kt < Trunk<DP>::KT && kt < nv_ && __ballot(ep == kt - 1) == (1ull << kt)) ++kt;
        }
    }
    // model-wide integers: the same in every lane, kept in scalar registers
    const int nv = __builtin_amdgcn_readfirstlane((int)M[T_NV]), frame_skip = __builtin_amdgcn_readfirstlane((int)M[T_FRAME_SKIP]);
    const int site_link = __builtin_amdgcn_readfirstlane((int)M[T_SITE_LINK]);
    const int n_sphere = __builtin_amdgcn_readfirstlane(min((int)M[T_N_SPHERE], NS));
    const int task = __builtin_amdgcn_readfirstlane((int)M[T_TASK]), obs_skip = __builtin_amdgcn_readfirstlane((int)M[T_OBS_SKIP]);
    const int nq = GEN ? __builtin_amdgcn_readfirstlane((int)model[T_NQ]) : nv;     // entries of MuJoCo's qpos
    const int dobs = task == 1 ? nq + nv - obs_skip : nq + nv + 6;
    const bool slide = FRIC && (int)model[T_JTYPE + l] == 2;      // (slide joints, springs, a medium: the full instantiation only)
    const int act_id = (int)model[T_ACT + l];
    const bool any_tendon_act = FRIC && __any((int)model[T_TPARTNER + l] >= 0 || model[T_TCOEF + l] != T(1));
    const bool fluid = FRIC && (M[T_DENSITY] > T(0) || M[T_VISCOSITY] > T(0));   // (the full instantiation only)

    // my place on the path of every contact point (5 bits each: 1 + the distance from the point's link up to me, 0 if I
    // am not on that path) - a constant of the model; the contact code reads it instead of two LDS tables
    unsigned long long own_path0 = 0;
    unsigned own_path1 = 0;
    {
        const int ns_ = min((int)M[T_N_SPHERE], NS);
        const bool dof_ = l < (int)M[T_NV];
        for (int s = 0; s < ns_; ++s) {
            const T* sp = M + T_SPH + s * TREE_SPH_STRIDE;
            const int idx = (int)sp[11] - depth;
            const int code = (dof_ && idx >= 0 && idx < DP && AT[idx * PL + (int)sp[0]] == l) ? idx + 1 : 0;
            if (s < 12) own_path0 |= (unsigned long long)code << (5 * s);
            else own_path1 |= (unsigned)code << (5 * (s - 12));
        }
    }
    // owner lanes (lane s < n_sphere owns contact point s): the lanes on my point's path, nearest first, five bits each
    // (paths of up to 12 links; longer ones read the ancestor table) - what the point-parallel walks index the broadcast
    // vector with
    unsigned long long pt_path = 0;
    if constexpr (DP <= 12) {
        if (l < min((int)M[T_N_SPHERE], NS)) {
            const T* sp = M + T_SPH + l * TREE_SPH_STRIDE;
            const int link = (int)sp[0], dsl = (int)sp[11];
            for (int c = 0; c < DP; ++c) pt_path |= (unsigned long long)(AT[(c <= dsl ? c : 0) * PL + link] & 31) << (5 * c);
        }
    }
    // (GEN = 3) the records that are contacts under an ELLIPTIC cone (PEXT[21] = impratio > 0): bit s, and bit 4 s + 1 of the
    // row mask - an elliptic record's nibble there is its ZONE (0 top: no force, 1 bottom: three quadratic rows, 2 middle: on
    // the cone), not four row bits
    unsigned ell16 = 0;
    unsigned long long ell_mid = 0;
    if constexpr (GEN >= 3) {
        const int ns_ = min((int)M[T_N_SPHERE], NS);
        for (int s = 0; s < ns_; ++s) {
            const int kind = (int)M[T_SPH + s * TREE_SPH_STRIDE + 12];
            if (kind != PT_CONNECT && kind != PT_WELD && kind != PT_DOFROW && M[T_SPH + s * TREE_SPH_STRIDE + 7] > T(0) &&
                PEXT[s * TREE_PEXT_STRIDE + 21] > T(0)) {
                ell16 |= 1u << s;
                ell_mid |= 2ull << (4 * s);
            }
        }
    }
    Topo tp;
    tp.parent = (int)model[T_PARENT + l];
    tp.subsize = (int)model[T_SUBSIZE + l];
    {
        const int nv_ = (int)model[T_NV];
        const unsigned long long rb = __ballot(half == 0 && l < nv_ && tp.parent < 0);      // the roots (same model for every particle)
        const unsigned later = l + 1 < 32 ? (unsigned)(rb >> (l + 1)) : 0u;
        tp.seg_end = l >= nv_ ? l + 1 : (later ? min(nv_, l + 1 + __builtin_ctz(later)) : nv_);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) tp.anc[k] = (int)model[T_ANC + k * TL + l];      // (kinematic ancestors: pointer jumping)
    tp.jumps = __builtin_amdgcn_readfirstlane((int)M[T_JUMPS]);
    tp.ancmask = (unsigned)model[T_ANCMASK + l] | ((unsigned)model[T_ANCMASK + TL + l] << 16);
    const bool dof = l < nv;
    // (GEN) my link's kind: a ball joint is three links - the first (BALL_X) holds the quaternion (q = x, qy, qz, qw) and
    // turns the frame, the other two ride along (identity transform; their axes are the body's own y and z) - a free
    // joint three slides along the world axes and a ball.  qadr / qoff: my coordinate's place in MuJoCo's qpos.
    const int lkind = GEN ? (int)model[T_JTYPE + l] : 1;
    const int ball_g = (GEN && dof && lkind >= LINK_BALL_X) ? lkind - LINK_BALL_X : -1;
    const bool has_ball = GEN && __builtin_amdgcn_readfirstlane((int)model[T_HAS_BALL]) != 0;
    const int qadr = GEN ? (int)model[T_QADR + l] : l;
    const T qoff = GEN ? model[T_QOFF + l] : T(0);
    // (BALL_X links) the joint's qpos0 quaternion q0: MuJoCo's qpos = q0 * (the link's quaternion, relative to the qpos0 pose)
    const T q0w = (GEN && ball_g == 0) ? model[T_QW0 + l] : T(1);
    const T q0y = (GEN && ball_g == 0) ? model[T_QOFF + l + 1] : T(0), q0z = (GEN && ball_g == 0) ? model[T_QOFF + l + 2] : T(0);
    const T floss = (GEN && dof) ? model[T_FRICTIONLOSS + l] : T(0);            // dry friction of my dof (0: no row)
    const T jmargin = (GEN && dof) ? model[T_JMARGIN + l] : T(0);               // MJCF joint margin: my limit row exists while dist < margin
    const bool any_floss = GEN && __any(floss > T(0));
    const int dofcls = (int)model[T_DOFCLS + l];                                // my dof's solver sets: limit row | friction-loss row << 3
    // my dof's friction-loss row (J = e_l, position 0): its D is the impedance at 0 - a constant of the model, worked out here once
    // (round 5; it was recomputed every substep from seven registers of solver parameters) - and its reference acceleration -B v
    T Df0 = T(0), fB = T(0);
    if constexpr (GEN) {
        T fsol[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) fsol[k] = model[T_SOLTAB + 7 * (dofcls >> 3) + k];
        T unused;
        tree_row_params(fsol, T(0), model[T_DOF_INVW + l], T(0), Df0, unused);
        Df0 = floss > T(0) ? Df0 : T(0);
        fB = fsol[1];
    }

    T q = dof ? (T)state[l] : T(0), v = dof ? (T)state[TL + l] : T(0);
    T qy = T(0), qz = T(0), qw = T(1);
    if constexpr (GEN) {
        if (ball_g == 0) { qy = (T)state[l + 1]; qz = (T)state[l + 2]; qw = (T)state[TREE_QW + l]; }
        if (ball_g > 0) q = T(0);
    }
    const T tgt[3] = {(T)state[2 * TL], (T)state[2 * TL + 1], (T)state[2 * TL + 2]};
    T sq, cq;
    sincos_(q, sq, cq);
    const T h = M[T_TIMESTEP];
    const T damping = M[T_DAMPING + l], armature = M[T_ARMATURE + l];
    const T pn[3] = {M[T_PLANE_N], M[T_PLANE_N + 1], M[T_PLANE_N + 2]};
    const bool has_u = l < A;
    int lim_mem = 0;                // inst | act << 1 of my limit row in the previous substep
    int fl_mem = 0;                 // (GEN) 1 | (state + 1) << 1 of my friction-loss row in the previous substep
    unsigned cinst_mem = 0;         // contact points of the previous substep ...
    mask_t cact_mem = 0;            // ... and which of their rows were active
    // closed_loop_linear (gym_env_wrapper.py:135-136): the first action needs the site of the fresh observation, which
    // a one-particle launch left in the state vector beforehand (site_out below, mjmpc_tree_rollout_cl)
    T hand_prev[3] = {T(0), T(0), T(0)}, q_prev = q, v_prev = v;
    T xa_prev = T(0);               // the constraint solver's acceleration of the previous substep (its warm start)
    T qy_prev = qy, qz_prev = qz, qw_prev = qw;
    if (clw)
        for (int k = 0; k < 3; ++k) hand_prev[k] = (T)state[2 * TL + 3 + k];
    // MuJoCo's reset on instability (TreeFusion::reset_rec): my particle's lanes as a ballot mask, the bound of mju_isBad
#ifdef MJMPC_NO_RESET         // developer A/B: the reset emulation compiled out (tools/ab_build.py)
    const double* rst = nullptr;
#else
    const double* rst = fuse.reset_rec ? fuse.reset_rec + (long)blockIdx.y * fuse.reset_stride : nullptr;
#endif
    const unsigned long long my_lanes = (PL == 32 ? 0xFFFFFFFFull : 0xFFFFull) << (PL * half);
    const T MJ_MAXVAL = T(1e10);
    // ONE test per substep, where it ends (the acceleration, the integrated qpos and qvel); `rst_any` is wave-uniform (a
    // scalar register): a wave none of whose particles ever reset pays that test and one scalar branch per substep.
    // rst_pend (per lane, uniform over a particle): qpos / qvel left MuJoCo's bounds - the NEXT mj_step's mj_checkPos /
    // mj_checkVel resets, where that substep begins (until then the state, e.g. in the next observation, is what MuJoCo shows)
    bool rst_any = false, rst_pend = false, rst_ever = false;       // (rst_ever: this particle has reset - TreeFusion::inf_on_reset)
    auto state_is_bad = [&]() -> bool {
        bool bad = dof && (!(fabs(q) <= MJ_MAXVAL) || !(fabs(v) <= MJ_MAXVAL));
        if constexpr (GEN) bad = bad || (ball_g == 0 && (!(fabs(qy) <= MJ_MAXVAL) || !(fabs(qz) <= MJ_MAXVAL) || !(fabs(qw) <= MJ_MAXVAL)));
        return bad;
    };
    if (rst) {                  // the start state (the first mj_step's mj_checkPos / mj_checkVel)
        const unsigned long long bb = __ballot(state_is_bad());
        if (bb != 0ull) {
            rst_any = true;
            rst_pend = (bb & my_lanes) != 0ull;
        }
    }
    TreeClock clk;
    clk.start(diag, blockIdx.x == 0 && threadIdx.x == 0);

    // the inputs of step t + 1 are fetched while step t computes: consumed where they are loaded, every env step would
    // begin with a trip to HBM for the particle's sample that a wave alone on its SIMD cannot hide
    T eps_next = T(0);
    double mean_next = 0.0;
    if (has_u && H > 0) {
        if (!clw) mean_next = mean[l];
        if (noise && live) eps_next = noise[(pid * H) * A + l];
    }
    // fused into the launch (TreeFusion): the recursive noise filter and the discounted cost-to-go
    double fb0 = 1.0, fb1 = 0.0, fb2 = 0.0, fe1 = 0.0, fe2 = 0.0, q0acc = 0.0;
    if (fuse.filt) { fb0 = fuse.filt[0]; fb1 = fuse.filt[1]; fb2 = fuse.filt[2]; }
    double gs_next = (fuse.gseq && H > 0) ? fuse.gseq[0] : 0.0;
    for (int t = 0; t < H; ++t) {
        T u = T(0);
        const T eps_cur = eps_next;
        const double mean_cur = mean_next, gs_cur = gs_next;
        if (has_u && t + 1 < H) {
            if (!clw) mean_next = mean[(t + 1) * A + l];
            if (noise && live) eps_next = noise[(pid * H + t + 1) * A + l];
        }
        if (fuse.gseq && t + 1 < H) gs_next = fuse.gseq[t + 1];
        if (clw) {
            // mean_act = W' [obs; 1] with the observation this step starts from (gym_env_wrapper.py:135-136): every
            // lane weighs the entries it holds, one 32-lane sum per action
            // (general models: qpos in MuJoCo's layout - my coordinate sits at qadr, offset by qoff; a ball joint's first
            // link holds the quaternion, which the observation shows as q0 * q_link; velocities follow the nq coordinates)
            const int skip = task == 1 ? obs_skip : 0;
            const int iq = (GEN ? qadr : l) - skip, iv = nq - skip + l;
            T oq[4] = {q + (GEN ? qoff : T(0)), T(0), T(0), T(0)};
            if (GEN && ball_g == 0) {
                const T x0 = qoff;
                oq[0] = q0w * qw - x0 * q - q0y * qy - q0z * qz;
                oq[1] = q0w * q + x0 * qw + q0y * qz - q0z * qy;
                oq[2] = q0w * qy - x0 * qz + q0y * qw + q0z * q;
                oq[3] = q0w * qz + x0 * qy - q0y * q + q0z * qw;
            }
            const bool has_q = dof && iq >= 0 && (!GEN || ball_g <= 0);
            for (int a = 0; a < A; ++a) {
                T part = T(0);
                if (dof) {
                    if (has_q) part += (T)clw[iq * A + a] * oq[0];
                    if (GEN && ball_g == 0 && iq >= 0)
                        for (int k = 1; k < 4; ++k) part += (T)clw[(iq + k) * A + a] * oq[k];
                    part += (T)clw[iv * A + a] * v;
                }
                if (task == 0 && l < 3)
                    part += (T)clw[(nq + nv + l) * A + a] * hand_prev[l] + (T)clw[(nq + nv + 3 + l) * A + a] * (hand_prev[l] - tgt[l]);
                const T sa = sum_lanes<PL>(part) + (T)clw[dobs * A + a];
                if (l == a) u = sa;
            }
        } else if (has_u) {
            u = (T)mean_cur;
        }
        if (has_u) {
            if (noise && live) {
                T eps = eps_cur;
                if (fuse.filt) {            // eps[t] = b0 eps[t] + b1 eps[t-1] + b2 eps[t-2], t >= 2 (control_utils.py:32-33)
                    const double f = t >= 2 ? fb0 * (double)eps + fb1 * fe1 + fb2 * fe2 : (double)eps;
                    fe2 = fe1;
                    fe1 = f;
                    eps = (T)f;
                }
                u += eps;
            }
            if (act && live) act[(pid * H + t) * A + l] = u;        // unclipped (gym_env_wrapper.py:151)
        }
        // lane a holds action a; the dof it drives picks it up (motors may sit on any subset of the joints)
        const T u_dof = __shfl(u, act_id >= 0 ? act_id : 0, PL);
        T tau_act = act_id >= 0 ? M[T_GEAR + l] * fmin(fmax(u_dof, M[T_CTRL_LO + l]), M[T_CTRL_HI + l]) : T(0);
        if (task == 1 && l == 0) X[A_MISC + 4] = q;        // qpos[0] when the env step starts
        T hand[3] = {T(0), T(0), T(0)}, haxis[3] = {T(0), T(0), T(0)};
        for (int sub = 0; sub < frame_skip; ++sub) {
            // ---- 0. mj_checkPos / mj_checkVel found a NaN or an entry beyond mjMAXVAL in my particle's state (noted where the
            //         previous substep ended): mj_resetData - qpos0, zero velocity, zero controls until the env step ends -
            //         and the substep runs from there
            if (__builtin_expect(rst_any, 0)) {
                if (rst_pend) {
                    rst_pend = false;
                    q = T(0); v = T(0); sq = T(0); cq = T(1);
                    qy = T(0); qz = T(0); qw = T(1);
                    tau_act = act_id >= 0 ? M[T_GEAR + l] * fmin(fmax(T(0), M[T_CTRL_LO + l]), M[T_CTRL_HI + l]) : T(0);
                    lim_mem = 0; fl_mem = 0; cinst_mem = 0; cact_mem = 0;
                    xa_prev = T(0);
                    rst_ever = true;
                    if (diag && l == 0 && live) { atomicAdd(diag + 1, 1u); if (state_out) atomicAdd(diag + TREE_DIAG_ENV_RESETS, 1u); }
                }
            }
            // ---- 1. forward kinematics: X_l = X_parent o (Rodrigues(axis, q), off), by pointer jumping
            clk.mark(7);
            T R[9], p[3], ax[3];
            {
                const T s = sq, c = cq, tt = T(1) - cq;
                for (int k = 0; k < 3; ++k) { ax[k] = M[T_AXIS + k * TL + l]; p[k] = M[T_OFF + k * TL + l]; }
                R[0] = c + tt * ax[0] * ax[0];
                R[1] = tt * ax[0] * ax[1] - s * ax[2];
                R[2] = tt * ax[0] * ax[2] + s * ax[1];
                R[3] = tt * ax[0] * ax[1] + s * ax[2];
                R[4] = c + tt * ax[1] * ax[1];
                R[5] = tt * ax[1] * ax[2] - s * ax[0];
                R[6] = tt * ax[0] * ax[2] - s * ax[1];
                R[7] = tt * ax[1] * ax[2] + s * ax[0];
                R[8] = c + tt * ax[2] * ax[2];
                if (slide) {            // a slide joint moves its frame along the axis and does not turn it
                    for (int k = 0; k < 9; ++k) R[k] = (k & 3) == 0 ? T(1) : T(0);
                    for (int k = 0; k < 3; ++k) p[k] += ax[k] * q;
                }
                if constexpr (GEN) {
                    if (ball_g > 0) {                   // the ball's second and third link: no transform of their own
                        for (int k = 0; k < 9; ++k) R[k] = (k & 3) == 0 ? T(1) : T(0);
                    } else if (ball_g == 0) {
                        // MuJoCo's quaternion has its vector part in the BODY frame; link frames are world-aligned at qpos0, where
                        // the body's axes are the three links' axes: v_w = x ax_x + y ax_y + z ax_z, R = quat2mat(w, v_w)
                        T vw[3];
                        for (int k = 0; k < 3; ++k)
                            vw[k] = q * ax[k] + qy * M[T_AXIS + k * TL + l + 1] + qz * M[T_AXIS + k * TL + l + 2];
                        const T w_ = qw, x_ = vw[0], y_ = vw[1], z_ = vw[2];
                        R[0] = T(1) - T(2) * (y_ * y_ + z_ * z_); R[1] = T(2) * (x_ * y_ - w_ * z_); R[2] = T(2) * (x_ * z_ + w_ * y_);
                        R[3] = T(2) * (x_ * y_ + w_ * z_); R[4] = T(1) - T(2) * (x_ * x_ + z_ * z_); R[5] = T(2) * (y_ * z_ - w_ * x_);
                        R[6] = T(2) * (x_ * z_ - w_ * y_); R[7] = T(2) * (y_ * z_ + w_ * x_); R[8] = T(1) - T(2) * (x_ * x_ + y_ * y_);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (k >= tp.jumps) break;
#pragma unroll
                for (int c = 0; c < 9; ++c) X[c * PL + l] = R[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) X[(9 + c) * PL + l] = p[c];
                TSYNC();
                const int a = tp.anc[k];
                if (a >= 0) {
                    T Ra[9], pa[3], Rn[9], tv[3];
#pragma unroll
                    for (int c = 0; c < 9; ++c) Ra[c] = X[c * PL + a];
#pragma unroll
                    for (int c = 0; c < 3; ++c) pa[c] = X[(9 + c) * PL + a];
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j)
                            Rn[3 * i + j] = Ra[3 * i] * R[j] + Ra[3 * i + 1] * R[3 + j] + Ra[3 * i + 2] * R[6 + j];
                    mv3(Ra, p, tv);
                    for (int c = 0; c < 3; ++c) p[c] = pa[c] + tv[c];
                    for (int c = 0; c < 9; ++c) R[c] = Rn[c];
                }
                TSYNC();
            }
            // tracked site (its link's lane publishes it; consumed once per env step)
            if (l == site_link) {
                const T sp[3] = {M[T_SITE_POS], M[T_SITE_POS + 1], M[T_SITE_POS + 2]};
                T tv[3];
                mv3(R, sp, tv);
                for (int k = 0; k < 3; ++k) X[A_MISC + k] = p[k] + tv[k];
                if (FRIC && task == 2) {        // the object's axis in the world
                    const T sa[3] = {M[T_SITE_AXIS], M[T_SITE_AXIS + 1], M[T_SITE_AXIS + 2]};
                    mv3(R, sa, tv);
                    for (int k = 0; k < 3; ++k) X[A_MISC + 5 + k] = tv[k];
                }
            }
            // contact points: centre, signed distance (and the capsule axis the contact frame is aligned with), one
            // point per lane - every link publishes its frame, lane s reads the frame of point s's link
            unsigned cinst = 0, ucinst = 0;     // points in contact: of my particle / of either particle of the wavefront
            if (n_sphere > 0) {
                bool ci_mine = false;
#pragma unroll
                for (int c = 0; c < 9; ++c) X[c * PL + l] = R[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) X[(9 + c) * PL + l] = p[c];
                if constexpr (GEN) VEC[l] = q;          // (dof rows read their joints' coordinates)
                TSYNC();
                // (a ball-joint limit's record reads its joint's quaternion from the link that holds it)
                T bq[3] = {T(0), T(0), T(0)};
                if (GEN && has_ball) {
                    const bool blim = l < n_sphere && M[T_SPH + l * TREE_SPH_STRIDE + 12] == T(PT_DOFROW) && PEXT[l * TREE_PEXT_STRIDE] == T(2);
                    const int src = blim ? (int)PEXT[l * TREE_PEXT_STRIDE + 1] : l;
                    bq[0] = __shfl(qy, src, PL);
                    bq[1] = __shfl(qz, src, PL);
                    bq[2] = __shfl(qw, src, PL);
                }
                if (l < n_sphere) {
                    const T* sp = M + T_SPH + l * TREE_SPH_STRIDE;
                    const int sl = (int)sp[0];
                    T Rl[9], pl[3], tv[3];
#pragma unroll
                    for (int c = 0; c < 9; ++c) Rl[c] = X[c * PL + sl];
                    T* cs = X + A_CS + l * CS;
                    if constexpr (!FRIC) {          // (the lean instantiation: spheres against the plane, frictionless)
                        T ctr[3];
                        mv3(Rl, sp + 1, tv);
                        for (int k = 0; k < 3; ++k) ctr[k] = X[(9 + k) * PL + sl] + tv[k];
                        for (int k = 0; k < 3; ++k) cs[k] = ctr[k];
                        const T cdist = dot3(ctr, pn) - M[T_PLANE_D] - sp[4];
                        cs[3] = cdist;
                        ci_mine = cdist < sp[5];
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) pl[c] = FRIC ? X[(9 + c) * PL + sl] : T(0);
                    bool extra = false;
                    if constexpr (GEN >= 2) {
                        if (sp[12] >= T(PT_PLANE_CYL)) {
                            // round 5's record kinds - a cylinder on the plane, a capsule against a box, two boxes (extra_geometry)
                            extra = true;
                            const int sb = (int)sp[13];
                            LinkPose<T> fa, fb;
#pragma unroll
                            for (int c = 0; c < 9; ++c) { fa.R[c] = Rl[c]; fb.R[c] = sb >= 0 ? X[c * PL + sb] : ((c & 3) == 0 ? T(1) : T(0)); }
#pragma unroll
                            for (int c = 0; c < 3; ++c) { fa.p[c] = pl[c]; fb.p[c] = sb >= 0 ? X[(9 + c) * PL + sb] : T(0); }
                            const ExtraContact<T> g = extra_geometry<T>((int)sp[12], fa, fb, sp, PEXT + l * TREE_PEXT_STRIDE, pn[0], pn[1], pn[2], M[T_PLANE_D]);
                            for (int k = 0; k < 3; ++k) { cs[k] = g.cp[k]; cs[8 + k] = g.n[k]; cs[11 + k] = g.t1[k]; }
                            cs[3] = g.dist;
                            ci_mine = g.hit;
                        }
                    }
                    if (!FRIC || extra) {
                    } else if (sp[12] == T(0)) {
                        // sphere / capsule end against the plane (mjc_PlaneSphere, mjc_PlaneCapsule)
                        T ctr[3];
                        mv3(Rl, sp + 1, tv);
                        for (int k = 0; k < 3; ++k) ctr[k] = pl[k] + tv[k];
                        const T cdist = dot3(ctr, pn) - M[T_PLANE_D] - sp[4];
                        cs[3] = cdist;
                        ci_mine = cdist < sp[5];            // mj_collision: included while dist < margin
                        if (GEN >= 2 && sp[23] == T(8)) {   // a box's corner (mjc_PlaneBox): not one above the box centre ...
                            T tc[3];
                            mv3(Rl, sp + 14, tc);
                            const T up = (tv[0] - tc[0]) * pn[0] + (tv[1] - tc[1]) * pn[1] + (tv[2] - tc[2]) * pn[2];
                            ci_mine = ci_mine && !(up > T(0));      // (... and at most four per box: below, behind the ballot)
                        }
                        if (FRIC) {                         // the contact point, and the frame aligned with the capsule axis
                            for (int k = 0; k < 3; ++k) cs[k] = ctr[k] - pn[k] * (sp[4] + T(0.5) * cdist);
                            mv3(Rl, sp + 8, tv);
                            frame_tangent(pn, tv, cs + 11);
                        } else {
                            for (int k = 0; k < 3; ++k) cs[k] = ctr[k];
                        }
                    } else if (GEN && sp[12] == T(PT_DOFROW)) {
                        // a row over one or two joint coordinates (anchor dof A = sp[0], the other, B, above it): a joint
                        // equality q1 = poly(q2) (always there, bilateral) or a fixed-tendon limit on cA qA + cB qB
                        const T* ex = PEXT + l * TREE_PEXT_STRIDE;
                        const int dB = (int)sp[13];
                        const T qA = VEC[sl], qB = dB >= 0 ? VEC[dB] : T(0);
                        if (ex[0] == T(2)) {
                            // ball-joint limit (mj_instantiateLimit, mjJNT_BALL): the joint quaternion as axis * angle
                            // (mju_quat2Vel: angle = 2 atan2(|xyz|, w), wrapped to (-pi, pi]); dist = max(range) - |angle|,
                            // J = -axis over the joint's three dofs: mine (the last link) in cs[0], the middle one's in
                            // cs[1], the first's in cs[2]
                            const T x = VEC[(int)ex[1]], y = bq[0], z = bq[1], w = bq[2];
                            const T sn = sqrt_(x * x + y * y + z * z);
                            T ang = T(2) * (T)atan2((double)sn, (double)w);
                            if (ang > T(3.14159265358979323846)) ang -= T(2 * 3.14159265358979323846);
                            const T k = (sn > T(1e-15)) ? (ang < T(0) ? T(1) : T(-1)) * rcp_(sn) : T(0);
                            cs[3] = ex[4] - fabs(ang);
                            cs[0] = k * z;
                            cs[1] = k * y;
                            cs[2] = k * x;
                            ci_mine = sn > T(1e-15) && cs[3] < T(0);
                        } else if (ex[0] == T(0)) {
                            const bool swapped = ex[1] != T(0);         // the anchor dof is joint 2, joint 1 rides above it
                            const T x = swapped ? qA : qB, q1 = swapped ? qB : qA;
                            const T poly = ex[6] + x * (ex[7] + x * (ex[8] + x * (ex[9] + x * ex[10])));
                            const T dpoly = ex[7] + x * (T(2) * ex[8] + x * (T(3) * ex[9] + x * T(4) * ex[10]));
                            cs[3] = q1 - poly;
                            cs[0] = swapped ? -dpoly : T(1);            // entry of dof A, of dof B
                            cs[1] = swapped ? T(1) : (dB >= 0 ? -dpoly : T(0));
                            ci_mine = true;
                        } else {
                            const T len = ex[1] * qA + ex[2] * qB, dlo = len - ex[3], dhi = ex[4] - len;
                            const bool lo = dlo < ex[5], hi = !lo && dhi < ex[5];
                            const T sg = lo ? T(1) : T(-1);
                            cs[3] = lo ? dlo : dhi;
                            cs[0] = sg * ex[1];
                            cs[1] = sg * ex[2];
                            ci_mine = lo || hi;
                        }
                        for (int k = 0; k < 3; ++k) { cs[8 + k] = T(0); cs[11 + k] = T(0); }
                    } else if (GEN && sp[12] == T(PT_CONNECT)) {
                        // connect equality: the anchor as a point of body A and as a point of body B (-1: the world)
                        const int sb = (int)sp[13];
                        mv3(Rl, sp + 1, tv);
                        for (int k = 0; k < 3; ++k) cs[k] = pl[k] + tv[k];
                        if (sb >= 0) {
                            T Rb[9];
#pragma unroll
                            for (int c = 0; c < 9; ++c) Rb[c] = X[c * PL + sb];
                            mv3(Rb, sp + 14, tv);
                            for (int k = 0; k < 3; ++k) cs[11 + k] = X[(9 + k) * PL + sb] + tv[k];
                        } else {
                            for (int k = 0; k < 3; ++k) cs[11 + k] = sp[14 + k];
                        }
                        cs[3] = T(0);
                        ci_mine = true;
                    } else if (GEN && sp[12] == T(PT_WELD)) {
                        // weld equality, rotation rows (mj_instantiateEquality, mjEQ_WELD): the error quaternion inv(q2) q1 qrel -
                        // with link frames world-aligned at qpos0 it is R0_2' (R_link2' R_link1) R0_2 - has the residual as
                        // its vector part and 1/2 (w u + u x v), u = R2' (w1 - w2), as its velocity: G = 1/2 R0_2' (w I - [v]x)
                        // R_link2' maps a dof's angular motion (w1 - w2) to its three Jacobian entries
                        const int sb = (int)sp[13];
                        const T* ex = PEXT + l * TREE_PEXT_STRIDE;
                        T Rb[9];
#pragma unroll
                        for (int c = 0; c < 9; ++c) Rb[c] = sb >= 0 ? X[c * PL + sb] : ((c & 3) == 0 ? T(1) : T(0));
                        const bool a_is_1 = ex[20] > T(0);
                        const T* R1 = a_is_1 ? Rl : Rb;
                        const T* R2 = a_is_1 ? Rb : Rl;
                        T E[9];
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j) E[3 * i + j] = R2[i] * R1[j] + R2[3 + i] * R1[3 + j] + R2[6 + i] * R1[6 + j];
                        T qe[4];
                        {
                            const T tr = E[0] + E[4] + E[8];
                            if (tr > T(0)) {
                                const T sq = sqrt_(tr + T(1)) * T(2), iv = T(1) / sq;
                                qe[0] = T(0.25) * sq; qe[1] = (E[7] - E[5]) * iv; qe[2] = (E[2] - E[6]) * iv; qe[3] = (E[3] - E[1]) * iv;
                            } else if (E[0] > E[4] && E[0] > E[8]) {
                                const T sq = sqrt_(T(1) + E[0] - E[4] - E[8]) * T(2), iv = T(1) / sq;
                                qe[0] = (E[7] - E[5]) * iv; qe[1] = T(0.25) * sq; qe[2] = (E[1] + E[3]) * iv; qe[3] = (E[2] + E[6]) * iv;
                            } else if (E[4] > E[8]) {
                                const T sq = sqrt_(T(1) + E[4] - E[0] - E[8]) * T(2), iv = T(1) / sq;
                                qe[0] = (E[2] - E[6]) * iv; qe[1] = (E[1] + E[3]) * iv; qe[2] = T(0.25) * sq; qe[3] = (E[5] + E[7]) * iv;
                            } else {
                                const T sq = sqrt_(T(1) + E[8] - E[0] - E[4]) * T(2), iv = T(1) / sq;
                                qe[0] = (E[3] - E[1]) * iv; qe[1] = (E[2] + E[6]) * iv; qe[2] = (E[5] + E[7]) * iv; qe[3] = T(0.25) * sq;
                            }
                            if (qe[0] < T(0)) for (int k = 0; k < 4; ++k) qe[k] = -qe[k];
                        }
                        // K = (w I - [v]x) R_link2'  (row i, column j), then G = 1/2 R0_2' K
                        const T w_ = qe[0], vx = qe[1], vy = qe[2], vz = qe[3];
                        const T Wm[9] = {w_, vz, -vy, -vz, w_, vx, vy, -vx, w_};
                        T K[9];
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j) K[3 * i + j] = Wm[3 * i] * R2[3 * j] + Wm[3 * i + 1] * R2[3 * j + 1] + Wm[3 * i + 2] * R2[3 * j + 2];
                        for (int i = 0; i < 3; ++i) {
                            for (int j = 0; j < 3; ++j)
                                cs[3 * i + j] = T(0.5) * (ex[i] * K[j] + ex[3 + i] * K[3 + j] + ex[6 + i] * K[6 + j]);
                            cs[11 + i] = ex[i] * vx + ex[3 + i] * vy + ex[6 + i] * vz;        // residual = R0_2' v
                        }
                        ci_mine = true;
                    } else if (GEN && (sp[12] == T(PT_SPHERE_BOX) || sp[12] == T(PT_BOX_SPHERE))) {
                        // a sphere against a box (mjc_SphereBox): the box's closest point to the sphere's centre, or - centre
                        // inside - the nearest face; normal from geom B to geom A, contact point midway between the surfaces
                        const bool boxA = sp[12] == T(PT_BOX_SPHERE);
                        const int sb = (int)sp[13];
                        const T* ex = PEXT + l * TREE_PEXT_STRIDE;
                        T Rb[9], oA[3], oB[3];
#pragma unroll
                        for (int c = 0; c < 9; ++c) Rb[c] = sb >= 0 ? X[c * PL + sb] : ((c & 3) == 0 ? T(1) : T(0));
                        mv3(Rl, sp + 1, tv);
                        for (int k = 0; k < 3; ++k) oA[k] = pl[k] + tv[k];
                        mv3(Rb, sp + 14, tv);
                        for (int k = 0; k < 3; ++k) oB[k] = (sb >= 0 ? X[(9 + k) * PL + sb] : T(0)) + tv[k];
                        const T* Rx = boxA ? Rl : Rb;                       // the box's link
                        const T* ob = boxA ? oA : oB;
                        const T* os = boxA ? oB : oA;
                        const T rs = boxA ? sp[17] : sp[4];
                        T Rw[9];
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j)
                                Rw[3 * i + j] = Rx[3 * i] * ex[3 + j] + Rx[3 * i + 1] * ex[6 + j] + Rx[3 * i + 2] * ex[9 + j];
                        const T rel[3] = {os[0] - ob[0], os[1] - ob[1], os[2] - ob[2]};
                        T loc[3], cl[3];
                        bool inside = true;
                        for (int i = 0; i < 3; ++i) {
                            loc[i] = Rw[i] * rel[0] + Rw[3 + i] * rel[1] + Rw[6 + i] * rel[2];
                            cl[i] = fmin(fmax(loc[i], -ex[i]), ex[i]);
                            inside = inside && cl[i] == loc[i];
                        }
                        T nb[3] = {T(0), T(0), T(0)}, len;
                        if (inside) {
                            int kk = 0;
                            T best = ex[0] - fabs(loc[0]);
                            for (int i = 1; i < 3; ++i) {
                                const T gap = ex[i] - fabs(loc[i]);
                                if (gap < best) { best = gap; kk = i; }
                            }
                            const T sg = loc[kk] >= T(0) ? T(1) : T(-1);
                            for (int i = 0; i < 3; ++i) {
                                if (i == kk) cl[i] = sg * ex[i];
                                nb[i] = sg * Rw[3 * i + kk];
                            }
                            len = -best;
                        }
                        T cb[3];
                        mv3(Rw, cl, cb);
                        for (int k = 0; k < 3; ++k) cb[k] += ob[k];
                        bool ok = true;
                        if (!inside) {
                            for (int k = 0; k < 3; ++k) nb[k] = os[k] - cb[k];
                            len = sqrt_(dot3(nb, nb));
                            ok = len > T(1e-14);
                            const T inv = ok ? rcp_(len) : T(0);
                            for (int k = 0; k < 3; ++k) nb[k] *= inv;
                        }
                        // nb points from the box to the sphere; the contact's normal from geom B to geom A
                        const T sgn = boxA ? T(-1) : T(1);
                        const T cdist = len - rs;
                        T nv3[3];
                        for (int k = 0; k < 3; ++k) {
                            nv3[k] = sgn * nb[k];
                            cs[8 + k] = nv3[k];
                            // midway between the surfaces: from geom B's surface point along the normal
                            const T onB = boxA ? os[k] + nv3[k] * rs : cb[k];
                            cs[k] = onB + nv3[k] * (T(0.5) * cdist);
                        }
                        const T zero3[3] = {T(0), T(0), T(0)};
                        frame_tangent(nv3, zero3, cs + 11);
                        cs[3] = cdist;
                        ci_mine = ok && cdist < sp[5];
                    } else {
                        // geom-geom (mjc_SphereSphere / SphereCapsule / CapsuleCapsule): closest points of the two
                        // segments; normal from the object's geom (B) to the manipulator's (A), point midway
                        const int sb = (int)sp[13];
                        T Rb[9], o1[3], d1[3], o2[3], d2[3];
#pragma unroll
                        for (int c = 0; c < 9; ++c) Rb[c] = (!GEN || sb >= 0) ? X[c * PL + sb] : ((c & 3) == 0 ? T(1) : T(0));
                        mv3(Rl, sp + 1, tv);
                        for (int k = 0; k < 3; ++k) o1[k] = pl[k] + tv[k];
                        mv3(Rl, sp + 8, d1);
                        mv3(Rb, sp + 14, tv);
                        for (int k = 0; k < 3; ++k) o2[k] = ((!GEN || sb >= 0) ? X[(9 + k) * PL + sb] : T(0)) + tv[k];
                        mv3(Rb, sp + 18, d2);
                        T ss, tt;
                        {
                            const T r3[3] = {o1[0] - o2[0], o1[1] - o2[1], o1[2] - o2[2]};
                            const T a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r3), EPS = T(1e-18);
                            auto clamp01 = [](T x) { return x < T(0) ? T(0) : (x > T(1) ? T(1) : x); };
                            // (round 6: the clamped closed form by selects and three reciprocals made side by side - the nested
                            // cases with their IEEE divisions ran one after the other for the record kinds of a wavefront: sphere /
                            // sphere, sphere / capsule and capsule / capsule pairs take different ones)
                            const T c = dot3(d1, r3), b = dot3(d1, d2), denom = a * e - b * b;
                            const bool za = a <= EPS, ze = e <= EPS, par = !(denom > T(1e-12) * a * e);
                            const T ia = rcp_(za ? T(1) : a), ie = rcp_(ze ? T(1) : e), idn = rcp_(par ? T(1) : denom);
                            const T ss_lo = clamp01(-c * ia), ss_hi = clamp01((b - c) * ia);
                            T ssg = par ? T(0) : clamp01((b * f - c * e) * idn);
                            T ttg = (b * ssg + f) * ie;
                            ssg = ttg < T(0) ? ss_lo : (ttg > T(1) ? ss_hi : ssg);
                            ttg = clamp01(ttg);
                            ss = za ? T(0) : (ze ? ss_lo : ssg);
                            tt = za ? (ze ? T(0) : clamp01(f * ie)) : (ze ? T(0) : ttg);
                        }
                        T c2[3], diff[3];
                        for (int k = 0; k < 3; ++k) {
                            c2[k] = o2[k] + tt * d2[k];
                            diff[k] = o1[k] + ss * d1[k] - c2[k];
                        }
                        const T dd = dot3(diff, diff), rdd = rsqrt_(dd > T(0) ? dd : T(1));
                        const T len = dd * rdd;
                        const T inv = len > T(1e-14) ? rdd : T(0);
                        const T cdist = len - sp[4] - sp[17];
                        cs[3] = cdist;
                        ci_mine = len > T(1e-14) && cdist < sp[5];
                        // (the contact's point, normal and frame only where there is a contact: a record out of its margin is
                        // never read - the swimmer's six pairs, never in touch on the bench's trajectories, skip a fifth of
                        // this branch)
                        if (ci_mine) {
                            T nv3[3];
                            for (int k = 0; k < 3; ++k) {
                                const T nk = diff[k] * inv;
                                nv3[k] = nk;
                                cs[8 + k] = nk;
                                cs[k] = c2[k] + nk * (sp[17] + T(0.5) * cdist);
                            }
                            const T zero3[3] = {T(0), T(0), T(0)};
                            frame_tangent(nv3, zero3, cs + 11);
                        }
                    }
                }
                unsigned long long b = __ballot(ci_mine);
                if constexpr (GEN >= 2) {
                    // mjc_PlaneBox keeps a box's first four contacts, in corner order: a corner with four counted corners of its
                    // box (records l - k .. l - 1 of its group) ahead of it is dropped
                    bool capped = false;
                    if (l < n_sphere && M[T_SPH + l * TREE_SPH_STRIDE + 23] == T(8)) {
                        const int k = (int)M[T_SPH + l * TREE_SPH_STRIDE + 22];
                        const unsigned mine = (unsigned)(b >> (PL * half));
                        const unsigned ahead = mine & ((1u << l) - 1u) & ~((1u << (l - k)) - 1u);
                        capped = __popc(ahead) >= 4;
                    }
                    if (__any(capped)) b = __ballot(ci_mine && !capped);
                }
                cinst = (unsigned)(b >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu);
                ucinst = (unsigned)b | (unsigned)(b >> 32);
                if (PL == 16) ucinst = (ucinst | (ucinst >> 16)) & 0xFFFFu;
            }
            TSYNC();
            if (sub == frame_skip - 1 || (t == 0 && sub == 0))
                for (int k = 0; k < 3; ++k) {
                    hand[k] = X[A_MISC + k];
                    if constexpr (FRIC) haxis[k] = X[A_MISC + 5 + k];      // (task 2 runs the full instantiation)
                }
            if (t == 0 && sub == 0) {
                for (int k = 0; k < 3; ++k) hand_prev[k] = hand[k];     // fresh observation after set_env_state
                if (site_out && pid == 0 && l < 3) site_out[l] = (double)hand[l];
                if (fuse.axis_out && pid == 0 && l < 3) fuse.axis_out[l] = (double)haxis[l];
            }

            // ---- 2. world-frame quantities of my link, about the world origin
            clk.mark(0);
            const T mass = M[T_MASS + l];
            T a[3], cw[3], tv[3], Ib[6], hm[3], sw[3], sv[3];
            mv3(R, ax, a);
            {
                const T com[3] = {M[T_COM + l], M[T_COM + TL + l], M[T_COM + 2 * TL + l]};
                mv3(R, com, tv);
            }
            for (int k = 0; k < 3; ++k) cw[k] = p[k] + tv[k];
            {
                T Il[6], RI[9];
                for (int k = 0; k < 6; ++k) Il[k] = M[T_INERTIA + k * TL + l];
                for (int i = 0; i < 3; ++i) {
                    const T* r = R + 3 * i;
                    RI[3 * i + 0] = r[0] * Il[0] + r[1] * Il[3] + r[2] * Il[4];
                    RI[3 * i + 1] = r[0] * Il[3] + r[1] * Il[1] + r[2] * Il[5];
                    RI[3 * i + 2] = r[0] * Il[4] + r[1] * Il[5] + r[2] * Il[2];
                }
                const T cc = dot3(cw, cw);
                Ib[0] = dot3(RI + 0, R + 0) + mass * (cc - cw[0] * cw[0]);
                Ib[1] = dot3(RI + 3, R + 3) + mass * (cc - cw[1] * cw[1]);
                Ib[2] = dot3(RI + 6, R + 6) + mass * (cc - cw[2] * cw[2]);
                Ib[3] = dot3(RI + 0, R + 3) - mass * cw[0] * cw[1];
                Ib[4] = dot3(RI + 0, R + 6) - mass * cw[0] * cw[2];
                Ib[5] = dot3(RI + 3, R + 6) - mass * cw[1] * cw[2];
            }
            // motion subspace about the world origin: hinge (a, p x a), slide (0, a)
            for (int k = 0; k < 3; ++k) { hm[k] = mass * cw[k]; sw[k] = slide ? T(0) : a[k]; }
            cross3(p, a, sv);
            if (slide)
                for (int k = 0; k < 3; ++k) sv[k] = a[k];

            // ---- 3. bias force: spatial velocity and velocity-product acceleration along the path to the root,
            //         body forces summed over the subtree
            T bias;
            {
                T V[6], Ac[6];
                for (int k = 0; k < 3; ++k) { V[k] = sw[k] * v; V[3 + k] = sv[k] * v; }
                T xw[3] = {V[0], V[1], V[2]}, xv[3] = {V[3], V[4], V[5]};
                path_sum<6, DP, PL, (DN > 0 ? DN : 16)>(V, tp, X, l);
                T dw[3], d1[3], d2[3];
                if (GEN && has_ball) {
                    // a ball joint's three axes move with the BODY: sum_k d/dt(S_k) v_k = V_before x sum_k S_k v_k, V_before the
                    // velocity in front of the joint (MuJoCo's mj_comVel takes all three cdof_dot with it) - my cumulative
                    // velocity less my own and my group predecessors' contributions
                    T Vb[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        const T own = c < 3 ? xw[c] : xv[c - 3];
                        const T s1 = __shfl_up(own, 1, PL), s2 = __shfl_up(own, 2, PL);
                        Vb[c] = V[c] - own - (ball_g >= 1 ? s1 : T(0)) - (ball_g == 2 ? s2 : T(0));
                    }
                    cross3(Vb, xw, dw);
                    cross3(Vb, xv, d1);
                    cross3(Vb + 3, xw, d2);
                } else {
                cross3(V, xw, dw);
                cross3(V, xv, d1);
                cross3(V + 3, xw, d2);
                }
                for (int k = 0; k < 3; ++k) { Ac[k] = dw[k]; Ac[3 + k] = d1[k] + d2[k]; }
                path_sum<6, DP, PL, (DN > 0 ? DN : 16)>(Ac, tp, X, l);
                for (int k = 0; k < 3; ++k) Ac[3 + k] -= M[T_GRAVITY + k];          // base acceleration -g
                // f = I A + V x* (I V),  I(w, v) = (Ib w + h x v, m v - h x w)
                T nV[3], fV[3], nA[3], fA[3], t1[3], t2[3], c1[3], c2[3], c3[3], f[6];
                symv3(Ib, V, nV);
                cross3(hm, V + 3, t1);
                cross3(hm, V, t2);
                for (int k = 0; k < 3; ++k) { nV[k] += t1[k]; fV[k] = mass * V[3 + k] - t2[k]; }
                symv3(Ib, Ac, nA);
                cross3(hm, Ac + 3, t1);
                cross3(hm, Ac, t2);
                for (int k = 0; k < 3; ++k) { nA[k] += t1[k]; fA[k] = mass * Ac[3 + k] - t2[k]; }
                cross3(V, nV, c1);
                cross3(V + 3, fV, c2);
                cross3(V, fV, c3);
                for (int k = 0; k < 3; ++k) { f[k] = nA[k] + c1[k] + c2[k]; f[3 + k] = fA[k] + c3[k]; }
                if (fluid) {
                    // MuJoCo mj_passive, inertia-box fluid model: viscous and drag wrench on the box of equal inertia,
                    // evaluated in the link's inertial frame at its centre of mass; an external wrench on my link
                    // leaves the force balance the subtree sums collect
                    T Xf[9], vc[3], lw[3], lv[3], lf[3], lt[3], bx[3], wf[3], wt[3], t3[3];
                    {
                        T Fr[9];
                        for (int k = 0; k < 9; ++k) Fr[k] = M[T_FROT + k * TL + l];
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j)
                                Xf[3 * i + j] = R[3 * i] * Fr[j] + R[3 * i + 1] * Fr[3 + j] + R[3 * i + 2] * Fr[6 + j];
                    }
                    cross3(V, cw, t3);
                    for (int k = 0; k < 3; ++k) { vc[k] = V[3 + k] + t3[k]; bx[k] = M[T_FBOX + k * TL + l]; }
                    for (int i = 0; i < 3; ++i) {
                        lw[i] = Xf[i] * V[0] + Xf[3 + i] * V[1] + Xf[6 + i] * V[2];
                        lv[i] = Xf[i] * vc[0] + Xf[3 + i] * vc[1] + Xf[6 + i] * vc[2];
                    }
                    const T visc = M[T_VISCOSITY], rho = M[T_DENSITY], PI = T(3.14159265358979323846);
                    const T diam = (bx[0] + bx[1] + bx[2]) * T(1.0 / 3.0);
                    for (int i = 0; i < 3; ++i) {
                        const int j1 = (i + 1) % 3, j2 = (i + 2) % 3;
                        const T b1 = bx[j1] * bx[j1], b2 = bx[j2] * bx[j2];
                        lt[i] = -PI * diam * diam * diam * visc * lw[i] -
                                rho * bx[i] * (b1 * b1 + b2 * b2) * fabs(lw[i]) * lw[i] * T(1.0 / 64.0);
                        lf[i] = -T(3) * PI * diam * visc * lv[i] - T(0.5) * rho * bx[j1] * bx[j2] * fabs(lv[i]) * lv[i];
                    }
                    mv3(Xf, lf, wf);
                    mv3(Xf, lt, wt);
                    cross3(cw, wf, t3);
                    if (mass > T(0))
                        for (int k = 0; k < 3; ++k) { f[k] -= wt[k] + t3[k]; f[3 + k] -= wf[k]; }
                }
                subtree_sum<6, PL, NLINKS>(f, tp, X, l);
                bias = dot3(sw, f) + dot3(sv, f + 3);
            }

            // ---- 4. composite inertia over the subtree, F = Ic S, my path-indexed mass-matrix row
            //         mrow[c] = M[l][ancestor at distance c] = S_anc . F_l
            clk.mark(1);
            T mrow[DP];
            // dense path: my row of M as a DENSE row, once per substep - the Newton matrices and the Euler matrix are this
            // row plus diagonal terms plus broadcast products of the contact Jacobians (no tile round trip per matrix)
            T md[DN > 0 ? DN : 1];
            {
                T c6[6] = {Ib[0], Ib[1], Ib[2], Ib[3], Ib[4], Ib[5]}, c4[4] = {mass, hm[0], hm[1], hm[2]};
                subtree_sum<6, PL, NLINKS>(c6, tp, X, l);
                subtree_sum<4, PL, NLINKS>(c4, tp, X, l);
                T F[6], t1[3], t2[3];
                symv3(c6, sw, F);
                cross3(c4 + 1, sv, t1);
                cross3(c4 + 1, sw, t2);
                for (int k = 0; k < 3; ++k) { F[k] += t1[k]; F[3 + k] = c4[0] * sv[k] - t2[k]; }
                if constexpr (DN > 0) {
                    // 16-lane particles: my DENSE row straight from DPP broadcasts - column j is S_j . F_l where j is one of
                    // my ancestors (or me), S_l . F_j where j lies in my subtree, zero otherwise; no LDS, no path-indexed row
                    const T Sl[6] = {sw[0], sw[1], sw[2], sv[0], sv[1], sv[2]};
                    if constexpr (DN == 32) {
                        T Se[6], So[6], Fe[6], Fo[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) { row_pair(Sl[k], Se[k], So[k]); row_pair(F[k], Fe[k], Fo[k]); }
                        dense32_mass_row<0>(md, Se, Fe, So, Fo, Sl, F, l, tp.ancmask, tp.subsize, dof);
                    } else
                    dense_mass_row<0, DN>(md, Sl, F, l, tp.ancmask, tp.subsize, dof);
#pragma unroll
                    for (int j = 0; j < DN; ++j) md[j] = (j == l) ? (dof ? md[j] + armature : T(1)) : md[j];
#pragma unroll
                    for (int c = 0; c < DP; ++c) mrow[c] = T(0);
                    mrow[0] = T(1);
                } else {
                T* S_ = X + A_SF;
                for (int k = 0; k < 3; ++k) { S_[k * PL + l] = sw[k]; S_[(3 + k) * PL + l] = sv[k]; }
                TSYNC();
#pragma unroll
                for (int c = 0; c < DP; ++c) {
                    const int an = AT[c * PL + l];
                    T sacc = T(0);
                    // (full instantiation: elimination-tree ancestors that are not kinematic ones - the object's links
                    // above a manipulator - have no entry; the lean one has no geom-geom pairs, the two trees coincide)
                    if (an >= 0 && (!FRIC || ((tp.ancmask >> an) & 1u))) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) sacc += S_[k * PL + an] * F[k];
                    }
                    mrow[c] = sacc;
                }
                mrow[0] = dof ? mrow[0] + armature : T(1);     // spare lanes: unit diagonal, no ancestors
                TSYNC();                    // S_ lies inside the area the factorisation publishes rows to
                }
            }
            // (position servos: the bias -kp * (gear q) of MJCF <position>, a stiffness gear^2 kp about 0 at the joint)
            // (... and the rest of an affine actuator bias: -gear^2 b2 v, gear b0 - explicit terms, mj_fwdActuation)
            // ... the actuator's joint torque clamped to its forcerange (+-inf without one) before it joins the passive forces)
            // (an actuator on a fixed tendon: length and velocity are the tendon's, over my dof and the tendon's other one, and my
            // share of its force is my coefficient's)
            T alen = q, avel = v, acoef = T(1);
            if (FRIC && any_tendon_act) {
                const int tp = (int)M[T_TPARTNER + l];
                const T pq = __shfl(q, tp >= 0 ? tp : 0, PL), pv = __shfl(v, tp >= 0 ? tp : 0, PL);
                acoef = M[T_TCOEF + l];
                alen = acoef * q + (tp >= 0 ? M[T_TPCOEF + l] * pq : T(0));
                avel = acoef * v + (tp >= 0 ? M[T_TPCOEF + l] * pv : T(0));
            }
            const T tau_a = FRIC ? acoef * fmin(fmax(tau_act - M[T_KPG + l] * alen - M[T_KVG + l] * avel + M[T_TAU0 + l], M[T_TAU_LO + l]), M[T_TAU_HI + l]) : tau_act;
            const T tau = dof ? -bias - damping * v - (FRIC ? M[T_STIFFNESS + l] * (q - M[T_SPRINGREF + l]) : T(0)) + tau_a : T(0);

            // ---- 5. constraint rows: joint limits (mj_instantiateLimit, strict dist < 0) ...
            clk.mark(2);
            T sig = T(0), dist = T(0), D = T(0), aref = T(0);
            bool inst = false;
            if (dof && M[T_LIMITED + l] != T(0)) {
                const T dlo = q - M[T_RANGE_LO + l], dhi = M[T_RANGE_HI + l] - q;
                if (dlo < jmargin) { sig = T(1); dist = dlo; inst = true; }
                else if (dhi < jmargin) { sig = T(-1); dist = dhi; inst = true; }
            }
            // ... and plane-sphere contacts (mjc_PlaneSphere / the two ends mjc_PlaneCapsule tests): Jacobian rows in
            // LDS, scalars per contact point.  condim 1: one row Jn.  condim 3 (FRIC): MuJoCo's pyramidal cone - the four
            // rows Jn +- mu Jt_k in the frame mju_makeFrame builds from the normal and the capsule axis, every row with
            // diagApprox = tran (1 + mu^2), R = 2 mu^2 R_first, and its own reference acceleration.
            // my entry of the Jacobian rows of point s: the distance from its link up to me, -1 if I am not on that path
            // (or the point is not in contact for my particle)
            // (where I sit on the point's path is a constant of the model, packed once per launch: own_path)
            auto own_idx = [&](int s) -> int {
                const int code = s < 12 ? (int)((own_path0 >> (5 * s)) & 31ull) : (int)((own_path1 >> (5 * (s - 12))) & 31u);
                return ((cinst >> s) & 1u) ? code - 1 : -1;
            };
            for (unsigned um = ucinst; um; um &= um - 1) {
                const int s = __builtin_ctz(um);
                const T* sp = M + T_SPH + s * TREE_SPH_STRIDE;
                const T* cs = X + A_CS + s * CS;
                T* jrow = X + A_JC + s * NJ * DP;
                const bool ci = (cinst >> s) & 1u;
                const int dsl = (int)sp[11], oi = own_idx(s);
                // velocity of the contact point per unit joint velocity, from the motion subspace about the world
                // origin: g = sw x c + sv  (hinge: a x (c - p); slide: a)
                // ... of the material point of every link that carries one of the two geoms: + on the side of geom A,
                // - on the side of geom B (the world's plane has no side), 0 for elimination-path links that carry neither
                T r[3], g[3], nbuf[3];
                const T* nrm = pn;
                T side = T(1);
                if constexpr (FRIC) {
                    const bool geomgeom = sp[12] != T(0);
                    for (int k = 0; k < 3; ++k) { r[k] = cs[k]; nbuf[k] = geomgeom ? cs[8 + k] : pn[k]; }
                    nrm = nbuf;
                    const int lA = (int)sp[0], lB = (int)sp[13];
                    side = T((lA >= l && lA < l + tp.subsize) ? 1 : 0) - T((lB >= l && lB < l + tp.subsize) ? 1 : 0);
                } else {
                    for (int k = 0; k < 3; ++k) r[k] = cs[k] - pn[k] * (sp[4] + T(0.5) * cs[3]);
                }
                cross3(sw, r, g);
                for (int k = 0; k < 3; ++k) g[k] += sv[k];
                T jc = oi >= 0 ? (FRIC ? side * dot3(nrm, g) : dot3(pn, g)) : T(0);
                T jgen1 = T(0), jgen2 = T(0);
                bool genrow = false;
                if constexpr (GEN) {
                    const int kind = (int)sp[12];
                    if (kind == PT_DOFROW) {            // my joint coordinate's coefficient
                        genrow = true;
                        const int lC = PEXT[s * TREE_PEXT_STRIDE] == T(2) ? (int)PEXT[s * TREE_PEXT_STRIDE + 1] : -1;      // (a ball limit's third dof)
                        jc = oi >= 0 ? (l == (int)sp[0] ? cs[0] : (l == (int)sp[13] ? cs[1] : (l == lC ? cs[2] : T(0)))) : T(0);
                    } else if (kind == PT_CONNECT) {
                        // rows along the world axes: what my dof moves body A's anchor, less what it moves body B's
                        genrow = true;
                        const int lA = (int)sp[0], lB = (int)sp[13];
                        const T inA = (lA >= l && lA < l + tp.subsize) ? T(1) : T(0);
                        const T inB = (lB >= l && lB < l + tp.subsize) ? T(1) : T(0);
                        const T pB[3] = {cs[11], cs[12], cs[13]};
                        T gB[3];
                        cross3(sw, pB, gB);             // (r = cs[0:3] is A's anchor: g above)
                        const T e0 = inA * g[0] - inB * (gB[0] + sv[0]), e1 = inA * g[1] - inB * (gB[1] + sv[1]),
                                e2 = inA * g[2] - inB * (gB[2] + sv[2]);
                        jc = oi >= 0 ? e0 : T(0);
                        jgen1 = oi >= 0 ? e1 : T(0);
                        jgen2 = oi >= 0 ? e2 : T(0);
                    } else if (kind == PT_WELD) {
                        // my dof's angular motion of body 1 less that of body 2, through the record's G
                        genrow = true;
                        const int lA = (int)sp[0], lB = (int)sp[13];
                        const T inA = (lA >= l && lA < l + tp.subsize) ? T(1) : T(0);
                        const T inB = (lB >= l && lB < l + tp.subsize) ? T(1) : T(0);
                        const T sd = PEXT[s * TREE_PEXT_STRIDE + 20] > T(0) ? inA - inB : inB - inA;
                        const T dwv[3] = {sd * sw[0], sd * sw[1], sd * sw[2]};
                        jc = oi >= 0 ? cs[0] * dwv[0] + cs[1] * dwv[1] + cs[2] * dwv[2] : T(0);
                        jgen1 = oi >= 0 ? cs[3] * dwv[0] + cs[4] * dwv[1] + cs[5] * dwv[2] : T(0);
                        jgen2 = oi >= 0 ? cs[6] * dwv[0] + cs[7] * dwv[1] + cs[8] * dwv[2] : T(0);
                    }
                }
                if (oi >= 0) jrow[oi] = jc;
                if (ci && l > dsl && l < DP) jrow[l] = T(0);        // past the root: read by shorter paths' lanes
                if constexpr (!FRIC) {      // (one frictionless row per point: a lane sum is cheaper than the walk below)
                    const T jv = sum_lanes<PL>(jc * v);
                    T Dc, arc;
                    tree_row_params(M + T_SOLTAB + 7 * (int)sp[21], cs[3] - sp[5], sp[6], jv, Dc, arc);
                    if (l == 0 && ci) {
                        X[A_CS + s * CS + 4] = Dc;
                        X[A_CS + s * CS + 5] = arc;
                    }
                }
                if (FRIC) {
                    // the contact frame's tangents: the first from the geometry stage (frame_tangent), t2 = n x t1
                    const T t1[3] = {cs[11], cs[12], cs[13]};
                    T t2[3];
                    cross3(nrm, t1, t2);
                    const bool fr = oi >= 0 && sp[7] > T(0);
                    T j1 = fr ? side * dot3(t1, g) : T(0), j2 = fr ? side * dot3(t2, g) : T(0);
                    if (GEN && genrow) { j1 = jgen1; j2 = jgen2; }
                    if (oi >= 0) { jrow[DP + oi] = j1; jrow[2 * DP + oi] = j2; }
                    if (ci && l > dsl && l < DP) { jrow[DP + l] = T(0); jrow[2 * DP + l] = T(0); }
                }
            }
            // ... then, one point per lane: the point's velocity along its three Jacobians (a walk along its path, the joint
            // velocities through the broadcast vector) and the row parameters D, aref (cs[4:8])
            if (FRIC && ucinst != 0) {
                VEC[l] = v;
                TSYNC();
                if (l < NS && ((cinst >> l) & 1u)) {
                    const T* sp = M + T_SPH + l * TREE_SPH_STRIDE;
                    const T* jrow = X + A_JC + l * NJ * DP;
                    T* cs = X + A_CS + l * CS;
                    const int link = (int)sp[0], dsl = (int)sp[11];
                    T jv = T(0), j1v = T(0), j2v = T(0);
                    int pa[DP];
#pragma unroll
                    for (int c = 0; c < DP; ++c)
                        pa[c] = DP <= 12 ? (int)((pt_path >> (5 * c)) & 31ull) : AT[(c <= dsl ? c : 0) * PL + link];
#pragma unroll
                    for (int c = 0; c < DP; ++c) {       // (branch-free, as in point_residuals: zeros past the root)
                        const T xv = VEC[pa[c]];
                        jv += jrow[c] * xv;
                        if (FRIC) { j1v += jrow[DP + c] * xv; j2v += jrow[2 * DP + c] * xv; }
                    }
                    const T mu = FRIC ? sp[7] : T(0);
                    T Dc, arc;
                    const int kind = GEN ? (int)sp[12] : 0;
                    if (GEN && (kind == PT_CONNECT || kind == PT_WELD)) {
                        // three bilateral rows (connect: along the world axes; weld: the rotation error's components), each
                        // with its own violation, D and reference acceleration
                        const T* ex = PEXT + l * TREE_PEXT_STRIDE;
                        const T jk[3] = {jv, j1v, j2v};
                        const T pos3[3] = {kind == PT_WELD ? cs[11] : cs[0] - cs[11], kind == PT_WELD ? cs[12] : cs[1] - cs[12],
                                           kind == PT_WELD ? cs[13] : cs[2] - cs[13]};
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            tree_row_params(ex + 12, pos3[k], sp[6], jk[k], Dc, arc);
                            cs[15 + k] = Dc;
                            cs[5 + k] = arc;
                        }
                        cs[4] = cs[15];
                    } else if (GEN && kind == PT_DOFROW) {
                        const T* ex = PEXT + l * TREE_PEXT_STRIDE;
                        if (ex[0] == T(0)) tree_row_params(ex + 12, cs[3], sp[6], jv, Dc, arc);              // joint equality
                        else tree_row_params(M + T_SOLTAB + 7 * (int)sp[21], cs[3] - sp[5], sp[6], jv, Dc, arc);     // tendon / ball-joint limit
                        cs[4] = Dc;
                        cs[5] = arc;
                        cs[6] = T(0);
                        cs[7] = T(0);
                    } else {
                    const T* csol = M + T_SOLTAB + 7 * (int)sp[21];         // the contact's own solver set
                    if (GEN >= 3 && ((ell16 >> l) & 1u)) {
                        // elliptic cone (mj_instantiateContact / mj_makeImpedance): the normal row's diagApprox is tran itself,
                        // the tangent rows are pure damping - reference accelerations -B Jt.v (their R = R_normal / impratio)
                        tree_row_params(csol, cs[3] - sp[5], sp[6], jv, Dc, arc);
                        cs[4] = Dc;
                        cs[5] = arc;
                        cs[6] = -csol[1] * j1v;
                        cs[7] = -csol[1] * j2v;
                    } else {
                    tree_row_params(csol, cs[3] - sp[5], sp[6] * (T(1) + mu * mu), jv, Dc, arc);
                    if (mu > T(0)) Dc *= T(0.5) * rcp_(mu * mu);
                    cs[4] = Dc;
                    cs[5] = arc;
                    if (FRIC) { cs[6] = mu * csol[1] * j1v; cs[7] = mu * csol[1] * j2v; }
                    }
                    }
                }
            }
            TSYNC();
            const bool any_rows = !(TREE_SKIP & 1) && (__any(inst || cinst != 0) || any_floss);
            T qfrc_c = T(0);
            // mj_forward's acceleration (the constraint solver's) where the wavefront has rows: what mj_checkAcc looks at - kept as
            // the two tests' outcomes, not as a value that would stay live through the Euler solve
            bool fwd_bad = false, fwd_odd = false;
            T erow[MERGE ? DP : 1];      // factor of the Euler matrix when it was computed beside the first Newton factor
            clk.mark(3);
            clk.count(8, 1);
            clk.count(9, __popc(cinst));
            if (any_rows) {
                clk.count(10, 1);
                clk.lap(-1);
                if (__any(inst)) {      // (the impedance arithmetic only when some lane of the wavefront has a limit row)
                    tree_row_params(M + T_SOLTAB + 7 * (dofcls & 7), dist - jmargin, M[T_DOF_INVW + l], sig * v, D, aref);
                    D = inst ? D : T(0);
                    aref = inst ? aref : T(0);
                }
                // rows of contact point s with friction mu (uniform per particle): all NR, else one
                auto rows_of = [&](int s) -> unsigned {
                    if (GEN && ((int)M[T_SPH + s * TREE_SPH_STRIDE + 12] == PT_CONNECT || (int)M[T_SPH + s * TREE_SPH_STRIDE + 12] == PT_WELD)) return 7u;
                    if (GEN >= 3 && ((ell16 >> s) & 1u)) return 7u;         // (three residuals: normal, two tangents)
                    return (FRIC && M[T_SPH + s * TREE_SPH_STRIDE + 7] > T(0)) ? 15u : 1u;
                };
                // (GEN) my record's kind; bilateral records (equalities) keep all their rows active on both sides
                const int my_kind = (GEN && l < NS) ? (int)M[T_SPH + l * TREE_SPH_STRIDE + 12] : 0;
                const bool my_bil = GEN && l < NS && PEXT[(l < NS ? l : 0) * TREE_PEXT_STRIDE + 19] != T(0);
                const bool my_ell = GEN >= 3 && l < NS && ((ell16 >> l) & 1u);
                // my elliptic record: friction coefficient, impratio = D_tangent / D_normal, regularised friction mu = fr / sqrt(impratio)
                const T ell_fr = my_ell ? M[T_SPH + l * TREE_SPH_STRIDE + 7] : T(1);
                const T ell_ir = my_ell ? PEXT[l * TREE_PEXT_STRIDE + 21] : T(1);
                const T ell_mu = ell_fr * rcp_(sqrt_(ell_ir));
                // zone of the cone at residuals r (MuJoCo PrimalUpdateConstraint): U = (mu r0, fr r1, fr r2), N = U0, T = |(U1, U2)|
                auto cone_zone = [&](const T* r) -> unsigned {
                    const T N = ell_mu * r[0], Tt = ell_fr * sqrt_(r[1] * r[1] + r[2] * r[2]);
                    if (N >= ell_mu * Tt || (Tt <= T(0) && N >= T(0))) return 0u;
                    if (ell_mu * N + Tt <= T(0) || (Tt <= T(0) && N < T(0))) return 1u;
                    return 2u;
                };
                // gradient of the cone's cost at r (its zone given), and - H != nullptr - the Hessian as 00 01 02 11 12 22
                auto cone_grad = [&](unsigned zone, const T* r, T D0, T* g, T* H) {
                    const T Dt = D0 * ell_ir;
                    if (zone == 1u) {
                        g[0] = D0 * r[0]; g[1] = Dt * r[1]; g[2] = Dt * r[2];
                        if (H) { H[0] = D0; H[1] = T(0); H[2] = T(0); H[3] = Dt; H[4] = T(0); H[5] = Dt; }
                        return;
                    }
                    if (zone == 0u) {
                        g[0] = g[1] = g[2] = T(0);
                        if (H) { H[0] = H[1] = H[2] = H[3] = H[4] = H[5] = T(0); }
                        return;
                    }
                    const T mu = ell_mu, fr = ell_fr;
                    const T N = mu * r[0], U1 = fr * r[1], U2 = fr * r[2], Tt = sqrt_(U1 * U1 + U2 * U2), iT = rcp_(Tt);
                    const T Dm = D0 * rcp_(mu * mu * (T(1) + mu * mu)), NmT = N - mu * Tt;
                    const T u1 = U1 * iT, u2 = U2 * iT;
                    g[0] = Dm * NmT * mu;
                    g[1] = -Dm * NmT * mu * fr * u1;
                    g[2] = -Dm * NmT * mu * fr * u2;
                    if (H) {
                        const T a = Dm * mu * mu, bq = -Dm * NmT * mu * fr * fr * iT;
                        H[0] = a;
                        H[1] = -a * fr * u1;
                        H[2] = -a * fr * u2;
                        H[3] = a * fr * fr * u1 * u1 + bq * (T(1) - u1 * u1);
                        H[4] = a * fr * fr * u1 * u2 - bq * u1 * u2;
                        H[5] = a * fr * fr * u2 * u2 + bq * (T(1) - u2 * u2);
                    }
                };
                // the quadratic model of my elliptic record at the base point's residuals r (zone z): H += J' W J, rhs += J' b with
                // b = W (r + aref) - grad, so that the solve's equation reads M xa - tau = -J' (grad + W (r(xa) - r))
                auto cone_model = [&](unsigned zone, const T* r) {
                    if (!my_ell || !((cinst >> l) & 1u)) return;
                    T* cs = X + A_CS + l * CS;
                    T g[3], W[6];
                    cone_grad(zone, r, cs[4], g, W);
                    const T x0 = r[0] + cs[5], x1 = r[1] + cs[6], x2 = r[2] + cs[7];
#pragma unroll
                    for (int k = 0; k < 6; ++k) cs[CS_W + k] = W[k];
                    cs[CS_B] = W[0] * x0 + W[1] * x1 + W[2] * x2 - g[0];
                    cs[CS_B + 1] = W[1] * x0 + W[3] * x1 + W[4] * x2 - g[1];
                    cs[CS_B + 2] = W[2] * x0 + W[4] * x1 + W[5] * x2 - g[2];
                };
                // (GEN) my dof's friction-loss row: J = e_l, pos = 0 -> D from the impedance at 0, aref = -B v
                T Df = T(0), areff = T(0);
                int fstate = 0;                     // -1: r <= -R f (force +f), 0: quadratic zone, +1: r >= R f (force -f)
                if (GEN && any_floss) {
                    Df = Df0;
                    areff = -fB * v;
                    fstate = (fl_mem & 1) ? (fl_mem >> 1) - 1 : 0;
                    fstate = floss > T(0) ? fstate : 0;
                }
                auto fl_state_of = [&](T xa_, int cur) -> int {
                    if (!(floss > T(0))) return 0;
                    const T sl = Df * (xa_ - areff);
                    // f32: a slope within rounding of the bound keeps its zone
                    const T bc = sizeof(T) == 4 ? T(2e-5) * (fabs(sl) + floss) : T(0);
                    if (cur < 0) return sl > -floss + bc ? (sl >= floss ? 1 : 0) : -1;
                    if (cur > 0) return sl < floss - bc ? (sl <= -floss ? -1 : 0) : 1;
                    return sl <= -floss - bc ? -1 : (sl >= floss + bc ? 1 : 0);
                };
                auto fl_force = [&](T xa_, int st) -> T {        // -s'(r) of the Huber cost
                    if (!(floss > T(0))) return T(0);
                    return st == 0 ? -Df * (xa_ - areff) : (st < 0 ? floss : -floss);
                };
                // POINT-PARALLEL residuals: lane s owns contact point s.  The solution goes through the broadcast vector; the
                // owner walks the point's path-indexed Jacobians (entry c belongs to the ancestor at distance c of the
                // point's link in the elimination tree) - one pass for all points instead of three lane sums per point.
                const bool my_pt = l < NS && ((cinst >> l) & 1u);
                auto point_residuals = [&](T xa_, T* res) {
                    VEC[l] = xa_;
                    TSYNC();
                    T an = T(0), a1 = T(0), a2 = T(0);
                    if (my_pt) {
                        const T* sp = M + T_SPH + l * TREE_SPH_STRIDE;
                        const T* jrow = X + A_JC + l * NJ * DP;
                        const int link = (int)sp[0], dsl = (int)sp[11];
                        // (branch-free: entries past the root are zero in the three rows - written so by the rows phase -
                        // and the index is clamped, so that all the loads of the walk are in flight together)
                        int pa[DP];
#pragma unroll
                        for (int c = 0; c < DP; ++c)
                            pa[c] = DP <= 12 ? (int)((pt_path >> (5 * c)) & 31ull) : AT[(c <= dsl ? c : 0) * PL + link];
#pragma unroll
                        for (int c = 0; c < DP; ++c) {
                            const T xv = VEC[pa[c]];
                            an += jrow[c] * xv;
                            if (FRIC) { a1 += jrow[DP + c] * xv; a2 += jrow[2 * DP + c] * xv; }
                        }
                    }
                    const T* cs = X + A_CS + (l < NS ? l : 0) * CS;
                    res[0] = an - cs[5];
                    if (FRIC) {
                        const T mu = M[T_SPH + (l < NS ? l : 0) * TREE_SPH_STRIDE + 7];
                        const T c1 = a1, c2 = a2;
                        a1 *= mu;
                        a2 *= mu;
                        res[0] = an + a1 - (cs[5] - cs[6]);
                        res[1 % NR] = an - a1 - (cs[5] + cs[6]);
                        res[2 % NR] = an + a2 - (cs[5] - cs[7]);
                        res[3 % NR] = an - a2 - (cs[5] + cs[7]);
                        if (!(mu > T(0))) res[0] = an - cs[5];
                        if (GEN && (my_kind == PT_CONNECT || my_kind == PT_WELD || my_ell)) {         // three independent rows, one per Jacobian
                            res[0] = an - cs[5];
                            res[1 % NR] = c1 - cs[6];
                            res[2 % NR] = c2 - cs[7];
                            res[3 % NR] = T(0);
                        }
                    }
                    TSYNC();
                };
                // the rows the owners' residuals ask for, as the particle's mask in every lane (bit s * NR + r: an OR over the
                // owners' nibbles): a row stays / becomes active while its residual is negative
                auto rows_from_res = [&](const T* res, mask_t cur) -> mask_t {
                    unsigned nb = 0;
                    if (GEN >= 3 && my_pt && my_ell) {
                        nb = cone_zone(res);            // (an elliptic record's nibble: its zone)
                    } else if (my_pt) {
                        const unsigned rows = rows_of(l);
                        const T ar5 = X[A_CS + l * CS + 5];
                        const unsigned mynib = (unsigned)(cur >> (l * NR)) & ((1u << NR) - 1u);
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const T arr = res[r];
                            // f32: a row whose residual is within rounding of zero keeps its state (as in arm_rollout.hip)
                            const T bc = sizeof(T) == 4 ? T(2e-5) * (fabs(ar5) + fabs(arr + ar5) + T(1)) : T(0);
                            // (selects, not branches: four rows x three short-circuit tests compiled to a dozen exec-mask branches)
                            const bool was = ((mynib >> r) & 1u) != 0u;
                            const bool stay = !(arr > bc), come = arr < -bc;
                            const bool on = (((rows >> r) & 1u) != 0u) & (my_bil | (was ? stay : come));
                            nb |= on ? (1u << r) : 0u;
                        }
                    }
                    if constexpr (NR == 1) {
                        return (mask_t)or_lanes<PL>(nb << (l & 31));
                    } else {
                        const unsigned lo = or_lanes<PL>(l < 8 ? nb << (4 * l) : 0u);
                        const unsigned hi = or_lanes<PL>((l >= 8 && l < 16) ? nb << (4 * (l - 8)) : 0u);
                        return (mask_t)(((unsigned long long)hi << 32) | lo);
                    }
                };
                // J' f of the rows of the set (act_, cact_) at the acceleration whose owner residuals are res (friction
                // instantiation): the owners sum their rows' forces per Jacobian into cs[8:11], every dof collects its entries
                // (elliptic records, GEN = 3: `lin` - the force of the record's quadratic model, b - W (r + aref), which is what the
                // solve's equation holds at its Newton point; else the cone's own force -grad at r)
                auto force_of = [&](T xa_, const T* res, bool act_, mask_t cact_, int fst_, bool lin = false) -> T {
                    T qf = act_ ? -D * (sig * xa_ - aref) * sig : T(0);
                    if constexpr (GEN) qf += fl_force(xa_, fst_);
                    if (ucinst != 0) {
                        if (my_pt) {
                            const unsigned bits = (unsigned)(cact_ >> (l * NR)) & ((1u << NR) - 1u);
                            T* cs = X + A_CS + l * CS;
                            const T Dc = cs[4];
                            T fn = T(0), f1 = T(0), f2 = T(0);
#pragma unroll
                            for (int r = 0; r < NR; ++r) {
                                const T fr = ((bits >> r) & 1u) ? -Dc * res[r] : T(0);
                                fn += fr;
                                if (FRIC) {
                                    if (r < 2) f1 += (r & 1) ? -fr : fr;
                                    else f2 += (r & 1) ? -fr : fr;
                                }
                            }
                            const T mu = FRIC ? M[T_SPH + l * TREE_SPH_STRIDE + 7] : T(0);
                            cs[8] = fn;
                            cs[9] = mu * f1;
                            cs[10] = mu * f2;
                            if (GEN && (my_kind == PT_CONNECT || my_kind == PT_WELD)) {     // f_k = -D_k r_k on the three Jacobians
                                cs[8] = -cs[15] * res[0];
                                cs[9] = -cs[16] * res[1 % NR];
                                cs[10] = -cs[17] * res[2 % NR];
                            }
                            if (GEN >= 3 && my_ell) {
                                if (lin) {
                                    const T x0 = res[0] + cs[5], x1 = res[1 % NR] + cs[6], x2 = res[2 % NR] + cs[7];
                                    const bool on = bits != 0u;
                                    const T* W = cs + CS_W;
                                    cs[8] = on ? cs[CS_B] - (W[0] * x0 + W[1] * x1 + W[2] * x2) : T(0);
                                    cs[9] = on ? cs[CS_B + 1] - (W[1] * x0 + W[3] * x1 + W[4] * x2) : T(0);
                                    cs[10] = on ? cs[CS_B + 2] - (W[2] * x0 + W[4] * x1 + W[5] * x2) : T(0);
                                } else {
                                    T g[3];
                                    cone_grad(cone_zone(res), res, Dc, g, (T*)nullptr);
                                    cs[8] = -g[0];
                                    cs[9] = -g[1];
                                    cs[10] = -g[2];
                                }
                            }
                        }
                        TSYNC();
                        for (unsigned um = ucinst; um; um &= um - 1) {
                            const int s = __builtin_ctz(um);
                            const unsigned bits = (unsigned)(cact_ >> (s * NR)) & ((1u << NR) - 1u);
                            if (!__any(bits != 0)) continue;
                            const int oi = own_idx(s);
                            if (oi >= 0 && bits) {
                                const T* jrow = X + A_JC + s * NJ * DP;
                                const T* cs = X + A_CS + s * CS;
                                qf += jrow[oi] * cs[8];
                                if (FRIC) qf += jrow[DP + oi] * cs[9] + jrow[2 * DP + oi] * cs[10];
                            }
                        }
                        TSYNC();
                    }
                    return qf;
                };
                // the active set a solution belongs to next
                auto next_set = [&](T xa_, mask_t cur) -> mask_t {
                    if constexpr (!FRIC) {
                        mask_t nxt = 0;
                        for (unsigned um = ucinst; um; um &= um - 1) {
                            const int s = __builtin_ctz(um);
                            const int oi = own_idx(s);
                            const T ar5 = X[A_CS + s * CS + 5];
                            const T arr = sum_lanes<PL>(oi >= 0 ? X[A_JC + s * NJ * DP + oi] * xa_ : T(0)) - ar5;
                            const T bc = sizeof(T) == 4 ? T(2e-5) * (fabs(ar5) + fabs(arr + ar5) + T(1)) : T(0);
                            const bool was = (cur >> s) & 1u;
                            if (((cinst >> s) & 1u) && (was ? !(arr > bc) : (arr < -bc))) nxt |= mask_t(1) << s;
                        }
                        return nxt;
                    }
                    if (ucinst == 0) return mask_t(0);
                    T res[NR];
                    point_residuals(xa_, res);
                    return rows_from_res(res, cur);
                };
                // initial active set: a row that existed in the previous substep keeps its state, a new row is active
                bool actv = inst && ((lim_mem & 1) ? (lim_mem & 2) != 0 : true);
                mask_t cact = 0;
                for (unsigned um = ucinst; um; um &= um - 1) {
                    const int s = __builtin_ctz(um);
                    if ((cinst >> s) & 1u) {
                        const mask_t rows = rows_of(s);
                        if (GEN >= 3 && ((ell16 >> s) & 1u)) {
                            // an elliptic record starts in the bottom zone (three quadratic rows: no base point needed) unless
                            // it was in the top zone (no force) a substep ago
                            const bool was_top = ((cinst_mem >> s) & 1u) && ((cact_mem >> (s * NR)) & mask_t(15)) == 0;
                            cact |= was_top ? mask_t(0) : (mask_t(1) << (s * NR));
                        } else
                        cact |= ((cinst_mem >> s) & 1u) ? (cact_mem & (rows << (s * NR))) : (rows << (s * NR));
                    }
                }
#ifdef TREE_WARM_START
                // developer A/B (round 6, measured and NOT kept: profiles/r06_warm_start_ab.txt - HalfCheetah 2.02 -> 2.045 ms per
                // 4096 x 32 launch, tray 2.755 -> 2.79, f32 32768 x 32 8.40 -> 8.565: the walk costs what the saved re-iterations
                // were worth).  MuJoCo's warm start for the rows that are NEW this substep: instead of "a new row is active", a new
                // row is active if its residual at the PREVIOUS substep's acceleration is not positive (mj_fwdConstraint starts
                // from qacc_warmstart and takes the rows that are violated there) - one walk of the owners' paths in the
                // substeps where a point comes into contact, against a re-iteration when a new pyramid's four rows do not
                // all end up active (the usual case: a sliding contact holds two or three)
                if constexpr (FRIC && GEN < 3) {
                    const unsigned newpts = cinst & ~cinst_mem;
                    if (__any(newpts != 0u)) {
                        T r0[NR];
#if TREE_WARM_START == 2        // ... at ZERO acceleration: the rows whose reference acceleration is positive - no walk (the owner's record)
                        {
                            const T* cs0 = X + A_CS + (l < NS ? l : 0) * CS;
                            const T mu0 = M[T_SPH + (l < NS ? l : 0) * TREE_SPH_STRIDE + 7];
                            r0[0] = -(cs0[5] - cs0[6]);
                            r0[1 % NR] = -(cs0[5] + cs0[6]);
                            r0[2 % NR] = -(cs0[5] - cs0[7]);
                            r0[3 % NR] = -(cs0[5] + cs0[7]);
                            if (!(mu0 > T(0))) r0[0] = -cs0[5];
                            if (GEN && (my_kind == PT_CONNECT || my_kind == PT_WELD)) { r0[0] = -cs0[5]; r0[1 % NR] = -cs0[6]; r0[2 % NR] = -cs0[7]; r0[3 % NR] = T(0); }
                        }
#else
                        point_residuals(xa_prev, r0);
#endif
                        const mask_t pred = rows_from_res(r0, cact);
                        unsigned long long x = newpts;          // bit s -> bit 4 s
                        x = (x | (x << 24)) & 0x000000ff000000ffull;
                        x = (x | (x << 12)) & 0x000f000f000f000full;
                        x = (x | (x << 6)) & 0x0303030303030303ull;
                        x = (x | (x << 3)) & 0x1111111111111111ull;
                        const mask_t nm = (mask_t)(x * 15ull);
                        cact = (cact & ~nm) | (pred & nm);
                    }
#if TREE_WARM_START != 2
                    if (inst && !(lim_mem & 1)) actv = !(sig * xa_prev - aref > T(0));
#endif
                }
#endif
                if constexpr (GEN >= 3) {
                    if (my_ell && my_pt) {
                        const T r0[3] = {T(0), T(0), T(0)};
                        cone_model(1u, r0);         // (bottom zone: W = diag(D), b = W aref - whatever the point)
                    }
                    TSYNC();
                }
                bool changed = true, act_pp = false;
                mask_t cact_pp = 0;
                int fst_pp = 0;
                // (round 6: models with FRICTION-LOSS rows from the third iteration on - a Huber row's zone changes drag each other
                // along and the plain iteration wandered to iteration 5 before the search took over; measured per 4096 x 32 launch:
                // door 1.02 -> 0.935 ms, pen-in-hand with dry finger joints 22.1 -> 18.35, cart-pole 0.415 -> 0.40; a model with
                // pyramids only LOSES by an earlier search - tray 2.755 -> 2.975 - and keeps 5; profiles/r06_ls_start_ab.txt)
                // (... and, once the search found its root by false position: from the SECOND iteration on in f64 - door 0.885 -> 0.86,
                // dry-jointed pen-in-hand 17.6 -> 16.5; f32 keeps the third: door 0.715 -> 0.775 with the second; r06_ls_start2_ab.txt)
#ifndef TREE_LS_START_FLOSS
#define TREE_LS_START_FLOSS (sizeof(T) == 8 ? 1 : 2)
#endif
#ifndef TREE_LS_START_PYR
#define TREE_LS_START_PYR 5
#endif
                const int LS_START = GEN >= 3 ? 0 : ((GEN && any_floss) ? TREE_LS_START_FLOSS : TREE_LS_START_PYR);      // iterations before the safeguard takes over (friction instantiation; elliptic cones: MuJoCo's Newton method from the start)
                bool ls_on = false;
                T a_b = T(0), g_b = T(0), rb[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) rb[r] = T(0);
                T xa = T(0);
                // Round 6: every decision of the iteration is taken PER PARTICLE.  A particle whose sets reproduce themselves is
                // DONE: its solution and sets are frozen (restored where an iteration ends) while the wavefront iterates on for
                // its mates, and whether a re-iteration is a rank-one correction is a matter of the particle's own changes - so a
                // particle's trajectory is the same bits whoever shares its wavefront (P = 1, the device-resident real env, and
                // the planner's copy of that particle; tests/test_locomotion_gpu.py).
                bool pdone = false;
                // (... and so are the owner residuals at that solution: the constraint force after the loop is made from them
                // without another walk of the points' paths - unless a line-search step, whose residuals are interpolated, or the
                // iteration cap ended a particle of the wavefront: rk_ok)
                T rN_k[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) rN_k[r] = T(0);
                bool rN_ok = false, rk_ok = false;
                T xa_k = T(0);
                bool actv_k = false;
                mask_t cact_k = 0;
                int fst_k = 0;
                auto mine_any = [&](bool x) -> bool { return (__ballot(x) & my_lanes) != 0ull; };
                TREE_FLAP(31);
                clk.lap(-1);
                for (int it = 0; it < (FRIC ? TREE_MAXIT_LS : TREE_MAXIT); ++it) {
                    if ((TREE_SKIP & 16) || ((TREE_SKIP & 32) && it == 1)) { changed = false; break; }      // (developer timing: no iteration / one)
                    T hrow[DP];
                    T hd[DN > 0 ? DN : 1], hdinv = T(1);        // DN > 0: my dense row of H, then of its factor
                    const T Dfq = (GEN && fstate == 0) ? Df : T(0);     // the friction-loss row in its quadratic zone
                    if constexpr (DN > 0) {
#pragma unroll
                        for (int j = 0; j < DN; ++j) hd[j] = md[j] + ((j == l && actv) ? D : T(0)) + (j == l ? Dfq : T(0));
                    } else {
#pragma unroll
                        for (int c = 0; c < DP; ++c) hrow[c] = mrow[c];
                        hrow[0] += (actv ? D : T(0)) + Dfq;
                    }
                    T rhs = tau + (actv ? D * sig * aref : T(0));
                    if constexpr (GEN) rhs += fstate == 0 ? Dfq * areff : (fstate < 0 ? floss : -floss);
                    TREE_FLAP(24);
                    for (unsigned um = ucinst; um; um &= um - 1) {
                        const int s = __builtin_ctz(um);
                        const unsigned bits = (unsigned)(cact >> (s * NR)) & ((1u << NR) - 1u);
                        if (!__any(bits != 0)) continue;
                        const T* cs = X + A_CS + s * CS;
                        const T* jrow = X + A_JC + s * NJ * DP;
                        const int oi = own_idx(s);
                        const T Dc = bits ? cs[4] : T(0), jl = oi >= 0 ? jrow[oi] : T(0);
                        // sum over the active rows r of D J_r J_r' and D J_r aref_r, J_r = Jn + s_r mu Jt_k(r):
                        // grouped by the three Jacobians kept per point (counts and signed counts of the active rows)
                        const T na = T(__popc(bits));
                        T wn = Dc * na * jl, w1 = T(0), w2 = T(0);
                        T rsum = na * cs[5];
                        T t1l = T(0), t2l = T(0);           // my entries of the two tangent Jacobians
                        const bool ells = GEN >= 3 && ((ell16 >> s) & 1u);
                        if (FRIC) {
                            t1l = oi >= 0 ? jrow[DP + oi] : T(0);
                            t2l = oi >= 0 ? jrow[2 * DP + oi] : T(0);
                        }
                        if (ells) {
                            // the cone's quadratic model at the base point (cone_model): H += J' W J, rhs += J' b
                            const T on = bits ? T(1) : T(0);
                            const T* W = cs + CS_W;
                            wn = on * (W[0] * jl + W[1] * t1l + W[2] * t2l);
                            w1 = on * (W[1] * jl + W[3] * t1l + W[4] * t2l);
                            w2 = on * (W[2] * jl + W[4] * t1l + W[5] * t2l);
                            rhs += on * (jl * cs[CS_B] + t1l * cs[CS_B + 1] + t2l * cs[CS_B + 2]);
                        } else if (FRIC) {
                            const T mu = M[T_SPH + s * TREE_SPH_STRIDE + 7];
                            const T j1 = mu * t1l, j2 = mu * t2l;
                            const T n1 = T(__popc(bits & 3u)), s1 = T((int)(bits & 1u) - (int)((bits >> 1) & 1u));
                            const T n2 = T(__popc(bits & 12u)), s2 = T((int)((bits >> 2) & 1u) - (int)((bits >> 3) & 1u));
                            wn += Dc * (s1 * j1 + s2 * j2);
                            w1 = Dc * mu * (s1 * jl + n1 * j1);
                            w2 = Dc * mu * (s2 * jl + n2 * j2);
                            rsum -= s1 * cs[6] + s2 * cs[7];
                            rhs += Dc * (j1 * (s1 * cs[5] - n1 * cs[6]) + j2 * (s2 * cs[5] - n2 * cs[7]));
                        }
                        if (ells) {
                        } else if (GEN && ((int)M[T_SPH + s * TREE_SPH_STRIDE + 12] == PT_CONNECT || (int)M[T_SPH + s * TREE_SPH_STRIDE + 12] == PT_WELD)) {
                            // three independent bilateral rows: H += sum_k D_k J_k J_k', rhs += sum_k D_k aref_k J_k
                            wn = cs[15] * jl;
                            w1 = cs[16] * t1l;
                            w2 = cs[17] * t2l;
                            rhs += w1 * cs[6] + w2 * cs[7];
                            rsum = cs[5];
                            rhs += wn * rsum;
                        } else
                        rhs += Dc * jl * rsum;
                        if constexpr (DN > 0) {
                            // dense row: H[l][j] += wn Jn[j] + w1 Jt1[j] + w2 Jt2[j], lane j's entries by DPP broadcast
                            if constexpr (DN == 32) {
                                T je[3], jo[3];
                                row_pair(jl, je[0], jo[0]);
                                if (FRIC) { row_pair(t1l, je[1], jo[1]); row_pair(t2l, je[2], jo[2]); }
                                dense32_contact<0, FRIC>(hd, je, jo, wn, w1, w2);
                            } else
                            dense_contact<0, DN, FRIC>(hd, jl, t1l, t2l, wn, w1, w2);
                        } else if (oi >= 0) {
                            // a contact row couples only dofs on one path: the pattern holds, and my ancestor at distance c
                            // sits c entries further along the point's rows (zeros past the root)
#pragma unroll
                            for (int c = 0; c < DP; ++c) {
                                if (oi + c < DP) {
                                    T acc = wn * jrow[oi + c];
                                    if (FRIC) acc += w1 * jrow[DP + oi + c] + w2 * jrow[2 * DP + oi + c];
                                    hrow[c] += acc;
                                }
                            }
                        }
                    }
                    TSYNC();
                    clk.lap(12);
                    if constexpr (DN > 0) {
                        dense_factor_any<DN>(hd, hdinv, l, X + A_ROW);
                    } else if (MERGE && it == 0 && !(TREE_SKIP & 2)) {      // ... and the Euler matrix M + h B rides along (consumed in step 6)
#pragma unroll
                        for (int c = 0; c < DP; ++c) erow[c] = mrow[c];
                        erow[0] += dof ? h * damping : T(0);
                        tree_factor2<DP, PL>(hrow, erow, ELIM, ROW, ROW2, l, n_rounds, kt, depth);
                    } else {
                        tree_factor<DP, PL>(hrow, ELIM, ROW, l, n_rounds, kt, depth);
                    }
                    clk.lap(13);
                    if constexpr (DN > 0) xa = dense_solve_any<DN>(hd, hdinv, rhs, l);
                    else xa = tree_solve<DP, PL>(hrow, rhs, ELIM, AT, ROW, VEC, l, n_rounds, depth, max_depth, kt);
                    clk.lap(14);
                    // f32: a row whose residual is within rounding of zero keeps its state (as in arm_rollout.hip)
                    const T resl = sig * xa - aref;
                    const T band = sizeof(T) == 4 ? T(2e-5) * (fabs(aref) + fabs(xa) + T(1)) : T(0);
                    bool act2 = inst && (actv ? !(resl > band) : (resl < -band));
                    T rN[NR];                                   // friction instantiation: my point's residuals at xa
                    mask_t cact2;
                    rN_ok = true;                   // (rN: the owner residuals at xa)
                    if constexpr (FRIC) {
                        if (ucinst != 0) {              // (wave-uniform: no point or record this substep - limit / friction-loss rows only)
                            point_residuals(xa, rN);
                            TREE_LAP_A;
                            cact2 = rows_from_res(rN, cact);
                        } else {
#pragma unroll
                            for (int r = 0; r < NR; ++r) rN[r] = T(0);
                            cact2 = 0;
                        }
                    } else {
                        cact2 = next_set(xa, cact);
                    }
                    int fst2 = 0;
                    if constexpr (GEN) fst2 = fl_state_of(xa, fstate);
                    changed = (act2 != actv) || (cact2 != cact) || (GEN && fst2 != fstate);
                    // (GEN = 3: an elliptic record of the model's set in its middle zone - the cost is not quadratic there, the
                    // Newton point of the model is not the minimiser: on, until the line search below stops moving)
                    if constexpr (GEN >= 3) changed = changed || (cact & (mask_t)ell_mid) != 0;
                    changed = changed && !pdone;        // (a frozen particle asks for nothing: no search, no correction on its account)
                    // SAFEGUARD (friction instantiation).  The plain iteration - solve with the set, adopt the set the solution
                    // asks for - has no line search and can cycle when several friction pyramids switch rows together
                    // (periods 3 and 4 seen on the pen-in-hand model; the iterate kept then was arbitrary).  From iteration
                    // LS_START on it becomes MuJoCo's Newton method: a base point a_b with its set, the Newton point xa of
                    // that set, and an EXACT line search on the true (piecewise quadratic, convex) objective between them:
                    //   phi'(al) = p.(M a - tau) + sum_rows D_r min(0, r_r(a)) (J_r p),  a = a_b + al p,  p = xa - a_b,
                    // where M a - tau is linear in al and known at both ends without a product with M: at a Newton point
                    // it equals the constraint force J' f of the rows of its set (the solve's own equation), and the base
                    // point inherits it by the same interpolation; the residuals are affine in al.
                    TREE_LAP_B;
                    if constexpr (FRIC) {
                        if (it >= LS_START && __any(changed)) {
                            const T gN = force_of(xa, rN, actv, cact, fstate, true);      // M xa - tau = J' f of the set's rows (the solve's equation)
                            if (!ls_on) {
                                a_b = xa;
                                g_b = gN;
#pragma unroll
                                for (int r = 0; r < NR; ++r) rb[r] = rN[r];
                                ls_on = true;
                            } else {
                                const unsigned long long fb = __ballot(changed);
                                const bool pch = ((unsigned)(fb >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu)) != 0u;
                                if (pch) {      // (a particle whose Newton point reproduces its set has converged: full step)
                                    const T pv = xa - a_b;
                                    const T gbp = sum_lanes<PL>(g_b * pv), gNp = sum_lanes<PL>(gN * pv);
                                    const T rl0 = sig * a_b - aref, rl1 = sig * xa - aref;
                                    const T Dc = my_pt ? X[A_CS + l * CS + 4] : T(0);
                                    const unsigned rows = my_pt ? rows_of(l) : 0u;
                                    T drb[NR];
#pragma unroll
                                    for (int r = 0; r < NR; ++r) drb[r] = ((rows >> r) & 1u) ? rN[r] - rb[r] : T(0);
                                    T al;
                                    bool at_floor = false;
                                    if constexpr (GEN >= 3) {
                                        const T* Dk = (my_pt && (my_kind == PT_CONNECT || my_kind == PT_WELD)) ? X + A_CS + l * CS + 15 : nullptr;
                                        T Dk3[NR];
#pragma unroll
                                        for (int r = 0; r < NR; ++r) Dk3[r] = (Dk && r < 3) ? Dk[r] : Dc;
                                        al = cone_line_search<PL, NR>(gbp, gNp - gbp, D, rl0, rl1 - rl0, Dc, rb[0], rb[1 % NR], rb[2 % NR], rb[3 % NR],
                                                                      drb[0], drb[1 % NR], drb[2 % NR], drb[3 % NR], Dk3[0], Dk3[1 % NR], Dk3[2 % NR], Dk3[3 % NR],
                                                                      my_bil, Df, floss, a_b - areff, pv, my_ell && my_pt, ell_fr, ell_ir, ell_mu);
                                        if (al < T(0)) { at_floor = true; al = T(1); }
                                    } else if constexpr (GEN) {
                                        const T* Dk = (my_pt && (my_kind == PT_CONNECT || my_kind == PT_WELD)) ? X + A_CS + l * CS + 15 : nullptr;
                                        T Dk3[NR];
#pragma unroll
                                        for (int r = 0; r < NR; ++r) Dk3[r] = (Dk && r < 3) ? Dk[r] : Dc;
                                        al = exact_line_search<PL, NR, true>(gbp, gNp - gbp, D, rl0, rl1 - rl0, Dc, rb[0], rb[1 % NR], rb[2 % NR], rb[3 % NR],
                                                                             drb[0], drb[1 % NR], drb[2 % NR], drb[3 % NR], Dk3[0], Dk3[1 % NR], Dk3[2 % NR],
                                                                             Dk3[3 % NR], my_bil, Df, floss, a_b - areff, pv);
                                    } else {
                                        al = exact_line_search<PL, NR, false>(gbp, gNp - gbp, D, rl0, rl1 - rl0, Dc, rb[0], rb[1 % NR], rb[2 % NR], rb[3 % NR],
                                                                              drb[0], drb[1 % NR], drb[2 % NR], drb[3 % NR], Dc, Dc, Dc, Dc, false,
                                                                              T(0), T(0), T(0), T(0));
                                    }
                                    // a step below the working precision of the iterate (a base point on a kink of the
                                    // objective, where a row at zero residual may be counted either way): the safeguarded
                                    // iteration has reached its fixed point - mj_solNewton stops likewise once the
                                    // improvement falls under its tolerance
                                    // (GEN = 3: against the particle's largest acceleration - with cones in their middle zone the iteration
                                    // goes on until it stops moving, and a dof that hardly accelerates keeps receiving the
                                    // rounding of the others' solve: steps of 1e-13 on accelerations of 300 went on to the cap)
                                    T a_scale = fabs(a_b);
                                    if constexpr (GEN >= 3) a_scale = sqrt_(sum_lanes<PL>(a_b * a_b));
                                    const bool moved = !at_floor && fabs(al * pv) > (sizeof(T) == 4 ? T(1e-6) : (GEN >= 3 ? T(1e-13) : T(1e-14))) * (a_scale + T(1));
                                    const bool pmoved = ((unsigned)(__ballot(moved) >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu)) != 0u;
                                    a_b += al * pv;
                                    g_b += al * (gN - g_b);
#pragma unroll
                                    for (int r = 0; r < NR; ++r) rb[r] += al * (rN[r] - rb[r]);
                                    xa = a_b;                   // the iterate: what is kept if the iterations run out
                                    rN_ok = false;              // (rN belongs to the Newton point, rb is interpolated)
                                    const T rlb = sig * a_b - aref;
                                    act2 = inst && (rlb < T(0));
                                    cact2 = rows_from_res(rb, cact);
                                    if constexpr (GEN) fst2 = fl_state_of(xa, 0);
                                    changed = pmoved;
#ifdef TREE_DEBUG_CAP
                                    if (it >= TREE_MAXIT_LS - 6) {
                                        const unsigned am = (unsigned)(__ballot(actv) >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu);
                                        const unsigned am2 = (unsigned)(__ballot(act2) >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu);
                                        const T pn = sum_lanes<PL>(pv * pv), an = sum_lanes<PL>(a_b * a_b);
                                        if (l == 0) printf("cap pid=%ld t=%d sub=%d it=%d al=%.17g |p|=%.3g |a|=%.3g gbp=%.3g gNp=%.3g act=%x->%x cact=%llx->%llx\n",
                                                           (long)pid, t, sub, it, (double)al, (double)sqrt_(pn), (double)sqrt_(an), (double)gbp, (double)gNp,
                                                           am, am2, (unsigned long long)cact, (unsigned long long)cact2);
                                    }
#endif
                                } else {
                                    a_b = xa;
                                    g_b = gN;
#pragma unroll
                                    for (int r = 0; r < NR; ++r) rb[r] = rN[r];
                                }
                            }
                            if constexpr (GEN >= 3) {
                                // the next model of my elliptic record: its zone and quadratic expansion at the base point
                                if (my_ell && my_pt) cone_model((unsigned)(cact2 >> (l * NR)) & 15u, rb);
                                TSYNC();
                            }
                        }
                    }
                    TREE_LAP_C;
                    // One limit row j of a particle changed state (the usual reason for another iteration):
                    // H' = H + c e_j e_j', c = +-D_j, rhs' = rhs + c sig_j aref_j e_j.  With z = H^-1 e_j - one more pair of
                    // triangular solves with the factor at hand, a third of a factor + solve - Sherman-Morrison gives
                    // a' = y - c z y_j / (1 + c z_j),  y = a + (c sig_j aref_j) z.  Several flips in one particle take the
                    // general path (next iteration refactors).  Which of the two a particle's change takes is ITS OWN matter
                    // (round 6; until then the wave's: one particle with several flips sent its mates down the general path,
                    // and a particle's rounding depended on them): a wavefront with both kinds pays the extra solve AND the
                    // next factorisation (measured, 4096 x 32, round 4: cheetah 2.28 -> 2.21 ms, tray 3.52 -> 3.35,
                    // pen-in-hand f64 17.0 -> 15.9, f32 16.2 -> 13.7 ms and no iteration-cap hits where there were 16).
                    if (__any(changed)) {
                        const bool flip = act2 != actv;
                        const unsigned nflip = __popcll(__ballot(flip) & my_lanes);
#ifdef TREE_STATS
                        {   // what kind of change asks for another iteration (particle 0)
                            const unsigned ncf = __popcll((unsigned long long)(cact2 ^ cact));
                            clk.count(16, 1);
                            clk.count(17, nflip == 1u && ncf == 0u);
                            clk.count(18, nflip == 0u && ncf == 1u);
                            clk.count(19, nflip + ncf > 1u);
                            clk.count(20, nflip == 0u && ncf == 0u && !(GEN && fst2 != fstate));     // only the other particle of the wave changed
                            clk.count(21, GEN && fst2 != fstate);        // my friction-loss rows changed zone
                        }
#endif
                        // ... or ONE contact row r of a point s (J = Jn +- mu Jt_k, c = +-D_s, its own aref): the same
                        // correction with z = H^-1 J' and lane sums for J z and J a
                        const mask_t cdiff = FRIC ? (cact2 ^ cact) : mask_t(0);
                        const unsigned ncf = FRIC ? (unsigned)__popcll((unsigned long long)cdiff) : (cact2 != cact ? 2u : 0u);
                        // (measured and not kept, round 4: the same correction for ONE friction-loss row changing its zone - the
                        // door model's iterations per substep 2.48 -> 2.09, the cart-pole's 1.77 -> 1.45, and both launches 4-6 %
                        // SLOWER: three corrections in four are followed by a refactorisation anyway, a zone change drags other
                        // rows along)
                        bool fchg = false;
                        if constexpr (GEN) fchg = mine_any(fst2 != fstate);
                        // (the 16-lane dense instantiations refactor instead: their factorisation + solve is ~1.8 k ticks, the
                        // correction's solve + second walk ~1.7 k, and with per-particle decisions a wavefront that holds a
                        // one-change AND a several-changes particle pays both - closed-loop HalfCheetah 1.957 -> 1.871 ms per
                        // step without it, the others within 0.4 %: profiles/r06_rank_one_ab.txt; the 32-lane dense and the
                        // tree-sparse instantiations, whose factorisations cost 9 - 12 k, keep the correction)
#ifndef TREE_RANK1_OFF_DN
#define TREE_RANK1_OFF_DN 16
#endif
                        constexpr bool RANK1 = !(DN > 0 && DN <= TREE_RANK1_OFF_DN);
                        const bool single = RANK1 && !pdone && nflip + ncf == 1u && !fchg;        // (uniform over my particle)
                        if (!(FRIC && it >= LS_START) && __any(single)) {
                            T jz_ = (single && flip) ? T(1) : T(0);         // my entry of the changed row
                            T cc = T(0), ar = T(0);
                            if (FRIC && single && ncf == 1u) {
                                const int bit = __builtin_ctzll((unsigned long long)cdiff), s = bit / NR, r = bit - s * NR;
                                const T* cs = X + A_CS + s * CS;
                                const T* jrow = X + A_JC + s * NJ * DP;
                                const int oi = own_idx(s), k = 1 + (r >> 1);
                                const T mu = M[T_SPH + s * TREE_SPH_STRIDE + 7];
                                const T sgn = (r & 1) ? T(-1) : T(1);
                                jz_ = oi >= 0 ? jrow[oi] + (mu > T(0) ? sgn * mu * jrow[k * DP + oi] : T(0)) : T(0);
                                cc = ((cact2 >> bit) & 1u) ? cs[4] : -cs[4];
                                ar = mu > T(0) ? cs[5] - sgn * cs[5 + k] : cs[5];
                            }
                            T zl;
                            if constexpr (DN > 0) zl = dense_solve_any<DN>(hd, hdinv, jz_, l);
                            else zl = tree_solve<DP, PL>(hrow, jz_, ELIM, AT, ROW, VEC, l, n_rounds, depth, max_depth, kt);
                            TREE_FLAP(27);
                            // (one formula for either kind of row: the lane sums have ONE non-zero term, they are exact)
                            const bool fl1 = single && flip;
                            const T cp = sum_lanes<PL>(fl1 ? (act2 ? D : -D) : T(0)) + cc;
                            const T arp = sum_lanes<PL>(fl1 ? sig * aref : T(0)) + ar;
                            const T jz = sum_lanes<PL>(jz_ * zl), ja = sum_lanes<PL>(jz_ * xa);
                            const T dl = cp * arp, yj = ja + dl * jz;
                            const T xa_c = (xa + dl * zl) - cp * zl * yj * rcp_(T(1) + cp * jz);
                            if (single) {               // the corrected solution and the sets it belongs to
                                xa = xa_c;
                                if (FRIC && ncf == 1u) cact = cact2;
                                actv = act2;
                            }
                            TREE_FLAP(28);
                            // (every lane walks the exchange; a particle that took no correction arrives at the sets it had)
                            const T resl2 = sig * xa - aref;
                            const T band2 = sizeof(T) == 4 ? T(2e-5) * (fabs(aref) + fabs(xa) + T(1)) : T(0);
                            const bool act2c = inst && (actv ? !(resl2 > band2) : (resl2 < -band2));
                            mask_t cact2c = 0;
                            if constexpr (FRIC) {
                                T rC[NR];
#pragma unroll
                                for (int r = 0; r < NR; ++r) rC[r] = T(0);
                                if (ucinst != 0) {
                                    point_residuals(xa, rC);
                                    cact2c = rows_from_res(rC, cact);
                                }
                                if (single) {
#pragma unroll
                                    for (int r = 0; r < NR; ++r) rN[r] = rC[r];
                                }
                            } else {
                                cact2c = next_set(xa, cact);
                            }
                            int fst2c = 0;
                            if constexpr (GEN) fst2c = fl_state_of(xa, fstate);
                            if (single) {
                                act2 = act2c;
                                cact2 = cact2c;
                                fst2 = fst2c;
                                changed = (act2 != actv) || (cact2 != cact) || (GEN && fst2 != fstate);
                            }
                            TREE_FLAP(29);
                        }
                    }
                    clk.lap(23);
                    // f32 only: with accelerations of 1e4 rad/s^2 on gram-sized finger links a row can sit within
                    // rounding of its switching point and flip back and forth; a particle whose set returns to the one
                    // of two iterations ago has converged to working precision (either set gives the same forces)
                    if (sizeof(T) == 4 && it >= 2 && act2 == act_pp && cact2 == cact_pp && (!GEN || fst2 == fst_pp)) changed = false;
                    // a particle whose accelerations have left the range of the arithmetic (not finite, or beyond 1e100 / 1e30
                    // rad/s^2, whose squares overflow) has diverged - its cost becomes +inf and the update gives it no weight:
                    // nothing to iterate on - counted apart from the solver's own failures below
                    if (FRIC && it >= LS_START) {
                        const unsigned long long nf = __ballot(!(fabs(xa) < T(sizeof(T) == 4 ? 1e30 : 1e100)));
                        if (((unsigned)(nf >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu)) != 0u) changed = false;
                    }
                    act_pp = actv;
                    cact_pp = cact;
                    fst_pp = fstate;
                    actv = act2;
                    cact = cact2;
                    fstate = fst2;
                    // a particle none of whose lanes asks for another iteration is done: what it holds now is what it keeps,
                    // whatever its lanes compute while the wavefront iterates on for its mates
                    {
                        const bool pchg = mine_any(changed);
                        if (pdone) {
                            xa = xa_k; actv = actv_k; cact = cact_k; fstate = fst_k;
                            changed = false;
                        } else if (!pchg) {
                            pdone = true;
                            xa_k = xa; actv_k = actv; cact_k = cact; fst_k = fstate;
                            rk_ok = rN_ok;
#pragma unroll
                            for (int r = 0; r < NR; ++r) rN_k[r] = rN[r];
                        }
                    }
                    clk.count(11, 1);
                    clk.lap(15);
                    if (!__any(changed)) break;
                }
                clk.lap(-1);
                if (diag) {        // one count per particle-substep whose rows were still changing when the iterations ran out
                    const unsigned long long cb = __ballot(changed);
                    const unsigned mine = (unsigned)(cb >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu);
                    if (l == 0 && mine != 0u) atomicAdd(diag, 1u);
                }
                xa_prev = xa;
                lim_mem = (inst ? 1 : 0) | (actv ? 2 : 0);
                if constexpr (GEN) fl_mem = floss > T(0) ? (1 | ((fstate + 1) << 1)) : 0;
                cinst_mem = cinst;
                cact_mem = cact;
                fwd_bad = !(fabs(xa) <= MJ_MAXVAL);
                fwd_odd = !(fabs(h * xa) <= T(1e5));
                qfrc_c = actv ? -D * (sig * xa - aref) * sig : T(0);
                if constexpr (!FRIC) {
                    for (unsigned um = ucinst; um; um &= um - 1) {
                        const int s = __builtin_ctz(um);
                        if (!__any((cact >> s) & 1u)) continue;
                        const int oi = own_idx(s);
                        const T jl = oi >= 0 ? X[A_JC + s * NJ * DP + oi] : T(0);
                        const T arr = sum_lanes<PL>(jl * xa) - X[A_CS + s * CS + 5];
                        qfrc_c += ((cact >> s) & 1u) ? -X[A_CS + s * CS + 4] * arr * jl : T(0);
                    }
                } else {
                    T res[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) res[r] = rN_k[r];
                    // (a particle the cap stopped, or whose last step was a line search: the wavefront walks the paths once more)
                    if (ucinst != 0 && __any(!(pdone && rk_ok))) point_residuals(xa, res);
                    qfrc_c = force_of(xa, res, actv, cact, fstate);
                }
                TREE_FLAP(30);
            } else {
                lim_mem = 0;
                fl_mem = 0;
                cinst_mem = 0;
                cact_mem = 0;
            }
            // ---- 6. mj_Euler with implicit joint damping: (M + h B) qacc = qfrc_smooth + qfrc_constraint
            clk.mark(4);
            T qacc;
            {
                if (TREE_SKIP & 2) {
                    qacc = (tau + qfrc_c) * rcp_(mrow[0] + (dof ? h * damping : T(0)));
                } else if (MERGE && any_rows) {
                    qacc = tree_solve<DP, PL>(erow, tau + qfrc_c, ELIM, AT, ROW2, VEC, l, n_rounds, depth, max_depth, kt);
                } else if constexpr (DN > 0) {
                    T ed[DN > 0 ? DN : 1], edinv;
#pragma unroll
                    for (int j = 0; j < DN; ++j) ed[j] = md[j] + ((j == l && dof) ? h * damping : T(0));
                    dense_factor_any<DN>(ed, edinv, l, X + A_ROW);
                    qacc = dense_solve_any<DN>(ed, edinv, tau + qfrc_c, l);
                } else {
                    mrow[0] += dof ? h * damping : T(0);
                    tree_factor<DP, PL>(mrow, ELIM, ROW, l, n_rounds, kt, depth);
                    qacc = tree_solve<DP, PL>(mrow, tau + qfrc_c, ELIM, AT, ROW, VEC, l, n_rounds, depth, max_depth, kt);
                }
            }
            clk.mark(5);
            // mj_checkAcc: a NaN or an entry beyond mjMAXVAL in the acceleration mj_forward arrived at - the constraint solver's
            // where the wavefront had rows (for a particle without rows of its own that is M^-1 qfrc_smooth, as in MuJoCo),
            // else the Euler solve's (M + h B)^-1 qfrc_smooth, which stands in for M^-1 qfrc_smooth
            const bool acc_bad = rst && dof && (any_rows ? fwd_bad : !(fabs(qacc) <= MJ_MAXVAL));
            if (!rst && diag) {  // (no reset record: one count per particle-substep whose acceleration has left the arithmetic)
                const unsigned long long nf = __ballot(!(fabs(qacc) < T(sizeof(T) == 4 ? 1e30 : 1e100)));
                if (l == 0 && ((unsigned)(nf >> (PL * half)) & (PL == 32 ? ~0u : 0xFFFFu)) != 0u) atomicAdd(diag + 1, 1u);
            }
            if (GEN && has_ball) {
                // mj_integratePos of a ball joint: q <- q * exp(h w / 2), w = the joint's three velocities (body frame) - the
                // first link gathers its followers' and owns the quaternion
                const T vn = v + h * qacc;
                const T wy = __shfl_down(vn, 1, PL), wz = __shfl_down(vn, 2, PL);
                if (ball_g == 0) {
                    const T nn = sqrt_(vn * vn + wy * wy + wz * wz);
                    if (nn > T(1e-15)) {
                        T sh, ch;
                        sincos_(T(0.5) * h * nn, sh, ch);
                        const T k = sh / nn, rx = vn * k, ry = wy * k, rz = wz * k;
                        const T w2 = qw * ch - q * rx - qy * ry - qz * rz, x2 = qw * rx + q * ch + qy * rz - qz * ry,
                                y2 = qw * ry - q * rz + qy * ch + qz * rx, z2 = qw * rz + q * ry - qy * rx + qz * ch;
                        const T inv = T(1) / sqrt_(w2 * w2 + x2 * x2 + y2 * y2 + z2 * z2);
                        qw = w2 * inv; q = x2 * inv; qy = y2 * inv; qz = z2 * inv;
                    }
                }
            }
            // Two wave-level tests guard what is rare about the integration.  `big`: a hinge step beyond the reach of the angle-
            // addition series (|dq| > 0.25 rad).  `chk`: MuJoCo's reset on instability - a NaN or an entry beyond mjMAXVAL = 1e10
            // in the acceleration or the integrated state can only come about through a velocity jump h |qacc| beyond 1e5 (1e10 h
            // is 1e6 or more for any time step of 1e-4 s or more; a NaN fails every <=) or a coordinate step beyond 0.25, so the
            // exact tests run behind it.  (Round 5's first version had ONE test, with the jump at 1e3, which also sent wavefronts
            // down the full sin / cos path when only an acceleration was large: pen-in-hand 9.90 ms per 4096 x 32 launch, 9.82
            // with the two tests apart, 9.67 with the emulation compiled out.)
            bool odd_r = rst && dof && !(fabs(h * qacc) <= T(1e5));
            if (any_rows) odd_r = odd_r || (rst && dof && fwd_odd);
            bool odd = false;
            const bool angle = dof && !(GEN && ball_g >= 0);        // my coordinate integrates as q += h v
            T dq = T(0);
            if (dof) v += h * qacc;
            if (angle) {
                dq = h * v;
                q += dq;
                odd = !slide && !(fabs(dq) <= T(0.25));
                odd_r = odd_r || (rst && !(fabs(dq) <= T(0.25)));
            }
            const bool big = __any(odd);
            const bool chk = rst && __any(odd_r);
            if (angle) {
                T sd, cd;
                sincos_small(dq, sd, cd);
                const T s1 = sq * cd + cq * sd, c1 = cq * cd - sq * sd;
                const T kk = T(1.5) - T(0.5) * (s1 * s1 + c1 * c1);
                sq = s1 * kk;
                cq = c1 * kk;
            }
            // (the full evaluation for the LANES whose step is beyond the series, behind a wave-level test: a lane's sine and
            // cosine do not depend on its wave-mates' steps - round 6)
            if (__builtin_expect(big, 0)) {
                if (odd) sincos_(q, sq, cq);
            }
            if (__builtin_expect(chk, 0)) {
                const unsigned long long bb = __ballot(acc_bad || state_is_bad());
                if (bb != 0ull) {
                    rst_any = true;
                    const bool acc_reset = (__ballot(acc_bad) & my_lanes) != 0ull;
                    rst_pend = !acc_reset && (bb & my_lanes) != 0ull;
                    if (acc_reset) {
                        // ... -> mj_resetData, mj_forward AGAIN and mj_Euler from there: the state one substep after the reset state -
                        // a constant of the model, made once per engine (TreeFusion::reset_rec); controls stay zero until the env
                        // step ends, and site_xpos is the reset state's (the second mj_forward's)
                        q = dof ? (T)rst[l] : T(0);
                        v = dof ? (T)rst[TL + l] : T(0);
                        if constexpr (GEN) {
                            qy = T(0); qz = T(0); qw = T(1);
                            if (ball_g == 0) { qy = (T)rst[l + 1]; qz = (T)rst[l + 2]; qw = (T)rst[TREE_QW + l]; }
                            if (ball_g > 0) q = T(0);
                        }
                        sincos_(q, sq, cq);
                        tau_act = act_id >= 0 ? M[T_GEAR + l] * fmin(fmax(T(0), M[T_CTRL_LO + l]), M[T_CTRL_HI + l]) : T(0);
                        lim_mem = 0; fl_mem = 0; cinst_mem = 0; cact_mem = 0;
                        xa_prev = T(0);
                        if (sub == frame_skip - 1)
                            for (int k = 0; k < 3; ++k) { hand[k] = (T)rst[TREE_STATE_LEN + k]; haxis[k] = (T)rst[TREE_STATE_LEN + 3 + k]; }
                        rst_ever = true;
                        if (diag && l == 0 && live) { atomicAdd(diag + 1, 1u); if (state_out) atomicAdd(diag + TREE_DIAG_ENV_RESETS, 1u); }
                    }
                }
            }
        }
        T cst;
        if (task == 1) {
            // reward = forward progress of qpos[0] over the env step / dt - c |a|^2, the action as given
            // (swimmer.py:10-19, half_cheetah.py:10-19)
            const T usq = sum_lanes<PL>(has_u ? u * u : T(0));
            cst = M[T_CTRL_COST] * usq - (__shfl(q, 0, PL) - X[A_MISC + 4]) * rcp_(h * T(frame_skip));
        } else {
            // reward = -(|h-g|_1 + 5 |h-g|_2), h = site position lagging one substep (reacher_env.py:31-35)
            const T dx = hand[0] - tgt[0], dy = hand[1] - tgt[1], dz = hand[2] - tgt[2];
            cst = fabs(dx) + fabs(dy) + fabs(dz) + T(5) * sqrt_(dx * dx + dy * dy + dz * dz);
            // task 2, the shape of pen-v0's reward (examples/configs/hand/pen-v0.yml:8): the object to its target
            // position, its axis to its target direction
            if (FRIC && task == 2)
                cst = sqrt_(dx * dx + dy * dy + dz * dz) -
                      (haxis[0] * M[T_TARGET_DIR] + haxis[1] * M[T_TARGET_DIR + 1] + haxis[2] * M[T_TARGET_DIR + 2]);
        }
        // Take delivery of the prefetched inputs HERE, before this step's stores are issued (as arm_rollout.hip does): loads and
        // stores share one in-order counter (vmcnt), and the register hand-over the compiler otherwise places on the loop's
        // back-edge waits with vmcnt(0) - i.e. for the cost / observation stores just issued
        asm volatile("" : "+v"(eps_next), "+v"(mean_next));
        if (__builtin_expect(rst_any, 0)) { if (fuse.inf_on_reset && rst_ever) cst = T(INFINITY); }
        if (live && l == 0) cost[pid * H + t] = cst;
        if (fuse.gseq) q0acc += gs_cur * (double)cst;
        // my link's coordinate(s) in MuJoCo's qpos layout (GEN: a ball's first link writes the quaternion, w first; a free
        // joint's translations are absolute positions; without GEN qadr = l, nq = nv)
        auto put_q = [&](T* dst, long o, int skip, T x_, T y_, T z_, T w_) {
            if (!dof || qadr < skip) return;
            if (GEN && ball_g == 0) {
                const T x0 = qoff;
                dst[o + qadr - skip] = q0w * w_ - x0 * x_ - q0y * y_ - q0z * z_;
                dst[o + qadr - skip + 1] = q0w * x_ + x0 * w_ + q0y * z_ - q0z * y_;
                dst[o + qadr - skip + 2] = q0w * y_ - x0 * z_ + q0y * w_ + q0z * x_;
                dst[o + qadr - skip + 3] = q0w * z_ + x0 * y_ - q0y * x_ + q0z * w_;
            } else if (!GEN || ball_g < 0) {
                dst[o + qadr - skip] = x_ + qoff;
            }
        };
        if (live && (obs || nobs) && task == 1) {           // obs = [qpos[skip:], qvel]
            const long o = (pid * H + t) * dobs;
            if (obs && dof) {
                put_q(obs, o, obs_skip, q_prev, qy_prev, qz_prev, qw_prev);
                obs[o + nq - obs_skip + l] = v_prev;
            }
            if (nobs && dof) {
                put_q(nobs, o, obs_skip, q, qy, qz, qw);
                nobs[o + nq - obs_skip + l] = v;
            }
        } else if (live && (obs || nobs)) {
            const long o = (pid * H + t) * dobs;
            if (obs) {
                if (dof) { put_q(obs, o, 0, q_prev, qy_prev, qz_prev, qw_prev); obs[o + nq + l] = v_prev; }
                if (l < 3) { obs[o + nq + nv + l] = hand_prev[l]; obs[o + nq + nv + 3 + l] = hand_prev[l] - tgt[l]; }
            }
            if (nobs) {
                if (dof) { put_q(nobs, o, 0, q, qy, qz, qw); nobs[o + nq + l] = v; }
                if (l < 3) { nobs[o + nq + nv + l] = hand[l]; nobs[o + nq + nv + 3 + l] = hand[l] - tgt[l]; }
            }
        }
        q_prev = q;
        v_prev = v;
        if constexpr (GEN) { qy_prev = qy; qz_prev = qz; qw_prev = qw; }
        for (int k = 0; k < 3; ++k) hand_prev[k] = hand[k];
    }
    // a rollout that diverged numerically carries a non-finite return: +inf - zero weight in the softmax updates, last in
    // the elite ranking - instead of a NaN in the mean (as the arm kernel)
    if (fuse.q0_out && live && l == 0) fuse.q0_out[pid] = fabs(q0acc) < (double)INFINITY ? q0acc : (double)INFINITY;
    // the "real env" kept on the device (mjmpc_tree_step_state): particle 0 leaves its state where the next rollout
    // reads it (the launch has one particle; `state` was read before the first step)
    if (state_out && pid == 0 && dof) {
        if (!GEN || ball_g <= 0) state_out[l] = (double)q;         // (a ball's first link owns its followers' qpos entries)
        if (GEN && ball_g == 0) {
            state_out[l + 1] = (double)qy;
            state_out[l + 2] = (double)qz;
            state_out[TREE_QW + l] = (double)qw;
        }
        state_out[TL + l] = (double)v;
    }
}

}  // namespace

// The 16-lane dense instantiations live in a translation unit of their own (tree_rollout_dense.hip, which includes this
// file with TREE_DENSE_TU defined): it is compiled with the iterative-ILP scheduling strategy, which pays for those
// one-wave-per-SIMD kernels and not for the others (mjmpc_amd/build.py).
struct TreeLaunchArgs {
    int model_stride, state_stride, n_shards, H, A;
    long P, shard;
    const double *state, *mean, *clw;
    unsigned* diag;
    double *state_out, *site_out;
    hipStream_t stream;
    TreeFusion fuse;
};
template <typename T>
hipError_t launch_tree_rollout_dense(int max_path, int nv, int gen, const T* model, const T* noise, T* cost, T* act, T* obs, T* nobs,
                                     const TreeLaunchArgs& a);
// ... and so do the instantiations for models with ELLIPTIC friction cones (GEN = 3; tree_rollout_cone.hip, TREE_CONE_TU)
template <typename T>
hipError_t launch_tree_rollout_cone(int max_path, int nv, const T* model, const T* noise, T* cost, T* act, T* obs, T* nobs,
                                    const TreeLaunchArgs& a);

#define MJMPC_TREE_LAUNCH(DP_, NS_, FR_, PL_) MJMPC_TREE_LAUNCH_D(DP_, NS_, FR_, PL_, 0, 0)
#define MJMPC_TREE_LAUNCH_G(DP_, NS_, FR_, PL_, G_) MJMPC_TREE_LAUNCH_D(DP_, NS_, FR_, PL_, 0, G_)
#define MJMPC_TREE_LAUNCH_D(DP_, NS_, FR_, PL_, DN_, GEN_)                                                            \
    {                                                                                                                 \
        constexpr int per_wg = wg_waves(DP_, FR_, sizeof(T), PL_) * (64 / PL_);                                       \
        hipLaunchKernelGGL((tree_rollout_kernel<T, DP_, NS_, FR_, PL_, DN_, GEN_>),                                   \
                           dim3((unsigned)((a.shard + per_wg - 1) / per_wg), (unsigned)a.n_shards),                   \
                           dim3(64 * wg_waves(DP_, FR_, sizeof(T), PL_)), 0, a.stream, model, a.model_stride, a.state, \
                           a.state_stride, a.P, a.shard, a.H, a.A, a.mean, noise, cost, act, obs, nobs, a.diag,       \
                           a.state_out, a.clw, a.site_out, a.fuse);                                                   \
    }

#ifdef TREE_DENSE_TU
template <typename T>
hipError_t launch_tree_rollout_dense(int max_path, int nv, int gen, const T* model, const T* noise, T* cost, T* act, T* obs, T* nobs,
                                     const TreeLaunchArgs& a) {
    // (16 lanes per particle: the dense in-register factorisation, sized for the model)
    if (nv > 16) {      // 32 lanes per particle, the dense factorisation over the particle's two DPP rows (dense32_factor)
        if (gen >= 2) MJMPC_TREE_LAUNCH_D(16, 16, true, 32, 32, 2)     // (round 5's record kinds: instantiations of their own)
        else if (gen) MJMPC_TREE_LAUNCH_D(16, 16, true, 32, 32, 1)     // (general models: paths of up to 16 links)
        else if (max_path <= 16) MJMPC_TREE_LAUNCH_D(16, 16, true, 32, 32, 0)
        else MJMPC_TREE_LAUNCH_D(32, 16, true, 32, 32, 0)
    }
    else if (gen >= 2) {
        if (max_path <= 4 && nv <= 4) MJMPC_TREE_LAUNCH_D(4, 16, true, 16, 4, 2)
        else if (max_path <= 8 && nv <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 8, 2)
        else if (max_path <= 8 && nv <= 12) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 12, 2)
        else if (max_path <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 16, 2)
        else MJMPC_TREE_LAUNCH_D(16, 16, true, 16, 16, 2)
    }
    else if (gen) {          // the general instantiation comes in four sizes (measured, 4096 x 32 f64: cart-pole 0.95 -> 0.68 ms and door 1.55 -> 1.22 ms with rows of 8 instead of 16)
        // (round 5: rows and paths of 4 for models of up to four dofs - cart-pole 0.65 -> 0.54 ms, door 1.25 -> 1.09, f32 cart-pole
        // 0.44 -> 0.36; eight record slots instead of sixteen on top of that: no change, not kept)
        if (max_path <= 2 && nv <= 2) MJMPC_TREE_LAUNCH_D(2, 16, true, 16, 2, 1)         // (... and of 2: cart-pole 0.54 -> 0.475 ms, f32 0.365 -> 0.32)
        else if (max_path <= 4 && nv <= 4) MJMPC_TREE_LAUNCH_D(4, 16, true, 16, 4, 1)
        else
        if (max_path <= 8 && nv <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 8, 1)
        else if (max_path <= 8 && nv <= 12) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 12, 1)
        else if (max_path <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 16, 1)
        else if (max_path <= 10 && nv <= 10) MJMPC_TREE_LAUNCH_D(10, 16, true, 16, 10, 1)      // (... tray, 10 dofs on paths of 10: 2.97 -> 2.74 ms, f32 1.37 -> 1.29)
        else if (max_path <= 12 && nv <= 12) MJMPC_TREE_LAUNCH_D(12, 16, true, 16, 12, 1)      // (round 5: tray 3.52 -> 2.99 ms, f32 1.59 -> 1.37)
        else MJMPC_TREE_LAUNCH_D(16, 16, true, 16, 16, 1)
    }
    // (... and the reference's two vendored locomotion models at their own sizes: Swimmer-v0, 7 dofs on a path of 7: 1.00 -> 0.97 ms;
    // HalfCheetah-v0, 9 dofs: 2.14 -> 2.07, f32 1.92 -> 1.87)
    else if (max_path <= 7 && nv <= 7) MJMPC_TREE_LAUNCH_D(7, 16, true, 16, 7, 0)
    else if (max_path <= 8 && nv <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 8, 0)
    else if (max_path <= 8 && nv <= 9) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 9, 0)
    // (paths of 6 instead of 8 for the cheetah: 1 %, not kept)
    else if (max_path <= 8 && nv <= 10) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 10, 0)    // (round 5: HalfCheetah, 9 dofs: 2.29 -> 2.16 ms, f32 2.06 -> 1.92, with rows of 10 instead of 12)
    else if (max_path <= 8 && nv <= 12) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 12, 0)
    else if (max_path <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 16, 0)
    else MJMPC_TREE_LAUNCH_D(16, 16, true, 16, 16, 0)
    return hipGetLastError();
}
template hipError_t launch_tree_rollout_dense<float>(int, int, int, const float*, const float*, float*, float*, float*, float*,
                                                     const TreeLaunchArgs&);
template hipError_t launch_tree_rollout_dense<double>(int, int, int, const double*, const double*, double*, double*, double*, double*,
                                                      const TreeLaunchArgs&);
#elif defined(TREE_CONE_TU)
template <typename T>
hipError_t launch_tree_rollout_cone(int max_path, int nv, const T* model, const T* noise, T* cost, T* act, T* obs, T* nobs,
                                    const TreeLaunchArgs& a) {
    if (nv > 16) {
        if (max_path <= 16) MJMPC_TREE_LAUNCH_D(16, 16, true, 32, 32, 3)       // dense over the particle's 32 lanes
        else MJMPC_TREE_LAUNCH_G(32, 16, true, 32, 3)                          // tree-sparse (elimination paths beyond 16 links)
    }
    else if (max_path <= 4 && nv <= 4) MJMPC_TREE_LAUNCH_D(4, 16, true, 16, 4, 3)
    else if (max_path <= 8 && nv <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 8, 3)
    else if (max_path <= 8 && nv <= 12) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 12, 3)
    else if (max_path <= 8) MJMPC_TREE_LAUNCH_D(8, 16, true, 16, 16, 3)
    else MJMPC_TREE_LAUNCH_D(16, 16, true, 16, 16, 3)
    return hipGetLastError();
}
template hipError_t launch_tree_rollout_cone<float>(int, int, const float*, const float*, float*, float*, float*, float*, const TreeLaunchArgs&);
template hipError_t launch_tree_rollout_cone<double>(int, int, const double*, const double*, double*, double*, double*, double*,
                                                     const TreeLaunchArgs&);
#else
template <typename T>
hipError_t launch_tree_rollout(const T* model, int n_model_shards, int max_path, bool full, int nv, const double* state, long P, int H,
                               int A, const double* mean, const T* noise, T* cost, T* act, T* obs, T* nobs, unsigned* diag,
                               hipStream_t stream, double* state_out, const double* clw, double* site_out, int n_state_shards,
                               int gen, TreeFusion fuse) {
    if (P <= 0 || H <= 0) return hipSuccess;
    if (gen && !full) return hipErrorInvalidValue;
    if ((state_out || site_out) && P != 1) return hipErrorInvalidValue;
    if (n_model_shards < 1 || n_state_shards < 1) return hipErrorInvalidValue;
    if (n_model_shards > 1 && n_state_shards > 1 && n_model_shards != n_state_shards) return hipErrorInvalidValue;
    TreeLaunchArgs a;
    a.n_shards = n_model_shards > n_state_shards ? n_model_shards : n_state_shards;
    if (P % a.n_shards != 0) return hipErrorInvalidValue;
    a.P = P;
    a.shard = P / a.n_shards;
    a.H = H;
    a.A = A;
    a.model_stride = n_model_shards > 1 ? TREE_BLOB_LEN : 0;
    a.state_stride = n_state_shards > 1 ? TREE_STATE_LEN : 0;
    a.state = state;
    a.mean = mean;
    a.clw = clw;
    a.diag = diag;
    a.state_out = state_out;
    a.site_out = site_out;
    a.stream = stream;
    a.fuse = fuse;
    // hinge trees in air with up to 8 frictionless contact points keep the lean instantiation; slide joints, springs,
    // friction cones, more points or a medium take the full one (three Jacobians per point, 16 points, fluid forces),
    // which also comes with 16 lanes per particle for models of up to 16 dofs (the reference's swimmer and cheetah)
    if (gen >= 3) return launch_tree_rollout_cone<T>(max_path, nv, model, noise, cost, act, obs, nobs, a);
    if (!full) {
        if (max_path <= 8) MJMPC_TREE_LAUNCH(8, 8, false, 32)
        else if (max_path <= 16) MJMPC_TREE_LAUNCH(16, 8, false, 32)
        else MJMPC_TREE_LAUNCH(32, 8, false, 32)
    } else if (nv <= 16) {
        return launch_tree_rollout_dense<T>(max_path, nv, gen, model, noise, cost, act, obs, nobs, a);
    } else if (gen >= 2 ? max_path <= 16 : (max_path > 8 && !(gen && max_path > 16) && !getenv("MJMPC_TREE_SPARSE"))) {
        // 17 .. 32 dofs on elimination paths of more than 8 links: dense over the particle's 32 lanes - measured at 4096 x 32:
        // pen-in-hand (paths of 16) f64 16.2 -> 9.8 ms, f32 13.8 -> 8.0; with paths of up to 8 links (a hand with friction
        // cones) the tree-sparse factorisation with its merged Euler matrix stays ahead, 4.15 against 4.27 ms
        // (MJMPC_TREE_SPARSE in the environment keeps the tree-sparse one everywhere: the A/B switch of tools/tree_time.py)
        return launch_tree_rollout_dense<T>(max_path, nv, gen, model, noise, cost, act, obs, nobs, a);
    } else if (gen >= 2) {
        // (round 5's record kinds: dense over 32 lanes whatever the path length up to 16 links - the one instantiation; the
        // tree-sparse kernel only for elimination paths beyond that)
        MJMPC_TREE_LAUNCH_G(32, 16, true, 32, 2)
    } else if (gen) {
        if (max_path <= 16) MJMPC_TREE_LAUNCH_G(16, 16, true, 32, 1)
        else MJMPC_TREE_LAUNCH_G(32, 16, true, 32, 1)
    } else {
        if (max_path <= 8) MJMPC_TREE_LAUNCH(8, 16, true, 32)
        else if (max_path <= 16) MJMPC_TREE_LAUNCH(16, 16, true, 32)
        else MJMPC_TREE_LAUNCH(32, 16, true, 32)
    }
    return hipGetLastError();
}

template hipError_t launch_tree_rollout<float>(const float*, int, int, bool, int, const double*, long, int, int, const double*,
                                               const float*, float*, float*, float*, float*, unsigned*, hipStream_t, double*, const double*, double*, int, int, TreeFusion);
template hipError_t launch_tree_rollout<double>(const double*, int, int, bool, int, const double*, long, int, int, const double*,
                                                const double*, double*, double*, double*, double*, unsigned*, hipStream_t, double*, const double*, double*, int, int, TreeFusion);
#endif
#undef MJMPC_TREE_LAUNCH
#undef MJMPC_TREE_LAUNCH_G
#undef MJMPC_TREE_LAUNCH_D

}  // namespace mjmpc

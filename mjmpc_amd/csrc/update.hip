// Sampling-distribution update kernels: the device side of the controllers' _update_distribution
// (reference mjmpc/control/{mppi,gaussian_dmd,cem,random_shooting,particle_filter_controller}.py).
//
// All of them reduce P particles x (H*A) actions to a handful of (H*A)- or (A*A)-sized moments, so
// the pattern is the same everywhere:
//   1. per-particle scalar(s)    (discounted cost-to-go, +lam * control cost)   one thread / particle
//   2. block-partial moments     fixed chunk of particles per workgroup, lanes run along the
//                                contiguous (H*A) axis of `actions` -> coalesced, deterministic order
//   3. ordered sum of partials   -> a small float64 RECORD per GPU
//   4. combine G records         (G = number of GPUs; records travel by one all-gather over xGMI)
// Accumulation is float64 regardless of the storage type T of costs/actions.  Nothing here uses
// atomics on floating point, so results are bit-reproducible run to run and identical on all ranks.
#include <hip/hip_runtime.h>

#include "noise_device.h"
#include "update.h"

namespace mjmpc {
namespace {

constexpr int BLK = 256;

__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// x[p][tw] = (-1/lam) * (cost_to_go(costs)[p][tw] + lam * cost_to_go(ctrl_cost)[p][tw]),  tw < Hw
// q0[p]    = cost_to_go(costs)[p][0]                                    (optional output)
// cost_to_go follows mjmpc/utils/control_utils.py:37-46 in the same summation order.
template <typename T>
__global__ void traj_cost_kernel(const T* __restrict__ costs, const T* __restrict__ actions,
                                 const double* __restrict__ mean, const double* __restrict__ un,
                                 const double* __restrict__ gseq, int gamma_zero, double lam, long P, int H, int A,
                                 int Hw, double* __restrict__ x, double* __restrict__ q0) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double neg_inv_lam = -1.0 / lam;
    double acc = 0.0, accc = 0.0;
    for (int t = H - 1; t >= 0; --t) {
        const double c = (double)costs[p * H + t];
        double qt, cq = 0.0;
        if (gamma_zero) {
            qt = c;
        } else {
            acc += gseq[t] * c;
            qt = acc / gseq[t];
        }
        if (un) {
            double cc = 0.0;
            for (int a = 0; a < A; ++a) {
                const double m = mean[t * A + a];
                const double d = (double)actions[(p * H + t) * A + a] - m;
                cc += 0.5 * un[t * A + a] * (m + 2.0 * d);
            }
            if (gamma_zero) {
                cq = cc;
            } else {
                accc += gseq[t] * cc;
                cq = accc / gseq[t];
            }
        }
        // (a rollout that diverged numerically - MuJoCo would have reset that simulation - carries a non-finite return:
        // +inf, i.e. zero weight / last in the ranking, instead of a NaN that poisons the mean)
        if (!(fabs(qt) < INFINITY)) qt = INFINITY;
        if (Hw > 1) x[p * Hw + t] = neg_inv_lam * (qt + lam * cq);
        else if (t == 0) x[p] = neg_inv_lam * (qt + lam * cq);
        if (t == 0 && q0) q0[p] = qt;
    }
}

// MPPIQ.calculate_returns (mjmpc/control/mppiq.py:104-126), one particle per thread:
//   total[t] = cost[t] + beta * control_cost[t]        (per step, not accumulated; mppiq.py:128-136)
//   td[t]    = total[t] + gamma q[t+1] - q[t],  t < H-1;   q = qvals, or 0 with q[H-1] = total[H-1]
//   out[t]   = q[t] + td_lam * cost_to_go(td, wseq)[t],    out[H-1] = q[H-1]
// cost_to_go in the order of control_utils.py:37-46 (reverse running sum, then / wseq; unchanged if wseq has a 0).
template <typename T>
__global__ void td_lambda_kernel(const T* __restrict__ costs, const T* __restrict__ actions, const T* __restrict__ qvals,
                                 const double* __restrict__ mean, const double* __restrict__ un,
                                 const double* __restrict__ wseq, int wseq_zero, double beta, double gamma,
                                 double td_lam, long P, int H, int A, T* __restrict__ out) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    auto total = [&](int t) {
        double c = (double)costs[p * H + t];
        if (un) {
            double cc = 0.0;
            for (int a = 0; a < A; ++a) {
                const double m = mean[t * A + a];
                const double d = (double)actions[(p * H + t) * A + a] - m;
                cc += 0.5 * un[t * A + a] * (m + 2.0 * d);
            }
            c = c + beta * cc;
        }
        return c;
    };
    const double qlast = qvals ? (double)qvals[p * H + H - 1] : total(H - 1);
    out[p * H + H - 1] = (T)qlast;
    double acc = 0.0, qnext = qlast;
    for (int t = H - 2; t >= 0; --t) {
        const double q = qvals ? (double)qvals[p * H + t] : 0.0;
        const double td = total(t) + gamma * qnext - q;
        double g;
        if (wseq_zero) {
            g = td;
        } else {
            acc += wseq[t] * td;
            g = acc / wseq[t];
        }
        out[p * H + t] = (T)(q + td_lam * g);
        qnext = q;
    }
}

// u_n = mean . cov^-1      (mppi.py:106)
__global__ void un_kernel(const double* __restrict__ mean, const double* __restrict__ covinv, int H, int A,
                          double* __restrict__ un) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H * A) return;
    const int t = j / A, a = j % A;
    double s = 0.0;
    for (int b = 0; b < A; ++b) s += mean[t * A + b] * covinv[b * A + a];
    un[j] = s;
}

// xmax[tw] = max_p x[p][tw]; one workgroup per column
__global__ void colmax_kernel(const double* __restrict__ x, long P, int Hw, double* __restrict__ xmax) {
    __shared__ double sm[BLK / 64];
    const int tw = blockIdx.x;
    double m = -INFINITY;
    for (long p = threadIdx.x; p < P; p += blockDim.x) m = fmax(m, x[p * Hw + tw]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLK / 64; ++w) m = fmax(m, sm[w]);
        xmax[tw] = m;
    }
}

// partial[b] = { S[Hw], W[H*A], C[A*A] } over the particle chunk of workgroup b
template <typename T>
__global__ void softmax_partial_kernel(const double* __restrict__ x, const double* __restrict__ xmax,
                                       const T* __restrict__ actions, const double* __restrict__ mean, long P, int H,
                                       int A, int Hw, int chunk, int want_cov, double* __restrict__ partial) {
    extern __shared__ double e_s[];                 // chunk * Hw
    const int HA = H * A, rec = Hw + HA + A * A;
    const long p0 = (long)blockIdx.x * chunk;
    const int n = (int)((P - p0) < chunk ? (P - p0) : chunk);
    double* out = partial + (long)blockIdx.x * rec;
    for (int i = threadIdx.x; i < n * Hw; i += blockDim.x) e_s[i] = exp(x[p0 * Hw + i] - xmax[i % Hw]);
    __syncthreads();
    for (int tw = threadIdx.x; tw < Hw; tw += blockDim.x) {
        double s = 0.0;
        for (int p = 0; p < n; ++p) s += e_s[p * Hw + tw];
        out[tw] = s;
    }
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        const int tw = Hw > 1 ? j / A : 0;
        double s = 0.0;
        for (int p = 0; p < n; ++p) s += e_s[p * Hw + tw] * (double)actions[(p0 + p) * HA + j];
        out[Hw + j] = s;
    }
    for (int idx = threadIdx.x; idx < A * A; idx += blockDim.x) {
        double s = 0.0;
        if (want_cov) {
            const int i = idx / A, k = idx % A;
            for (int p = 0; p < n; ++p) {
                double sp = 0.0;
                const T* ap = actions + (p0 + p) * HA;
                for (int t = 0; t < H; ++t)
                    sp += ((double)ap[t * A + i] - mean[t * A + i]) * ((double)ap[t * A + k] - mean[t * A + k]);
                s += e_s[p * Hw] * sp;
            }
        }
        out[Hw + HA + idx] = s;
    }
}

// sum over workgroups of entry j of the partials: one wavefront per entry, lane l adds partials
// l, l+64, ... in order, then a fixed butterfly - the same tree every run, on every GPU
__device__ __forceinline__ double wave_entry_sum(const double* __restrict__ partial, int nb, int rec, int j) {
    const int l = threadIdx.x & 63;
    double s = 0.0;
    for (int b = l; b < nb; b += 64) s += partial[(long)b * rec + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// record = [ xmax[Hw] | sum over workgroups of partial ]
__global__ void softmax_record_kernel(const double* __restrict__ partial, const double* __restrict__ xmax, int nb,
                                      int Hw, int rec, double* __restrict__ record) {
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= rec) return;
    const double s = wave_entry_sum(partial, nb, rec, j);
    if ((threadIdx.x & 63) == 0) {
        record[Hw + j] = s;
        if (j < Hw) record[j] = xmax[j];
    }
}

// mean <- (1-eta) mean + eta * sum_g sc_g W_g / sum_g sc_g S_g,   sc_g = exp(xmax_g - max_g xmax_g);
// cov likewise (mode 1: diagonal only, 2: full); value = -lam * logsumexp(x, b = 1/P_total).
__global__ void softmax_combine_kernel(const double* __restrict__ records, int G, int H, int A, int Hw, double lam,
                                       double step, int cov_mode, double P_total, double* __restrict__ mean,
                                       double* __restrict__ cov, double* __restrict__ value,
                                       double* __restrict__ wnorm) {
    const int HA = H * A, rlen = 2 * Hw + HA + A * A;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    auto scale = [&](int g, int tw, double M) { return exp(records[(long)g * rlen + tw] - M); };
    auto colmax = [&](int tw) {
        double M = -INFINITY;
        for (int g = 0; g < G; ++g) M = fmax(M, records[(long)g * rlen + tw]);
        return M;
    };
    auto total = [&](int tw, double M) {
        double S = 0.0;
        for (int g = 0; g < G; ++g) S += scale(g, tw, M) * records[(long)g * rlen + Hw + tw];
        return S;
    };
    if (j < HA) {
        const int tw = Hw > 1 ? j / A : 0;
        const double M = colmax(tw), S = total(tw, M);
        double W = 0.0;
        for (int g = 0; g < G; ++g) W += scale(g, tw, M) * records[(long)g * rlen + 2 * Hw + j];
        mean[j] = (1.0 - step) * mean[j] + step * (W / S);
    }
    if (cov_mode && j < A * A) {
        const int i = j / A, k = j % A;
        const double M = colmax(0), S = total(0, M);
        double C = 0.0;
        for (int g = 0; g < G; ++g) C += scale(g, 0, M) * records[(long)g * rlen + 2 * Hw + HA + j];
        double upd = C / S / (double)H;
        if (cov_mode == 1 && i != k) upd = 0.0;
        cov[j] = (1.0 - step) * cov[j] + step * upd;
    }
    if (j == 0) {
        const double M = colmax(0), S = total(0, M);
        if (value) *value = -lam * (log(S / P_total) + M);
        if (wnorm) { wnorm[0] = M; wnorm[1] = S; }
    }
}

// w[p] = exp(x[p] - M) / S          (PFMPC weights, particle_filter_controller.py:104-113)
__global__ void softmax_weights_kernel(const double* __restrict__ x, const double* __restrict__ wnorm, long P,
                                       double* __restrict__ w) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) w[p] = exp(x[p] - wnorm[0]) / wnorm[1];
}

// ---- CEM ------------------------------------------------------------------------------------------
// Elite selection = the k smallest particles in (q0, global index) order.  A most-significant-byte radix select
// over the order-preserving integer image of the doubles finds the k-th smallest key in 8 passes of one
// workgroup (the first version ranked every particle against every other: 1.2 ms at 16 384 particles);
__device__ __forceinline__ unsigned long long order_key(double q) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(q);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// thr = { k-th smallest key T, (unused), cut }: among the particles whose key EQUALS T, those with global index <= cut
// complete the elite set (ties go to the smaller index).
// NPER > 0: every thread keeps its NPER keys (particles tid, tid + 1024, ...) in registers for all passes - the
// passes are then pure register / LDS work instead of 8 x P/1024 dependent trips to L2 (138 -> ~15 us at 16 384).
// Round 3: the leading bits all keys share are found by one AND / OR reduction instead of a pass each (the costs of one
// population share sign, exponent and often more) and the 8-bit digit windows count down from the first differing bit; once the bucket that holds the rank has at most KTH_CAND keys
// they are ranked against each other directly in (key, index) order - which also settles ties exactly, so the
// barrier-per-round tie walk only runs when more than KTH_CAND particles hold the k-th key itself.
// sum_{g < n} p[g * stride] in a FIXED order with eight loads in flight: a plain `for (g) acc += p[g * stride]` waits for
// every load before it issues the next - half a microsecond per term when the terms are in L2 (measured: 59 partials of
// the fused CEM step summed that way cost 31 us)
__device__ __forceinline__ double sum_strided(const double* __restrict__ p, int n, long stride) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
    int g = 0;
    for (; g + 8 <= n; g += 8) {
        const double v0 = p[(long)g * stride], v1 = p[(long)(g + 1) * stride], v2 = p[(long)(g + 2) * stride],
                     v3 = p[(long)(g + 3) * stride], v4 = p[(long)(g + 4) * stride], v5 = p[(long)(g + 5) * stride],
                     v6 = p[(long)(g + 6) * stride], v7 = p[(long)(g + 7) * stride];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
    }
    for (; g < n; ++g) a0 += p[(long)g * stride];
    return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
constexpr int NOISE_MAXA_U = 8;     // as noise.hip's NOISE_MAXA: action channels the in-register sampler unrolls to
constexpr int KTH_CAND = 32;        // (the ranking is a loop of dependent LDS reads: 256 candidates cost more than a pass)
// Round 4, MOM = true (cem_select_moments): the SAME selection run redundantly by every workgroup of a grid - the keys are
// 128 KB that every CU reads from L2, the selection is a few dozen barriers of one workgroup either way - so that each
// workgroup knows the elite list and goes straight on to the moments of ITS slice of it (E elite rows): the sum of the
// rows and their scatter about a PROVISIONAL centre (the mean of the first CEM_HEAD elite rows, which every workgroup
// forms alike) - the shifted-data form of the two-pass np.cov: cem_finish subtracts N (mu - c)(mu - c)' with mu - c a
// fraction of a standard deviation, so nothing cancels.  One launch replaces selection, list, row sums, their reduction,
// the mean, and the scatter pass.  Workgroup 0 also snapshots what the finish launch must read unmodified while it
// rewrites it (mean, covariance, step counter).
constexpr int CEM_HEAD = 64, CEM_E_MAX = 64;
#ifndef CEM_E_ROWS
#define CEM_E_ROWS 32
#endif
template <typename TA>
struct CemMoments {
    const TA* actions;
    const double *mean, *cov;
    const long long* d_step;
    double *partial, *cprime, *mean_prev, *cov_prev;
    long long* step_prev;
    int H, A, E;
};
template <int NPER, typename TA = double, bool MOM = false>
__global__ __launch_bounds__(1024) void kth_key_kernel(const double* __restrict__ q_all, long P_all, long k,
                                                       unsigned long long* __restrict__ thr, long offset, long P_local,
                                                       int* __restrict__ list, int* __restrict__ count,
                                                       CemMoments<TA> mo = CemMoments<TA>()) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long prefix_s, red_and[16], red_or[16], cand_key[KTH_CAND];
    __shared__ long need_s, cut_s, sel_need, wtot[16], scan_v[256], cand_idx[KTH_CAND];
    __shared__ int sel_bin, sel_cnt, ncand;
    bool single = false;                    // the selection has named the last elite particle itself (T and cut)
    const int tid = threadIdx.x;
    if (k <= 0) {                                           // empty elite set
        if (tid == 0) {
            thr[0] = 0ull; thr[1] = 0ull; thr[2] = ~0ull;
            if (list) *count = 0;
        }
        return;
    }
    constexpr int NK = NPER > 0 ? NPER : 1;
    unsigned long long keys[NK];
    if constexpr (NPER > 0) {
#pragma unroll
        for (int r = 0; r < NPER; ++r) {
            const long j = (long)r * 1024 + tid;
            keys[r] = j < P_all ? order_key(q_all[j]) : ~0ull;
        }
    }
    const long rounds = NPER > 0 ? NPER : (P_all + 1023) / 1024;
    auto key_at = [&](long r) -> unsigned long long {
        if constexpr (NPER > 0) return keys[r];
        const long j = r * 1024 + tid;
        return j < P_all ? order_key(q_all[j]) : ~0ull;
    };
    // bits every key shares: the passes start at the first bit in which two keys differ
    unsigned long long kand = ~0ull, kor = 0ull;
#pragma unroll
    for (long r = 0; r < rounds; ++r)
        if (r * 1024 + tid < P_all) {
            const unsigned long long key = key_at(r);
            kand &= key;
            kor |= key;
        }
    for (int o = 32; o > 0; o >>= 1) {
        kand &= __shfl_xor(kand, o);
        kor |= __shfl_xor(kor, o);
    }
    if ((tid & 63) == 0) { red_and[tid >> 6] = kand; red_or[tid >> 6] = kor; }
    if (tid == 0) { need_s = k; cut_s = -1; }       // need = rank still to be located
    __syncthreads();
    for (int w = 0; w < 16; ++w) { kand &= red_and[w]; kor |= red_or[w]; }
    // digits are 8-bit windows counted down from the most significant bit in which two keys differ (not byte-aligned:
    // the first window then spreads the population over up to 256 buckets instead of the handful a byte that is mostly
    // exponent offers - few buckets mean 64 lanes serialising on a few LDS words - and the second usually ends it)
    const unsigned long long differ = kand ^ kor;
    const int top = differ ? 64 - __clzll((long long)differ) : 0;       // 0: all keys are equal
    if (tid == 0) prefix_s = top == 64 ? 0ull : (kand & (~0ull << top));
    __syncthreads();
#if defined(KTH_STOP) && KTH_STOP == 1       // developer builds: phase timing by early exits (tools/cem_phases.sh)
    return;
#endif
    for (int hi = top; hi > 0;) {
        const int width = hi >= 8 ? 8 : hi, shift = hi - width;
        hi = shift;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        const unsigned long long himask = shift + width == 64 ? 0ull : (~0ull << (shift + width));
#pragma unroll
        for (long r = 0; r < rounds; ++r) {
            const unsigned long long key = key_at(r);
            const bool in = (r * 1024 + tid < P_all) && (key & himask) == prefix;
            const unsigned bin = (unsigned)(key >> shift) & ((1u << width) - 1u);
            // costs of one population share their leading bytes: when every candidate of the wavefront falls
            // into the same bin, one lane adds the count instead of 64 lanes serialising on one LDS word
            const unsigned long long m = __ballot(in);
            if (m) {
                const unsigned lead = (unsigned)__builtin_amdgcn_readlane((int)bin, __ffsll((long long)m) - 1);  // (v_readlane: no LDS trip)
                if (__all(!in || bin == lead)) {
                    if ((tid & 63) == 0) atomicAdd(&hist[lead], (unsigned)__popcll(m));
                } else if (in) {
                    atomicAdd(&hist[bin], 1u);
                }
            }
        }
        __syncthreads();
        // bucket holding the rank: inclusive scan of the 256 counts by 4 wavefronts (a serial walk by one thread
        // costs an LDS round trip per bucket - 10 us per pass)
        if (tid < 256) {
            const long v = hist[tid];
            long incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const long u = __shfl_up(incl, o);
                if ((tid & 63) >= o) incl += u;
            }
            if ((tid & 63) == 63) wtot[tid >> 6] = incl;
            scan_v[tid] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            long off = 0;
            for (int w = 0; w < (tid >> 6); ++w) off += wtot[w];
            const long incl = scan_v[tid] + off, excl = incl - (long)hist[tid], need = need_s;
            const long total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            if ((excl < need && need <= incl) || (tid == (1 << width) - 1 && total < need)) {   // (k > P_all: everything is elite)
                sel_bin = tid;
                sel_need = need - excl;
                sel_cnt = (int)(hist[tid] < 0x7fffffffu ? hist[tid] : 0x7fffffffu);
            }
        }
        __syncthreads();
        if (tid == 0) {
            need_s = sel_need;
            prefix_s = prefix | ((unsigned long long)sel_bin << shift);
            ncand = 0;
        }
        __syncthreads();
        // Few keys left in the selected bucket (and the rank lies inside it): rank them against each other in
        // (key, index) order - the need-th one IS the k-th key, and its index the tie cut; the remaining passes are
        // skipped (costs of one population separate after two 8-bit windows)
        if (sel_cnt <= KTH_CAND && sel_need <= (long)sel_cnt) {
            const unsigned long long want = prefix_s, mask = ~0ull << shift;
            const int n = sel_cnt;
#pragma unroll
            for (long r = 0; r < rounds; ++r) {
                const unsigned long long key = key_at(r);
                if ((r * 1024 + tid < P_all) && (key & mask) == want) {
                    const int slot = atomicAdd(&ncand, 1);
                    cand_key[slot] = key;
                    cand_idx[slot] = r * 1024 + tid;
                }
            }
            __syncthreads();
            if (tid < n) {
                const unsigned long long mk = cand_key[tid];
                const long mi = cand_idx[tid];
                long rank = 0;
                for (int u = 0; u < n; ++u) {
                    const unsigned long long uk = cand_key[u];
                    rank += (uk < mk) || (uk == mk && cand_idx[u] < mi);
                }
                if (rank + 1 == need_s) { prefix_s = mk; cut_s = mi; }
            }
            __syncthreads();
            single = true;
            break;
        }
    }
#if defined(KTH_STOP) && KTH_STOP == 2
    return;
#endif
    // unless the ranking above has named it: the index of the need_s-th particle (in index order) equal to T
    const unsigned long long T = prefix_s;
    const long room = need_s;
    long seen = 0;
#pragma unroll
    for (long r = 0; r < (single ? 0 : rounds); ++r) {
        const long j = r * 1024 + tid;
        const unsigned long long key = key_at(r);
        const bool eq = (j < P_all) && key == T;
        // rank of this particle among the ties, in index order: ballot inside the wavefront + wavefront totals
        const unsigned long long m = __ballot(eq);
        const int lane = tid & 63;
        const int within = __popcll(m & ((lane == 63 ? 0ull : (1ull << (lane + 1))) - 1ull));     // inclusive
        if (lane == 0) wtot[tid >> 6] = __popcll(m);
        __syncthreads();
        long before = seen, n = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < (tid >> 6)) before += wtot[w];
            n += wtot[w];
        }
        if (eq && before + within == room) cut_s = j;
        seen += n;
        __syncthreads();
    }
    if (tid == 0 && (!MOM || blockIdx.x == 0)) { thr[0] = T; thr[1] = 0ull; thr[2] = (unsigned long long)(cut_s < 0 ? P_all : cut_s); }
    // With the keys in registers the elite LIST of the local block [offset, offset + P_local) follows at once (what
    // elite_list_kernel does from memory for the streamed instantiation): ballots per round, one scan of their counts.
    if constexpr (NPER > 0) {
        if (!list) return;
        __shared__ int lcnt[NPER * 16];
        const int cut = (int)(cut_s < 0 ? P_all : cut_s), pall = (int)P_all;        // (NPER > 0: at most 32 768 keys)
        const unsigned ploc = (unsigned)P_local;
        const int lane = tid & 63, wave = tid >> 6, jrel = tid - (int)offset;
        unsigned flags = 0u;
#pragma unroll
        for (int r = 0; r < NPER; ++r) {
            const int j = r * 1024 + tid;
            const bool e = j < pall && (unsigned)(r * 1024 + jrel) < ploc && (keys[r] < T || (keys[r] == T && j <= cut));
            const unsigned long long m = __ballot(e);
            if (lane == 0) lcnt[r * 16 + wave] = __popcll(m);
            flags |= e ? (1u << r) : 0u;
        }
        __syncthreads();
        constexpr int n = NPER * 16;            // <= 1024: one entry per thread
        const int own = tid < n ? lcnt[tid] : 0;
        int incl = own;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        __syncthreads();                        // (wtot is reused: the tie walk's last reads are behind this barrier)
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int at = incl - own, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) at += (int)wtot[w];
            total += (int)wtot[w];
        }
        if (tid < n) lcnt[tid] = at;
        __syncthreads();
        __shared__ int head[CEM_HEAD], mine[CEM_E_MAX];
#pragma unroll
        for (int r = 0; r < NPER; ++r) {
            const bool e = (flags >> r) & 1u;
            const unsigned long long m = __ballot(e);
            if (e) {
                const int pos = lcnt[r * 16 + wave] + __popcll(m & ((1ull << lane) - 1ull));
                const int idx = r * 1024 + jrel;
                if (!MOM || blockIdx.x == 0) list[pos] = idx;
                if constexpr (MOM) {
                    if (pos < CEM_HEAD) head[pos] = idx;
                    const int rel = pos - (int)blockIdx.x * mo.E;
                    if (rel >= 0 && rel < mo.E) mine[rel] = idx;
                }
            }
        }
        if (tid == 0 && (!MOM || blockIdx.x == 0)) *count = total;
#if defined(KTH_STOP) && KTH_STOP == 3
        return;
#endif
        if constexpr (MOM) {
            extern __shared__ double dyn[];         // tile[E * HA] | red[S * AA] | cdiff[HA] | cprime[A]
            const int H = mo.H, A = mo.A, HA = H * A, AA = A * A, E = mo.E;
            double* tile = dyn;
            double* red = tile + E * HA;
            double* cdiff = red + (1024 / AA) * AA;
            double* cpr = cdiff + HA;
            __syncthreads();
            const int n = total, e0 = (int)blockIdx.x * E;
            const int ne = n - e0 < 0 ? 0 : (n - e0 < E ? n - e0 : E);
            const int nh = n < CEM_HEAD ? n : CEM_HEAD;
            // the provisional centre: mean of the first elite rows (every workgroup alike), as a mean DELTA per channel
            // (HA entries x nh rows over 1024 threads: a thread takes one entry and a quarter of the rows - independent loads -
            // and the quarters are added in a fixed order)
            {
                double* part = tile;                    // [4][HA] (the tile is free until the rows are staged)
                const int q4 = tid / 256, j0 = tid - q4 * 256;
                for (int j = j0; j < HA; j += 256) {
                    double c = 0.0;
                    for (int e = q4; e < nh; e += 4) c += (double)mo.actions[(long)head[e] * HA + j];
                    part[q4 * HA + j] = c;
                }
                __syncthreads();
                for (int j = tid; j < HA; j += 1024) {
                    const double c = (part[j] + part[HA + j]) + (part[2 * HA + j] + part[3 * HA + j]);
                    cdiff[j] = (nh > 0 ? c / (double)nh : 0.0) - mo.mean[j];
                }
            }
            __syncthreads();
#if defined(KTH_STOP) && KTH_STOP == 4
            return;
#endif
            if (tid < A) {
                double c = 0.0;
                for (int t = 0; t < H; ++t) c += cdiff[t * A + tid];
                cpr[tid] = c / (double)H;
                if (blockIdx.x == 0) mo.cprime[tid] = cpr[tid];
            }
            if (blockIdx.x == 0) {      // what cem_finish reads while it rewrites mean / cov / the step counter
                for (int j = tid; j < HA; j += 1024) mo.mean_prev[j] = mo.mean[j];
                for (int j = tid; j < AA; j += 1024) mo.cov_prev[j] = mo.cov[j];
                if (tid == 0) *mo.step_prev = mo.d_step ? *mo.d_step : 0ll;
            }
            __syncthreads();
            double* out = mo.partial + (long)blockIdx.x * (1 + HA + AA);
            if (tid == 0) out[0] = (double)ne;
            // my rows: staged once (deltas about the mean and the provisional centre), summed per entry, scattered
            for (int w = tid; w < ne * HA; w += 1024) {
                const int e = w / HA, j = w - e * HA;
                tile[w] = (double)mo.actions[(long)mine[e] * HA + j] - mo.mean[j] - cpr[j % A];
            }
            __syncthreads();
            for (int j = tid; j < HA; j += 1024) {          // sum of the action rows = sum of deltas + ne (mean + c)
                double sacc = 0.0;
                for (int e = 0; e < ne; ++e) sacc += tile[e * HA + j];
                out[1 + j] = sacc + (double)ne * (mo.mean[j] + cpr[j % A]);
            }
            const int S = 1024 / AA, rows = ne * H;
            if (tid < S * AA) {
                const int sl = tid / AA, idx = tid - sl * AA, i = idx / A, kk = idx - i * A;
                double sacc = 0.0;
                for (int r = sl; r < rows; r += S) sacc += tile[r * A + i] * tile[r * A + kk];
                red[tid] = sacc;
            }
            __syncthreads();
            if (tid < AA) {
                double sacc = 0.0;
                for (int sl = 0; sl < S; ++sl) sacc += red[sl * AA + tid];
                out[1 + HA + tid] = sacc;
            }
        }
    }
}

// list[0..n) = the local indices of the elite particles in index order, *count = n: the moment kernels below then read
// elite rows only (a tenth of the population at elite_frac 0.1) with independent loads, where a walk over the flags of
// all particles paid one dependent trip to memory per elite.  One workgroup; coalesced rounds of 1024 particles, the
// wavefront ballots' popcounts scanned once (dynamic LDS: 16 counts per round), then every elite lane knows its slot.
__global__ __launch_bounds__(1024) void elite_list_kernel(const double* __restrict__ q_local, long P_local, long offset,
                                                          const unsigned long long* __restrict__ thr,
                                                          int* __restrict__ list, int* __restrict__ count) {
    extern __shared__ int cnt[];            // [rounds][16] ballot popcounts -> exclusive prefixes
    __shared__ int wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rounds = (int)((P_local + 1023) / 1024), n = rounds * 16;
    const unsigned long long T = thr[0];
    const long cut = (long)thr[2];
    auto is_elite = [&](long i) {
        if (i >= P_local) return false;
        const unsigned long long key = order_key(q_local[i]);
        return key < T || (key == T && offset + i <= cut);
    };
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long m = __ballot(is_elite((long)r * 1024 + tid));
        if (lane == 0) cnt[r * 16 + wave] = __popcll(m);
    }
    __syncthreads();
    // exclusive scan of the n counts in (round, wavefront) order: a run of `per` entries per thread
    const int per = (n + 1023) / 1024, c0 = tid * per;
    int own = 0;
    for (int k = 0; k < per; ++k) own += c0 + k < n ? cnt[c0 + k] : 0;
    int incl = own;
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int at = incl - own, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) at += wtot[w];
        total += wtot[w];
    }
    for (int k = 0; k < per; ++k)
        if (c0 + k < n) {
            const int v = cnt[c0 + k];
            cnt[c0 + k] = at;
            at += v;
        }
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const long i = (long)r * 1024 + tid;
        const bool e = is_elite(i);
        const unsigned long long m = __ballot(e);
        if (e) list[cnt[r * 16 + wave] + __popcll(m & ((1ull << lane) - 1ull))] = (int)i;
    }
    if (tid == 0) *count = total;
}

// partial[b] = { number of elite rows in block b, sum of their action rows [H*A] }   (E list entries per workgroup)
template <typename T>
__global__ void elite_rows_sum_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                      const T* __restrict__ actions, int HA, int E, double* __restrict__ partial) {
    const int n = *count, e0 = blockIdx.x * E;
    if (e0 >= n && blockIdx.x > 0) return;      // (its partial is not read: ordered_sum_counted_kernel)
    const int ne = n - e0 < 0 ? 0 : (n - e0 < E ? n - e0 : E);
    double* out = partial + (long)blockIdx.x * (1 + HA);
    if (threadIdx.x == 0) out[0] = (double)ne;
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        double s = 0.0;
        for (int e = 0; e < ne; ++e) s += (double)actions[(long)list[e0 + e] * HA + j];
        out[1 + j] = s;
    }
}

// partial[b] = sum over block b's elite rows and all steps of (d - dm)(d - dm)' [A*A], for tiles that fit LDS: the E x H
// deltas are staged once, every thread owns one (i, k) entry and one slice of the E * H rows, and the slices are added
// in a fixed order
template <typename T>
__global__ __launch_bounds__(1024) void elite_rows_scatter_lds_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                                                      const T* __restrict__ actions,
                                                                      const double* __restrict__ mean,
                                                                      const double* __restrict__ dmean, int H, int A, int E,
                                                                      double* __restrict__ partial) {
    extern __shared__ double tile[];        // d[E * H][A] | red[S][A * A]
    const int n = *count, e0 = blockIdx.x * E, HA = H * A, AA = A * A;
    if (e0 >= n && blockIdx.x > 0) return;
    const int ne = n - e0 < 0 ? 0 : (n - e0 < E ? n - e0 : E);
    for (int w = threadIdx.x; w < ne * HA; w += blockDim.x) {
        const int e = w / HA, j = w - e * HA;
        tile[w] = (double)actions[(long)list[e0 + e] * HA + j] - mean[j] - dmean[j % A];
    }
    __syncthreads();
    const int S = (int)blockDim.x / AA, rows = ne * H;
    double* red = tile + E * HA;
    if ((int)threadIdx.x < S * AA) {
        const int sl = threadIdx.x / AA, idx = threadIdx.x - sl * AA, i = idx / A, k = idx - i * A;
        double s = 0.0;
        for (int r = sl; r < rows; r += S) s += tile[r * A + i] * tile[r * A + k];
        red[threadIdx.x] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < AA) {
        double s = 0.0;
        for (int sl = 0; sl < S; ++sl) s += red[sl * AA + threadIdx.x];
        partial[(long)blockIdx.x * AA + threadIdx.x] = s;
    }
}

// partial[b * TQ + tq] = sum over block b's elite rows and the steps t = tq, tq + TQ, ... of (d - dm)(d - dm)' [A*A]
template <typename T>
__global__ void elite_rows_scatter_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                          const T* __restrict__ actions, const double* __restrict__ mean,
                                          const double* __restrict__ dmean, int H, int A, int E, int TQ,
                                          double* __restrict__ partial) {
    const int n = *count, e0 = blockIdx.x * E, HA = H * A, AA = A * A;
    if (e0 >= n && blockIdx.x > 0) return;
    const int ne = n - e0 < 0 ? 0 : (n - e0 < E ? n - e0 : E);
    for (int w = threadIdx.x; w < AA * TQ; w += blockDim.x) {
        const int tq = w / AA, idx = w - tq * AA, i = idx / A, k = idx - i * A;
        const double di = dmean[i], dk = dmean[k];
        double s = 0.0;
        for (int e = 0; e < ne; ++e) {
            const T* ap = actions + (long)list[e0 + e] * HA;
            for (int t = tq; t < H; t += TQ)
                s += ((double)ap[t * A + i] - mean[t * A + i] - di) * ((double)ap[t * A + k] - mean[t * A + k] - dk);
        }
        partial[((long)blockIdx.x * TQ + tq) * AA + idx] = s;
    }
}

// sum over the partials of the workgroups that held elite rows: ceil(*count / E) * mult of them (at least mult)
__global__ void ordered_sum_counted_kernel(const double* __restrict__ partial, const int* __restrict__ count, int E,
                                           int mult, int rec, double* __restrict__ out) {
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= rec) return;
    const int n = *count, nbe = n > 0 ? (n + E - 1) / E : 1;
    const double s = wave_entry_sum(partial, nbe * mult, rec, j);
    if ((threadIdx.x & 63) == 0) out[j] = s;
}

// records[g] = {count, sum a[H*A]} -> elite_mean[H*A], dmean[A] = mean over (H * k) elite deltas
__global__ void cem_mean_kernel(const double* __restrict__ records, int G, int H, int A,
                                const double* __restrict__ mean, double* __restrict__ elite_mean,
                                double* __restrict__ dmean) {
    const int HA = H * A;
    double cnt = 0.0;
    for (int g = 0; g < G; ++g) cnt += records[(long)g * (1 + HA)];
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        double s = 0.0;
        for (int g = 0; g < G; ++g) s += records[(long)g * (1 + HA) + 1 + j];
        elite_mean[j] = s / cnt;
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < H; ++t) s += elite_mean[t * A + a] - mean[t * A + a];
        dmean[a] = s / (double)H;
    }
}

// cem.py:76-86: diag -> np.var (ddof 0), full -> np.cov (ddof 1) over the H*k elite deltas
__global__ void cem_final_kernel(const double* __restrict__ crec, int G, int H, int A, double n_elite, int full,
                                 double step, const double* __restrict__ elite_mean, double* __restrict__ mean,
                                 double* __restrict__ cov) {
    const int HA = H * A;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < A * A) {
        const int i = j / A, k = j % A;
        double C = 0.0;
        for (int g = 0; g < G; ++g) C += crec[(long)g * A * A + j];
        const double N = (double)H * n_elite;
        double upd = full ? C / (N - 1.0) : (i == k ? C / N : 0.0);
        cov[j] = (1.0 - step) * cov[j] + step * upd;
    }
    if (j < HA) mean[j] = (1.0 - step) * mean[j] + step * elite_mean[j];
}

// Sharded CEM in ONE record exchange after the q0 gather (SURVEY 8e budgets two collectives per iteration): every
// GPU sends  rec_g = { n_g | sum_elite a [H*A] | S_g [A*A] }  with S_g the scatter of ITS elite deltas about ITS OWN
// mean delta mu_g; the pooled scatter about the global mean mu follows from the pairwise-variance identity
//   S = sum_g S_g + sum_g (H n_g) (mu_g - mu)(mu_g - mu)'
// which is as stable as the two-pass formula (no difference of large squares) and is exactly the two-pass result on
// one GPU (mu_0 == mu, evaluated by the same expression).  One workgroup: new mean, new covariance (cem.py:76-86).
__global__ void cem_combine_kernel(const double* __restrict__ rec, int G, int H, int A, double n_elite, int full,
                                   double step, double* __restrict__ mean, double* __restrict__ cov) {
    extern __shared__ double sh[];          // dsum[G][A] | mu[A]
    const int HA = H * A, R = 1 + HA + A * A;
    double* dsum = sh;
    double* mu = sh + G * A;
    double cnt = 0.0;
    for (int g = 0; g < G; ++g) cnt += rec[(long)g * R];
    for (int idx = threadIdx.x; idx < G * A; idx += blockDim.x) {
        const int g = idx / A, a = idx % A;
        const double* r = rec + (long)g * R;
        double sacc = 0.0;
        for (int t = 0; t < H; ++t) sacc += r[1 + t * A + a] / r[0] - mean[t * A + a];     // elite_mean_g - mean, as cem_mean_kernel
        dsum[idx] = r[0] > 0.0 ? sacc / (double)H : 0.0;                                   // mu_g[a]
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        if (G == 1) { mu[a] = dsum[a]; continue; }
        double sacc = 0.0;
        for (int t = 0; t < H; ++t) {
            double e = 0.0;
            for (int g = 0; g < G; ++g) e += rec[(long)g * R + 1 + t * A + a];
            sacc += e / cnt - mean[t * A + a];
        }
        mu[a] = sacc / (double)H;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < A * A; j += blockDim.x) {
        const int i = j / A, k = j % A;
        double C = 0.0;
        for (int g = 0; g < G; ++g) {
            const double* r = rec + (long)g * R;
            C += r[1 + HA + j];
            if (G > 1) C += (double)H * r[0] * (dsum[g * A + i] - mu[i]) * (dsum[g * A + k] - mu[k]);
        }
        const double N = (double)H * n_elite;
        const double upd = full ? C / (N - 1.0) : (i == k ? C / N : 0.0);
        cov[j] = (1.0 - step) * cov[j] + step * upd;
    }
    __syncthreads();                        // every read of the old mean is done
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        double e = 0.0;
        for (int g = 0; g < G; ++g) e += rec[(long)g * R + 1 + j];
        mean[j] = (1.0 - step) * mean[j] + step * (e / cnt);
    }
}

// ---- the fused CEM step (round 4): cem_select_moments (above) -> [cem_record -> all-gather] -> cem_finish -----------------
// cem_record (sharded runs): the workgroups' partials {n_b, sum a, scatter about the provisional centre c} -> THIS GPU's
// record {n_g | sum a [H*A] | S_g [A*A]}, S_g the scatter about its OWN mean delta mu_g = what cem_combine_kernel pools:
// S_g = S_c - (H n_g) (mu_g - c)(mu_g - c)'.  One workgroup.
__global__ void cem_record_kernel(const double* __restrict__ partial, int NB, int H, int A, const double* __restrict__ mean,
                                  const double* __restrict__ cprime, double* __restrict__ rec) {
    extern __shared__ double sh[];          // sum[HA] | d[A]
    const int HA = H * A, AA = A * A, R = 1 + HA + AA;
    const double cnt = sum_strided(partial, NB, R);
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        const double sacc = sum_strided(partial + 1 + j, NB, R);
        sh[j] = sacc;
        rec[1 + j] = sacc;
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        double sacc = 0.0;
        for (int t = 0; t < H; ++t) sacc += sh[t * A + a] / cnt - mean[t * A + a];
        sh[HA + a] = cnt > 0.0 ? sacc / (double)H - cprime[a] : 0.0;       // mu_g - c
    }
    __syncthreads();
    for (int j = threadIdx.x; j < AA; j += blockDim.x) {
        const double C = sum_strided(partial + 1 + HA + j, NB, R);
        rec[1 + HA + j] = C - (double)H * cnt * sh[HA + j / A] * sh[HA + j % A];
    }
    if (threadIdx.x == 0) rec[0] = cnt;
}

// cem_finish: everything behind the moments in ONE launch - refit of mean and covariance (cem.py:76-86: np.var ddof 0 on
// the diagonal / np.cov ddof 1), the covariance's growth of the shift (cem.py:94), its Cholesky factor, the action
// (device copy + mapped host copy), the horizon shift, the step counter, and the raw Philox samples of the NEXT control
// step drawn with the new factor (the sampler kernel's stream, sample for sample).  Every workgroup forms the new
// covariance and its factor itself (a few KB of partials / records, A <= 8); workgroup 0 writes the state; all of them
// then draw.  They read mean / cov / step counter from the snapshots the selection launch left, never from the
// buffers workgroup 0 rewrites.  mode 0: `in` = NB partials about the provisional centre (one GPU); mode 1: `in` = G
// gathered records about their own means (cem_combine_kernel's pooling).
struct CemFinish {
    const double *in, *cprime, *mean_prev, *cov_prev, *grow_diag;
    const long long* step_prev;
    double *mean, *cov, *chol, *action_out, *action_host;
    long long* step_counter;
    int* status;
    void* noise;
    int n_in, mode, H, A, full, shift_mode;
    double n_elite, step, grow_scale;
    unsigned long long seed, offset;
    long particle_offset, P;
};
constexpr int CEM_FIN_THREADS = 256;
template <typename T>
__global__ __launch_bounds__(CEM_FIN_THREADS) void cem_finish_kernel(CemFinish f) {
    extern __shared__ double sh[];          // sumA[HA] | mu[(G + 1) * A] | C[AA] | L[AA] | per-wave store tiles
    const int H = f.H, A = f.A, HA = H * A, AA = A * A, R = 1 + HA + AA, G = f.n_in;
    const int tid = threadIdx.x;
    double* sumA = sh;
    double* mu = sumA + HA;                 // mode 0: mu - c [A]; mode 1: mu_g [G][A], then mu [A]
    double* Cn = mu + (G + 1) * A;
    double* L = Cn + AA;
    T* tiles = (T*)(L + AA);
    const double cnt = sum_strided(f.in, G, R);
    for (int j = tid; j < HA; j += CEM_FIN_THREADS) sumA[j] = sum_strided(f.in + 1 + j, G, R);
    __syncthreads();
    // The action (row 0 of the new mean) is known here, before covariance and factor: it goes to mapped pinned host memory
    // at once, followed - once those writes are visible system-wide - by the new step count as a completion flag (one
    // wavefront: lane 0's flag write follows every lane's action write in program order).  The host picks the action up
    // and enqueues the next iteration while this launch is still refitting, drawing, and the env step has not run.
    if (blockIdx.x == 0 && tid < A) {
        const double act = (1.0 - f.step) * f.mean_prev[tid] + f.step * (sumA[tid] / cnt);
        if (f.action_out) f.action_out[tid] = act;
        if (f.action_host) {
            f.action_host[tid] = act;
            __threadfence_system();
            if (tid == 0) { f.action_host[A] = (double)(*f.step_prev + 1); __threadfence_system(); }
        }
    }
    if (f.mode == 0) {
        for (int a = tid; a < A; a += CEM_FIN_THREADS) {
            double sacc = 0.0;
            for (int t = 0; t < H; ++t) sacc += sumA[t * A + a] / cnt - f.mean_prev[t * A + a];
            mu[a] = cnt > 0.0 ? sacc / (double)H - f.cprime[a] : 0.0;
        }
    } else {
        for (int idx = tid; idx < G * A; idx += CEM_FIN_THREADS) {
            const int g = idx / A, a = idx % A;
            const double* r = f.in + (long)g * R;
            double sacc = 0.0;
            for (int t = 0; t < H; ++t) sacc += r[1 + t * A + a] / r[0] - f.mean_prev[t * A + a];
            mu[idx] = r[0] > 0.0 ? sacc / (double)H : 0.0;
        }
        for (int a = tid; a < A; a += CEM_FIN_THREADS) {
            double sacc = 0.0;
            for (int t = 0; t < H; ++t) sacc += sumA[t * A + a] / cnt - f.mean_prev[t * A + a];
            mu[G * A + a] = sacc / (double)H;
        }
    }
    __syncthreads();
    for (int j = tid; j < AA; j += CEM_FIN_THREADS) {
        const int i = j / A, k = j % A;
        double C = sum_strided(f.in + 1 + HA + j, G, R);
        if (f.mode == 1 && G > 1)
            for (int g = 0; g < G; ++g)
                C += (double)H * f.in[(long)g * R] * (mu[g * A + i] - mu[G * A + i]) * (mu[g * A + k] - mu[G * A + k]);
        if (f.mode == 0) C -= (double)H * cnt * mu[i] * mu[k];
        const double N = (double)H * f.n_elite;
        const double upd = f.full ? C / (N - 1.0) : (i == k ? C / N : 0.0);
        double c = (1.0 - f.step) * f.cov_prev[j] + f.step * upd;
        if (i == k) c += f.grow_scale * (f.grow_diag ? f.grow_diag[i] : 1.0);          // cem.py:94
        Cn[j] = c;
        L[j] = k <= i ? c : 0.0;
    }
    __syncthreads();
    // Cholesky factor of the new covariance, as cholesky_kernel: positive SEMI-definite input gets a zero column
    if (tid < 64) {
        double tol = 0.0;
        for (int k = 0; k < A; ++k) tol = fmax(tol, fabs(Cn[k * A + k]));
        tol *= 1e-13;
        const int i = tid;
        for (int k = 0; k < A; ++k) {
            const double d = L[k * A + k];
            const bool dead = !(d > tol);
            __builtin_amdgcn_wave_barrier();
            if (i == k) {
                if ((!(d >= -tol) || !(d == d)) && f.status && blockIdx.x == 0) *f.status = 1;
                L[k * A + k] = dead ? 0.0 : sqrt(d);
            }
            __builtin_amdgcn_wave_barrier();
            if (i > k && i < A) L[i * A + k] = dead ? 0.0 : L[i * A + k] / L[k * A + k];
            __builtin_amdgcn_wave_barrier();
            if (i > k && i < A) for (int j = k + 1; j <= i; ++j) L[i * A + j] -= L[i * A + k] * L[j * A + k];
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int j = tid; j < AA; j += CEM_FIN_THREADS) { f.cov[j] = Cn[j]; if (f.chol) f.chol[j] = L[j]; }
        // new mean, action, shift (olgaussian_mpc.py:69-78, 116-129)
        for (int j = tid; j < HA; j += CEM_FIN_THREADS) sumA[j] = (1.0 - f.step) * f.mean_prev[j] + f.step * (sumA[j] / cnt);
        __syncthreads();
        if (tid == 0 && f.step_counter) *f.step_counter = *f.step_prev + 1;
        for (int j = tid; j < HA; j += CEM_FIN_THREADS) {
            double v = sumA[j];
            if (f.shift_mode >= 0) {
                const int t = j / A, a = j % A;
                if (t + 1 < H) v = sumA[j + A];
                else v = f.shift_mode == 0 ? 0.0 : sumA[(H - 1) * A + a];
            }
            f.mean[j] = v;
        }
    }
    if (!f.noise) return;
    // the next step's raw samples: noise_full_kernel's work items (particle, t-quad), 64 per wavefront per trip, through a
    // per-wavefront LDS tile so that full rows go out (H a multiple of 4)
    const int H4 = (H + 3) / 4, lane = tid & 63, wave = tid >> 6;
    const long items = f.P * H4, nwaves = (long)gridDim.x * (CEM_FIN_THREADS / 64);
    const unsigned long long offset = f.offset + (unsigned long long)(*f.step_prev + 1);
    const bool staged = (H & 3) == 0;
    const int run = 4 * A, pad = run + 1;
    T* tile = tiles + wave * 64 * (4 * A + 1);
    T* noise = (T*)f.noise;
    for (long base = ((long)blockIdx.x * (CEM_FIN_THREADS / 64) + wave) * 64; base < items; base += nwaves * 64) {
        const long gid = base + lane;
        const bool live = gid < items;
        const int t4 = (int)(gid % H4);
        const long p = gid / H4;
        float z[NOISE_MAXA_U][4];
#pragma unroll
        for (int b = 0; b < NOISE_MAXA_U; ++b) {
#pragma unroll
            for (int k = 0; k < 4; ++k) z[b][k] = 0.0f;
            if (b < A && live) normal_quad(f.seed, offset, (unsigned long long)((p + f.particle_offset) * A + b), (unsigned)t4, z[b]);
        }
        const int t = 4 * t4;
#pragma unroll
        for (int a = 0; a < NOISE_MAXA_U; ++a) {
            if (a < A) {
                double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    const double l = L[a * A + b];
                    if (l != 0.0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[k] += l * (double)z[b][k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (staged) tile[lane * pad + k * A + a] = (T)x[k];
                    else if (live && t + k < H) noise[(p * H + t + k) * A + a] = (T)x[k];
                }
            }
        }
        if (staged) {
            __builtin_amdgcn_wave_barrier();
            const long first = base * run, total = f.P * (long)H * A;
            for (int i = lane; i < 64 * run; i += 64) {
                const int th = i / run, off = i - th * run;
                if (first + i < total) noise[first + i] = tile[th * pad + off];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- random shooting --------------------------------------------------------------------------------
// first index of the minimum (np.argmin), single workgroup
__global__ void argmin_kernel(const double* __restrict__ q, long P, double* __restrict__ out_val, long* __restrict__ out_idx) {
    __shared__ double sv[BLK];
    __shared__ long si[BLK];
    double bv = INFINITY;
    long bi = P;
    for (long p = threadIdx.x; p < P; p += blockDim.x) {
        const double v = q[p];
        if (v < bv) { bv = v; bi = p; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = BLK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double v = sv[threadIdx.x + s];
            const long i = si[threadIdx.x + s];
            if (v < sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *out_val = sv[0]; *out_idx = si[0]; }
}

// best[g] = {value, global index, action[H*A]} records -> mean update (random_shooting.py:52-62)
__global__ void rs_combine_kernel(const double* __restrict__ records, int G, int HA, double step, double* __restrict__ mean) {
    int best = 0;
    for (int g = 1; g < G; ++g) {
        const double v = records[(long)g * (2 + HA)], b = records[(long)best * (2 + HA)];
        if (v < b || (v == b && records[(long)g * (2 + HA) + 1] < records[(long)best * (2 + HA) + 1])) best = g;
    }
    for (int j = threadIdx.x; j < HA; j += blockDim.x)
        mean[j] = (1.0 - step) * mean[j] + step * records[(long)best * (2 + HA) + 2 + j];
}

template <typename T>
__global__ void rs_record_kernel(const double* __restrict__ val, const long* __restrict__ idx, long offset,
                                 const T* __restrict__ actions, int HA, double* __restrict__ record) {
    if (threadIdx.x == 0) { record[0] = *val; record[1] = (double)(offset + *idx); }
    for (int j = threadIdx.x; j < HA; j += blockDim.x) record[2 + j] = (double)actions[(*idx) * HA + j];
}

__global__ void mean_value_kernel(const double* __restrict__ q, long P, double* __restrict__ sum_out) {
    __shared__ double sm[BLK];
    double s = 0.0;
    for (long p = threadIdx.x; p < P; p += blockDim.x) s += q[p];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *sum_out = sm[0];
}

// OLGaussianMPC._shift (olgaussian_mpc.py:116-129): mode 0 null, 1 repeat, 2 row supplied by the host
__global__ void shift_kernel(double* __restrict__ mean, int H, int A, int mode, const double* __restrict__ row) {
    __shared__ double last[64];
    const int a = threadIdx.x;
    if (a >= A) return;
    for (int t = 0; t + 1 < H; ++t) mean[t * A + a] = mean[(t + 1) * A + a];   // column-wise: no cross-thread hazard
    last[a] = mode == 0 ? 0.0 : (mode == 1 ? (H >= 2 ? mean[(H - 2) * A + a] : mean[a]) : row[a]);
    mean[(H - 1) * A + a] = last[a];
}

// The tail of a control step in one launch: read the action out (mean[0] -> device copy and mapped host copy), shift the
// horizon, advance the device step counter, and grow the covariance by scale * diag(d) (the shift of CEM / DMD-MPC with
// update_cov) - five stream operations of ~5 us each when issued one by one.  Any of the outputs may be null.
__global__ void step_tail_kernel(double* __restrict__ mean, int H, int A, int mode, const double* __restrict__ row,
                                 double* __restrict__ action_out, double* __restrict__ action_host,
                                 long long* __restrict__ step_counter, double* __restrict__ cov,
                                 const double* __restrict__ d, double scale) {
    const int a = threadIdx.x;
    long long count = 0;
    if (a == 0 && step_counter) count = (*step_counter += 1);
    if (a >= A) return;
    const double act = mean[a];
    if (action_out) action_out[a] = act;
    // mapped pinned host memory: the action, then - behind a system-scope fence - the new step count as a completion flag
    // (one wavefront: lane 0's flag write follows every lane's action write in program order), so that the host can pick
    // the action up while the rest of the captured iteration (the real env's step) is still running
    if (action_host) {
        action_host[a] = act;
        __threadfence_system();
        if (a == 0) { action_host[A] = (double)count; __threadfence_system(); }
    }
    for (int t = 0; t + 1 < H; ++t) mean[t * A + a] = mean[(t + 1) * A + a];
    mean[(H - 1) * A + a] = mode == 0 ? 0.0 : (mode == 1 ? (H >= 2 ? mean[(H - 2) * A + a] : act) : row[a]);
    if (cov) cov[a * A + a] += scale * (d ? d[a] : 1.0);
}

// Device-resident covariance (CEM, DMD-MPC with update_cov): the factor the sampler colours its normals with is
// computed where the covariance lives, so an adapting covariance never leaves the GPU.
// chol = lower Cholesky factor of cov (numpy.linalg.cholesky in control_utils.generate_noise's place); one
// workgroup, column by column.  The reference samples with np.random.multivariate_normal, whose SVD colouring
// accepts positive SEMI-definite covariances (full-covariance CEM with fewer than A independent elite rows, a
// variance that underflowed): a pivot that vanishes to rounding (|d| <= 1e-13 * largest diagonal entry) gets a zero
// column - a valid factor L L' = cov of the rank-deficient matrix - instead of a NaN factor.  *status = 1 only for a
// genuinely indefinite matrix (pivot below -tolerance) or non-finite input; the host raises on it
// (DeviceUpdater.check_status).
__global__ void cholesky_kernel(const double* __restrict__ cov, int A, double* __restrict__ chol, int* status) {
    __shared__ double L[64 * 64];
    __shared__ double tol;
    const int i = threadIdx.x;
    if (i < A) for (int j = 0; j < A; ++j) L[i * A + j] = j <= i ? cov[i * A + j] : 0.0;
    if (i == 0) {
        double mx = 0.0;
        for (int k = 0; k < A; ++k) mx = fmax(mx, fabs(cov[k * A + k]));
        tol = 1e-13 * mx;
    }
    __syncthreads();
    for (int k = 0; k < A; ++k) {
        const double d = L[k * A + k];          // every lane reads the pivot before lane k overwrites it
        const bool dead = !(d > tol);
        __syncthreads();
        if (i == k) {
            if ((!(d >= -tol) || !(d == d)) && status) *status = 1;
            L[k * A + k] = dead ? 0.0 : sqrt(d);
        }
        __syncthreads();
        if (i > k && i < A) L[i * A + k] = dead ? 0.0 : L[i * A + k] / L[k * A + k];
        __syncthreads();
        if (i > k && i < A) for (int j = k + 1; j <= i; ++j) L[i * A + j] -= L[i * A + k] * L[j * A + k];
        __syncthreads();
    }
    if (i < A) for (int j = 0; j < A; ++j) chol[i * A + j] = L[i * A + j];
}

// cov += scale * diag(d)   (CEM._shift cem.py:94, DMDMPC._shift gaussian_dmd.py:111-112); d == nullptr: identity
__global__ void cov_add_diag_kernel(double* __restrict__ cov, int A, const double* __restrict__ d, double scale) {
    const int i = threadIdx.x;
    if (i < A) cov[i * A + i] += scale * (d ? d[i] : 1.0);
}

// ---- fused MPPI fast path (time_based_weights off, control cost off, no covariance update) -------------
// Two launches instead of six: per-workgroup softmax partials straight from the cost-to-go the rollout
// kernel already produced, then ONE workgroup that merges them (max-rescaled, fixed order), updates the
// mean, extracts the action and shifts the horizon.
constexpr int FCH = 64;     // particles per workgroup

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// partial[b] = { m_b, S_b, W_b[H*A] } with weights exp(x_p - m_b), x_p = -q0_p / lam
template <typename T>
__global__ void fused_partial_kernel(const double* __restrict__ q0, const T* __restrict__ actions, double lam, long P,
                                     int HA, double* __restrict__ partial, const long long* __restrict__ d_step,
                                     long long* __restrict__ step_snapshot) {
    __shared__ double e_s[FCH];
    const long p0 = (long)blockIdx.x * FCH;
    const int n = (int)((P - p0) < FCH ? (P - p0) : FCH);
    double* out = partial + (long)blockIdx.x * (2 + HA);
    // the step this iteration belongs to, for the sampler workgroups of the NEXT launch (which runs while that
    // launch's first workgroup advances the counter)
    if (step_snapshot && blockIdx.x == 0 && threadIdx.x == 0) *step_snapshot = d_step ? *d_step : 0;
    if (threadIdx.x < 64) {
        const double x = threadIdx.x < n ? (-1.0 / lam) * q0[p0 + threadIdx.x] : -INFINITY;
        const double m = wave_max(x);
        const double e = threadIdx.x < n ? exp(x - m) : 0.0;
        e_s[threadIdx.x] = e;
        const double s = wave_sum(e);
        if (threadIdx.x == 0) { out[0] = m; out[1] = s; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        // eight loads in flight per lane; the summation order over particles stays 0, 1, 2, ...
        double acc = 0.0;
        int p = 0;
        for (; p + 8 <= n; p += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (double)actions[(p0 + p + u) * HA + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += e_s[p + u] * v[u];
        }
        for (; p < n; ++p) acc += e_s[p] * (double)actions[(p0 + p) * HA + j];
        out[2 + j] = acc;
    }
}

// merge partials -> mean update (mppi.py:69-82) -> action = mean[0] (olgaussian_mpc.py:71) -> shift
// (olgaussian_mpc.py:116-129; shift_mode < 0: no shift).  Optionally leaves the GPU record
// [xmax | S | W] for the multi-GPU combine and the value -lam logsumexp (mppi.py:113-131).
// Workgroups past the first belong to a different job riding in the same launch: they draw the raw samples of the
// NEXT control step (the noise buffer is free once the rollout has finished).  Workgroup 0 publishes the action
// first, so the sampler runs in the shadow of the host's round trip instead of on the critical path.
template <typename T>
__global__ void fused_final_kernel(const double* __restrict__ partial, int nb, int H, int A, double lam, double step,
                                   int shift_mode, double P_total, double* __restrict__ mean,
                                   double* __restrict__ action_out, double* __restrict__ record,
                                   double* __restrict__ value, double* __restrict__ action_host,
                                   long long* __restrict__ step_counter, NextNoise nn, long P,
                                   const long long* __restrict__ step_snapshot) {
    if (blockIdx.x > 0) {
        noise_element<T>((T*)nn.noise, ((long)blockIdx.x - 1) * blockDim.x + threadIdx.x, P, H, A, nn.chol, nn.seed,
                         nn.offset + (unsigned long long)*step_snapshot, nn.particle_offset, nn.diag_only);
        return;
    }
    // sc[nb] | ss[nb] | nm[H*A] | red[32] | part[nsl * H*A].  A workgroup of 1024 threads (launches with many partials:
    // the merge is a chain of nb / 8 trips to L2 per entry, 16 us of the critical path at 256 partials) splits the partials
    // into nsl = 4 slices, one per 256 threads, and adds the slices in order; 256 threads keep one slice and the order of
    // the earlier rounds.  Either way the order is fixed: bit-reproducible.
    extern __shared__ double sh[];
    const int HA = H * A, rec = 2 + HA, tid = threadIdx.x, nth = blockDim.x, nw = nth >> 6;
    const int nsl = nth >= 512 ? nth / 256 : 1, sl = nsl > 1 ? tid / 256 : 0, jt = nsl > 1 ? tid % 256 : tid, jstep = nsl > 1 ? 256 : nth;
    double* sc = sh;
    double* ss = sh + nb;
    double* nm = ss + nb;
    double* red = nm + HA;
    double* part = red + 32;
    double m = -INFINITY;
    for (int b = tid; b < nb; b += nth) m = fmax(m, partial[(long)b * rec]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    double M = red[0];
    for (int w = 1; w < nw; ++w) M = fmax(M, red[w]);
    for (int b = tid; b < nb; b += nth) {
        sc[b] = exp(partial[(long)b * rec] - M);
        ss[b] = partial[(long)b * rec + 1];
    }
    __syncthreads();
    double S = 0.0;
    if (nsl == 1) {
        for (int b = 0; b < nb; ++b) S += sc[b] * ss[b];                        // every thread, same order
    } else {                    // (1024 partials: the serial walk is 8 us of LDS round trips) strided sums, then a fixed tree
        double sloc = 0.0;
        for (int b = tid; b < nb; b += nth) sloc += sc[b] * ss[b];
        sloc = wave_sum(sloc);
        if ((tid & 63) == 0) red[16 + (tid >> 6)] = sloc;
        __syncthreads();
        for (int w = 0; w < nw; ++w) S += red[16 + w];
    }
    const int per = (nb + nsl - 1) / nsl, b0 = sl * per, b1 = b0 + per < nb ? b0 + per : nb;
    for (int j = jt; j < HA; j += jstep) {
        double W = 0.0;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {                   // eight loads in flight; same order of summation
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(long)(b + u) * rec + 2 + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) W += sc[b + u] * v[u];
        }
        for (; b < b1; ++b) W += sc[b] * partial[(long)b * rec + 2 + j];
        if (nsl > 1) {
            part[sl * HA + j] = W;
        } else {
            if (record) record[2 + j] = W;
            nm[j] = (1.0 - step) * mean[j] + step * (W / S);
        }
    }
    if (nsl > 1) {
        __syncthreads();
        for (int j = tid; j < HA; j += nth) {
            double W = part[j];
            for (int q = 1; q < nsl; ++q) W += part[q * HA + j];
            if (record) record[2 + j] = W;
            nm[j] = (1.0 - step) * mean[j] + step * (W / S);
        }
    }
    if (threadIdx.x == 0) {
        if (record) { record[0] = M; record[1] = S; }
        if (value) *value = -lam * (log(S / P_total) + M);
    }
    __syncthreads();
    if (action_out && threadIdx.x < A) action_out[threadIdx.x] = nm[threadIdx.x];
    // mapped pinned host memory: the action, then - once every lane's write is visible system-wide - the new step
    // count as a completion flag, so that the host can pick the action up without waiting for the stream to drain
    if (action_host && threadIdx.x < A) {
        action_host[threadIdx.x] = nm[threadIdx.x];
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long count = 0;
        if (step_counter) count = (*step_counter += 1);                               // noise stream of the next step
        if (action_host) {
            action_host[A] = (double)count;
            __threadfence_system();
        }
    }
    for (int j = threadIdx.x; j < HA; j += blockDim.x) {
        double v = nm[j];
        if (shift_mode >= 0) {
            const int t = j / A, a = j % A;
            if (t + 1 < H) v = nm[j + A];
            else v = shift_mode == 0 ? 0.0 : nm[(H - 1) * A + a];      // 'null' / 'repeat' (the row that was last)
        }
        mean[j] = v;
    }
}

inline int nblocks(long n, int b) { return (int)((n + b - 1) / b); }

}  // namespace

// ---- host-side launchers ------------------------------------------------------------------------------
constexpr int CHUNK = 16;      // particles per workgroup in the partial-moment kernels

long update_workspace_doubles(long P, int H, int A) {
    const long Hw = H, HA = (long)H * A, rec = Hw + HA + (long)A * A;
    const long nb = (P + CHUNK - 1) / CHUNK;
    return HA /*un*/ + P * Hw /*x*/ + P /*q0*/ + Hw /*xmax*/ + nb * rec /*partials*/ + 64 + (P + 1) / 2 /*elite ints*/ + 2 * HA + 2 * A;
}

struct Ws {
    double *un, *x, *q0, *xmax, *partial, *scratch, *elite_mean, *dmean;
    int* elite;
    Ws(double* base, long P, int H, int A) {
        const long Hw = H, HA = (long)H * A, rec = Hw + HA + (long)A * A, nb = (P + CHUNK - 1) / CHUNK;
        un = base;
        x = un + HA;
        q0 = x + P * Hw;
        xmax = q0 + P;
        partial = xmax + Hw;
        scratch = partial + nb * rec;
        elite = (int*)(scratch + 64);
        elite_mean = scratch + 64 + (P + 1) / 2;
        dmean = elite_mean + HA;
    }
};

// number of entries of the elite list (elite_list_kernel), behind the selection thresholds in the scratch block
static inline int* elite_count(const Ws& w) { return (int*)(w.scratch + 8); }

template <typename T>
hipError_t traj_cost(const T* costs, const T* actions, const double* mean, const double* covinv, const double* gseq,
                     int gamma_zero, double lam, int alpha, int tbw, long P, int H, int A, double* ws,
                     hipStream_t s) {
    Ws w(ws, P, H, A);
    const double* un = nullptr;
    if (alpha == 0) {
        hipLaunchKernelGGL(un_kernel, dim3(nblocks(H * A, BLK)), dim3(BLK), 0, s, mean, covinv, H, A, w.un);
        un = w.un;
    }
    const int Hw = tbw ? H : 1;
    hipLaunchKernelGGL(traj_cost_kernel<T>, dim3(nblocks(P, BLK)), dim3(BLK), 0, s, costs, actions, mean, un, gseq,
                       gamma_zero, lam, P, H, A, Hw, w.x, w.q0);
    return hipGetLastError();
}

template <typename T>
hipError_t td_lambda_returns(const T* costs, const T* actions, const T* qvals, const double* mean, const double* covinv,
                             const double* wseq, int wseq_zero, double beta, int alpha, double gamma, double td_lam,
                             long P, int H, int A, T* out, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    const double* un = nullptr;
    if (alpha == 0) {
        hipLaunchKernelGGL(un_kernel, dim3(nblocks(H * A, BLK)), dim3(BLK), 0, s, mean, covinv, H, A, w.un);
        un = w.un;
    }
    hipLaunchKernelGGL(td_lambda_kernel<T>, dim3(nblocks(P, BLK)), dim3(BLK), 0, s, costs, actions, qvals, mean, un, wseq,
                       wseq_zero, beta, gamma, td_lam, P, H, A, out);
    return hipGetLastError();
}

template <typename T>
hipError_t softmax_stats(const T* costs, const T* actions, const double* mean, const double* covinv,
                         const double* gseq, int gamma_zero, double lam, int alpha, int tbw, int want_cov, long P,
                         int H, int A, double* record, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    hipError_t e = traj_cost<T>(costs, actions, mean, covinv, gseq, gamma_zero, lam, alpha, tbw, P, H, A, ws, s);
    if (e != hipSuccess) return e;
    const int Hw = tbw ? H : 1, HA = H * A, rec = Hw + HA + A * A;
    const int nb = nblocks(P, CHUNK);
    hipLaunchKernelGGL(colmax_kernel, dim3(Hw), dim3(BLK), 0, s, w.x, P, Hw, w.xmax);
    hipLaunchKernelGGL(softmax_partial_kernel<T>, dim3(nb), dim3(BLK), sizeof(double) * CHUNK * Hw, s, w.x, w.xmax,
                       actions, mean, P, H, A, Hw, CHUNK, want_cov, w.partial);
    hipLaunchKernelGGL(softmax_record_kernel, dim3(nblocks(rec, BLK / 64)), dim3(BLK), 0, s, w.partial, w.xmax, nb, Hw,
                       rec, record);
    return hipGetLastError();
}

hipError_t softmax_combine(const double* records, int G, int H, int A, int tbw, double lam, double step, int cov_mode,
                           double P_total, double* mean, double* cov, double* value, double* wnorm, hipStream_t s) {
    const int Hw = tbw ? H : 1;
    const int n = H * A > A * A ? H * A : A * A;
    hipLaunchKernelGGL(softmax_combine_kernel, dim3(nblocks(n, BLK)), dim3(BLK), 0, s, records, G, H, A, Hw, lam, step,
                       cov_mode, P_total, mean, cov, value, wnorm);
    return hipGetLastError();
}

hipError_t softmax_weights(long P, const double* wnorm, double* ws, int H, int A, double* weights, hipStream_t s) {
    Ws w(ws, P, H, A);
    hipLaunchKernelGGL(softmax_weights_kernel, dim3(nblocks(P, BLK)), dim3(BLK), 0, s, w.x, wnorm, P, weights);
    return hipGetLastError();
}

double* workspace_q0(double* ws, long P, int H, int A) { return Ws(ws, P, H, A).q0; }

template <typename T>
hipError_t cem_elite_sums(const T* actions, const double* q_all, long P_all, long offset, long k, long P, int H, int A,
                          double* record, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    const int HA = H * A;
    const double* qa = q_all ? q_all : w.q0;
    unsigned long long* thr = (unsigned long long*)w.scratch;
    const long Pa = q_all ? P_all : P;
    // the local elite rows as a list (index order), then moments over those rows only
    const long kmax = k < P ? (k > 0 ? k : 0) : P;
    const int nbe = kmax > 0 ? nblocks(kmax, CHUNK) : 1;
    const long off = q_all ? offset : 0;
    if (Pa <= 4 * 1024) {          // (keys per thread by the population: padding rounds cost as much as full ones)
        hipLaunchKernelGGL(kth_key_kernel<4>, dim3(1), dim3(1024), 0, s, qa, Pa, k, thr, off, P, w.elite, elite_count(w));
    } else if (Pa <= 8 * 1024) {
        hipLaunchKernelGGL(kth_key_kernel<8>, dim3(1), dim3(1024), 0, s, qa, Pa, k, thr, off, P, w.elite, elite_count(w));
    } else if (Pa <= 16 * 1024) {
        hipLaunchKernelGGL(kth_key_kernel<16>, dim3(1), dim3(1024), 0, s, qa, Pa, k, thr, off, P, w.elite, elite_count(w));
    } else if (Pa <= 32 * 1024) {
        hipLaunchKernelGGL(kth_key_kernel<32>, dim3(1), dim3(1024), 0, s, qa, Pa, k, thr, off, P, w.elite, elite_count(w));
    } else {
        const size_t list_lds = sizeof(int) * 16 * (size_t)((P + 1023) / 1024);
        if (list_lds > 48 * 1024) return hipErrorInvalidValue;          // (> 786 432 particles on one GPU)
        hipLaunchKernelGGL(kth_key_kernel<0>, dim3(1), dim3(1024), 0, s, qa, Pa, k, thr, 0L, 0L, (int*)nullptr, (int*)nullptr);
        hipLaunchKernelGGL(elite_list_kernel, dim3(1), dim3(1024), list_lds, s, w.q0, P, off, thr, w.elite, elite_count(w));
    }
    hipLaunchKernelGGL(elite_rows_sum_kernel<T>, dim3(nbe), dim3(BLK), 0, s, w.elite, elite_count(w), actions, HA, CHUNK,
                       w.partial);
    hipLaunchKernelGGL(ordered_sum_counted_kernel, dim3(nblocks(1 + HA, BLK / 64)), dim3(BLK), 0, s, w.partial,
                       elite_count(w), CHUNK, 1, 1 + HA, record);
    return hipGetLastError();
}

// ---- the fused CEM step (round 4) -----------------------------------------------------------------------------------------
// elite rows per workgroup of cem_select_moments: as many as a 60 KB tile holds (no opt-in to large dynamic LDS), at most 64
static int cem_fused_rows(int H, int A) {
    const int HA = H * A, AA = A * A;
    const long room = 150 * 1024 / 8 - (1024 / AA) * AA - HA - A;      // (160 KB of LDS per CU; large dynamic LDS is opted in)
    const long e = room / HA;
    // (32 rows per workgroup measured best at 16384 x 32 x 7: more rows lengthen the moments phase of every workgroup - 64:
    // 39.5 us for the launch -, fewer multiply the partials the finish launch sums - 28: 34.7 us but 59 partials)
    const long cap = CEM_E_ROWS < CEM_E_MAX ? CEM_E_ROWS : CEM_E_MAX;
    return (int)(e > cap ? cap : e);
}
bool cem_fused_supported(long P_all, long P, long k, int H, int A) {
    if (A < 1 || A > NOISE_MAXA_U || H < 1 || k < 1 || P_all > 32768 || P < 1 || A * A > 1024 || A * A > H * A + A) return false;
    const int E = cem_fused_rows(H, A);
    return E >= 4 && (k + E - 1) / E <= (P + CHUNK - 1) / CHUNK;        // (the partials fit the workspace's partial area)
}
static inline double* cem_cprime(const Ws& w) { return w.scratch + 16; }
static inline long long* cem_step_prev(const Ws& w) { return (long long*)(w.scratch + 32); }
static inline double* cem_cov_prev(const Ws& w, int A) { return w.dmean + A; }

template <typename T>
hipError_t cem_select_moments(const T* actions, const double* q_all, long P_all, long offset, long k, long P, int H, int A,
                              const double* mean, const double* cov, const long long* d_step, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    const double* qa = q_all ? q_all : w.q0;
    const long Pa = q_all ? P_all : P, off = q_all ? offset : 0;
    if (!cem_fused_supported(Pa, P, k, H, A)) return hipErrorInvalidValue;
    const int E = cem_fused_rows(H, A), HA = H * A, AA = A * A;
    const int NB = (int)((k + E - 1) / E);
    CemMoments<T> mo;
    mo.actions = actions; mo.mean = mean; mo.cov = cov; mo.d_step = d_step;
    mo.partial = w.partial; mo.cprime = cem_cprime(w); mo.mean_prev = w.elite_mean; mo.cov_prev = cem_cov_prev(w, A);
    mo.step_prev = cem_step_prev(w);
    mo.H = H; mo.A = A; mo.E = E;
    const size_t lds = sizeof(double) * ((size_t)(E > 4 ? E : 4) * HA + (1024 / AA) * AA + HA + A);
    unsigned long long* thr = (unsigned long long*)w.scratch;
    // keys per thread by the population: the launch is bound by its per-key instructions, padding rounds cost
    // as much as full ones
#define MJMPC_SELECT_MOMENTS(NPER_)                                                                                          \
    do {                                                                                                                     \
        if (lds > 64 * 1024)                                                                                                 \
            (void)hipFuncSetAttribute((const void*)kth_key_kernel<NPER_, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                             \
        hipLaunchKernelGGL((kth_key_kernel<NPER_, T, true>), dim3(NB), dim3(1024), lds, s, qa, Pa, k, thr, off, P, w.elite,  \
                           elite_count(w), mo);                                                                              \
    } while (0)
    if (Pa <= 4096) MJMPC_SELECT_MOMENTS(4);
    else if (Pa <= 8192) MJMPC_SELECT_MOMENTS(8);
    else if (Pa <= 16384) MJMPC_SELECT_MOMENTS(16);
    else MJMPC_SELECT_MOMENTS(32);
#undef MJMPC_SELECT_MOMENTS
    return hipGetLastError();
}

hipError_t cem_record(long k, long P, int H, int A, const double* mean, double* record, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    const int E = cem_fused_rows(H, A), NB = (int)((k + E - 1) / E);
    hipLaunchKernelGGL(cem_record_kernel, dim3(1), dim3(BLK), sizeof(double) * (H * A + A), s, w.partial, NB, H, A, mean,
                       cem_cprime(w), record);
    return hipGetLastError();
}

template <typename T>
hipError_t cem_finish(const double* records, int G, long k, long P, int H, int A, double n_elite, int full, double step,
                      int shift_mode, double* mean, double* cov, double* chol, int* status, const double* grow_diag,
                      double grow_scale, double* action_out, double* action_host, long long* step_counter, T* noise,
                      unsigned long long seed, unsigned long long offset, long particle_offset, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    const int E = cem_fused_rows(H, A), HA = H * A, AA = A * A;
    CemFinish f;
    f.mode = records ? 1 : 0;
    f.in = records ? records : w.partial;
    f.n_in = records ? G : (int)((k + E - 1) / E);
    f.cprime = cem_cprime(w); f.mean_prev = w.elite_mean; f.cov_prev = cem_cov_prev(w, A); f.grow_diag = grow_diag;
    f.step_prev = cem_step_prev(w);
    f.mean = mean; f.cov = cov; f.chol = chol; f.action_out = action_out; f.action_host = action_host;
    f.step_counter = step_counter; f.status = status; f.noise = (void*)noise;
    f.H = H; f.A = A; f.full = full; f.shift_mode = shift_mode;
    f.n_elite = n_elite; f.step = step; f.grow_scale = grow_scale;
    f.seed = seed; f.offset = offset; f.particle_offset = particle_offset; f.P = P;
    const long items = P * ((H + 3) / 4);
    long nwg = noise ? (items + CEM_FIN_THREADS - 1) / CEM_FIN_THREADS : 1;       // one trip per wavefront up to 1024 workgroups
    if (nwg < 1) nwg = 1;
    if (nwg > 1024) nwg = 1024;
    const size_t lds = sizeof(double) * ((size_t)HA + (f.n_in + 1) * A + 2 * AA) +
                       sizeof(T) * (CEM_FIN_THREADS / 64) * 64 * (4 * A + 1);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)cem_finish_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cem_finish_kernel<T>, dim3((unsigned)nwg), dim3(CEM_FIN_THREADS), lds, s, f);
    return hipGetLastError();
}

template <typename T>
hipError_t cem_elite_cov(const T* actions, const double* mean, const double* sum_records, int G, long P, int H, int A,
                         double* crecord, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    // (every local row may be elite: the list is at most P long; the partials were sized for P / CHUNK workgroups of
    // H + H A + A A entries, which bounds the number of step-slices a workgroup may keep apart)
    const int nbe = nblocks(P, CHUNK) < (1 << 20) ? nblocks(P, CHUNK) : (1 << 20);
    int tq = (H + H * A + A * A) / (A * A);
    tq = tq < 1 ? 1 : (tq > 4 ? 4 : tq);
    tq = tq > H ? H : tq;
    hipLaunchKernelGGL(cem_mean_kernel, dim3(1), dim3(BLK), 0, s, sum_records, G, H, A, mean, w.elite_mean, w.dmean);
    const size_t lds = sizeof(double) * ((size_t)CHUNK * H * A + (size_t)(1024 / (A * A > 1024 ? 1024 : A * A)) * A * A);
    if (A * A <= 1024 && lds <= 48 * 1024) {
        hipLaunchKernelGGL(elite_rows_scatter_lds_kernel<T>, dim3(nbe), dim3(1024), lds, s, w.elite, elite_count(w), actions,
                           mean, w.dmean, H, A, CHUNK, w.partial);
        tq = 1;
    } else {
        hipLaunchKernelGGL(elite_rows_scatter_kernel<T>, dim3(nbe), dim3(BLK), 0, s, w.elite, elite_count(w), actions, mean,
                           w.dmean, H, A, CHUNK, tq, w.partial);
    }
    hipLaunchKernelGGL(ordered_sum_counted_kernel, dim3(nblocks(A * A, BLK / 64)), dim3(BLK), 0, s, w.partial,
                       elite_count(w), CHUNK, tq, A * A, crecord);
    return hipGetLastError();
}

hipError_t cem_final(const double* crecords, int G, long P, int H, int A, double n_elite, int full, double step,
                     double* mean, double* cov, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    const int n = H * A > A * A ? H * A : A * A;
    hipLaunchKernelGGL(cem_final_kernel, dim3(nblocks(n, BLK)), dim3(BLK), 0, s, crecords, G, H, A, n_elite, full, step,
                       w.elite_mean, mean, cov);
    return hipGetLastError();
}

hipError_t cem_combine(const double* records, int G, int H, int A, double n_elite, int full, double step, double* mean,
                       double* cov, hipStream_t s) {
    if (G < 1 || A < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cem_combine_kernel, dim3(1), dim3(BLK), sizeof(double) * (G + 1) * A, s, records, G, H, A, n_elite,
                       full, step, mean, cov);
    return hipGetLastError();
}

template <typename T>
hipError_t rs_best(const T* actions, long offset, long P, int H, int A, double* record, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    long* idx = (long*)(w.scratch + 1);
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(BLK), 0, s, w.q0, P, w.scratch, idx);
    hipLaunchKernelGGL(rs_record_kernel<T>, dim3(1), dim3(BLK), 0, s, w.scratch, idx, offset, actions, H * A, record);
    return hipGetLastError();
}

hipError_t rs_combine(const double* records, int G, int H, int A, double step, double* mean, hipStream_t s) {
    hipLaunchKernelGGL(rs_combine_kernel, dim3(1), dim3(BLK), 0, s, records, G, H * A, step, mean);
    return hipGetLastError();
}

template <typename T>
hipError_t mppi_fused_update(const double* q0, const T* actions, double lam, double step, int shift_mode, long P, int H,
                             int A, double* mean, double* action_out, double* record, double* value, double* ws,
                             hipStream_t s, double* action_host, long long* step_counter, const NextNoise* next) {
    Ws w(ws, P, H, A);
    const int nb = nblocks(P, FCH), HA = H * A;
    if (!q0) q0 = w.q0;
    NextNoise nn{};
    int extra = 0;
    long long* snap = nullptr;
    // many partials (P > 4096): the merging workgroup takes 1024 threads and four slices of the partials
    const int fth = nb > 64 ? 1024 : BLK, nsl = fth >= 512 ? fth / 256 : 1;
    if (next && next->noise) {
        nn = *next;
        extra = nblocks(P * ((H + 3) / 4) * A, fth);
        snap = (long long*)(w.scratch + 8);
    }
    hipLaunchKernelGGL(fused_partial_kernel<T>, dim3(nb), dim3(BLK), 0, s, q0, actions, lam, P, HA, w.partial, nn.d_step,
                       snap);
    const size_t lds = sizeof(double) * (2 * (size_t)nb + HA + 32 + (nsl > 1 ? (size_t)nsl * HA : 0));
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)fused_final_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fused_final_kernel<T>, dim3(1 + extra), dim3(fth), lds, s, w.partial, nb,
                       H, A, lam, step, shift_mode, (double)P, mean, action_out, record, value, action_host, step_counter,
                       nn, P, (const long long*)snap);
    return hipGetLastError();
}

// Sharded runs: the G all-gathered records [xmax | S | W[H*A]] have the layout of the per-workgroup partials, so
// the same kernel merges them (rank order = fixed order -> bit-identical on every rank), updates the mean, reads
// out and publishes the action, advances the step counter and shifts - one launch after the all-gather.
hipError_t mppi_fused_combine(const double* records, int G, double P_total, double lam, double step, int shift_mode,
                              int H, int A, double* mean, double* action_out, double* value, double* action_host,
                              long long* step_counter, hipStream_t s) {
    const int HA = H * A;
    hipLaunchKernelGGL(fused_final_kernel<double>, dim3(1), dim3(BLK), sizeof(double) * (2 * G + HA + 32), s, records, G, H,
                       A, lam, step, shift_mode, P_total, mean, action_out, (double*)nullptr, value, action_host,
                       step_counter, NextNoise{}, 0L, (const long long*)nullptr);
    return hipGetLastError();
}

hipError_t q0_sum(long P, int H, int A, double* out, double* ws, hipStream_t s) {
    Ws w(ws, P, H, A);
    hipLaunchKernelGGL(mean_value_kernel, dim3(1), dim3(BLK), 0, s, w.q0, P, out);
    return hipGetLastError();
}

hipError_t cholesky_lower(const double* cov, int A, double* chol, int* status, hipStream_t s) {
    if (A < 1 || A > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cholesky_kernel, dim3(1), dim3(64), 0, s, cov, A, chol, status);
    return hipGetLastError();
}

hipError_t cov_add_diag(double* cov, int A, const double* d, double scale, hipStream_t s) {
    if (A < 1 || A > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cov_add_diag_kernel, dim3(1), dim3(64), 0, s, cov, A, d, scale);
    return hipGetLastError();
}

hipError_t shift_mean(double* mean, int H, int A, int mode, const double* row, hipStream_t s) {
    if (A > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(shift_kernel, dim3(1), dim3(64), 0, s, mean, H, A, mode, row);
    return hipGetLastError();
}

hipError_t step_tail(double* mean, int H, int A, int mode, const double* row, double* action_out, double* action_host,
                     long long* step_counter, double* cov, const double* d, double scale, hipStream_t s) {
    if (A > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(step_tail_kernel, dim3(1), dim3(64), 0, s, mean, H, A, mode, row, action_out, action_host,
                       step_counter, cov, d, scale);
    return hipGetLastError();
}

#define INST(T)                                                                                                      \
    template hipError_t td_lambda_returns<T>(const T*, const T*, const T*, const double*, const double*,             \
                                             const double*, int, double, int, double, double, long, int, int, T*,    \
                                             double*, hipStream_t);                                                  \
    template hipError_t traj_cost<T>(const T*, const T*, const double*, const double*, const double*, int, double,   \
                                     int, int, long, int, int, double*, hipStream_t);                                \
    template hipError_t softmax_stats<T>(const T*, const T*, const double*, const double*, const double*, int,       \
                                         double, int, int, int, long, int, int, double*, double*, hipStream_t);     \
    template hipError_t cem_elite_sums<T>(const T*, const double*, long, long, long, long, int, int, double*,        \
                                          double*, hipStream_t);                                                     \
    template hipError_t cem_elite_cov<T>(const T*, const double*, const double*, int, long, int, int, double*,       \
                                         double*, hipStream_t);                                                      \
    template hipError_t cem_select_moments<T>(const T*, const double*, long, long, long, long, int, int, const double*, \
                                              const double*, const long long*, double*, hipStream_t);                \
    template hipError_t cem_finish<T>(const double*, int, long, long, int, int, double, int, double, int, double*,   \
                                      double*, double*, int*, const double*, double, double*, double*, long long*, T*, \
                                      unsigned long long, unsigned long long, long, double*, hipStream_t);           \
    template hipError_t rs_best<T>(const T*, long, long, int, int, double*, double*, hipStream_t);                   \
    template hipError_t mppi_fused_update<T>(const double*, const T*, double, double, int, long, int, int, double*,  \
                                             double*, double*, double*, double*, hipStream_t, double*, long long*,   \
                                             const NextNoise*);
INST(float)
INST(double)

}  // namespace mjmpc

// Seed-identical noise on the device: the reference's own random stream, regenerated on the GPU.
//
// control_utils.generate_noise (reference mjmpc/utils/control_utils.py:24-34) draws its samples with
//     np.random.seed(base_seed);  np.random.multivariate_normal(0, c*I, size=(P, H))
// i.e. numpy's LEGACY generator: MT19937 seeded by the Knuth LCG (init_genrand), 53-bit doubles made of two
// consecutive outputs, and the polar (Marsaglia) method with rejection, in C order over (P, H, A).  For an
// isotropic covariance the SVD colouring is the plain scale sqrt(c).
//
// Everything except the Mersenne twister recurrence itself is embarrassingly parallel once one notices that
// every polar ATTEMPT consumes exactly four 32-bit outputs (two doubles), accepted or not:
//   1. mt_stream_kernel     one workgroup walks the recurrence x[n] = x[n-227] ^ twist(x[n-624], x[n-623]).
//                           Unrolled three times, x[n] = x[n-681] ^ tw(n) ^ tw(n-227) ^ tw(n-454), so 623
//                           consecutive words are independent and are produced per barrier, from an LDS ring.
//   2. polar_flags_kernel   attempt i <- words 4i..4i+3: accept iff 0 < r2 < 1; per-workgroup counts
//   3. block_offsets_kernel exclusive scan of the counts
//   4. polar_emit_kernel    the k-th accepted attempt yields normals 2k (f*x2) and 2k+1 (f*x1), f = sqrt(-2 ln r2 / r2)
// All arithmetic that decides acceptance is exact IEEE double (no contraction), so the stream ALIGNMENT is
// identical to numpy's; log() may differ from glibc in the last bit, i.e. the samples agree to <= 1-2 ulp.
#include <hip/hip_runtime.h>

#include "noise_mt.h"

// numpy evaluates the polar method with separately rounded multiplies and adds: no FMA contraction in this
// file.  Plain operators are used on purpose: HIP's __dadd_rn / __dmul_rn are inline functions compiled under
// the headers' own fp-contract=fast and DO get fused after inlining (observed: r2 off by one ulp in 0.1 % of
// the attempts, visible as 1e-12 errors where r2 -> 1).
#pragma clang fp contract(off)

namespace mjmpc {
namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int CHUNK = 623;           // words per barrier (limited by the shortest look-back of twist: n-623)
constexpr int FBLK = 256;

__device__ __forceinline__ unsigned twist(unsigned a, unsigned b) {
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// Twister words, UNTEMPERED (tempering is a pure per-word function and is left to the massively parallel
// consumers: the serial kernel keeps only what is serial).  all[0..624) = the seeded state x[0..624),
// all[624 + j] = x[624 + j] = j-th generated word.
//
// One workgroup produces one SEGMENT of the stream:
//   head launch (1 workgroup, jump_idx == nullptr): seed -> words [0, head_words)
//   body launch (G workgroups): workgroup g owns words [head_words + g*seg_words, +seg_words) and starts from
//     the 624-word state at that offset, obtained WITHOUT walking the stream (jump-ahead, see
//     mjmpc_amd/control/mt_jump.py):  x[J + w] = XOR over the set bits i of (t^J mod phi) of x[i + w];
//     segment 0 starts where the head stopped and simply reads its state.
// The LDS buffer is LINEAR: addresses are one running base + compile-time offsets (no per-read index
// arithmetic); every SPAN steps the last LOOKBACK words are moved back to the front.
constexpr int JSLICES = 8;                     // workgroups sharing one segment's jump reduction
constexpr int JHEAD_MAX = MT_N + 19968;        // words of the head staged in LDS by mt_jump_kernel (82 KB)

constexpr int LOOKBACK = 1078;                 // deepest word the 3x-unrolled recurrence reads
constexpr int SPAN = 22;                       // steps between compactions: LOOKBACK + SPAN * CHUNK words of LDS
constexpr int LINEAR = LOOKBACK + SPAN * CHUNK;

// Jump-ahead reduction: workgroup (g, slice) XORs the windows x[i .. i+624) of its share of segment g's
// set-bit list.  The head of the stream (82 KB) is staged in LDS once per workgroup - the 10^4 x 624 reads per
// segment are then LDS traffic spread over 8 CUs instead of L2 latency in one.
__global__ __launch_bounds__(640) void mt_jump_kernel(const unsigned* __restrict__ all, int head_total,
                                                      const int* __restrict__ jump_idx,
                                                      const int* __restrict__ jump_starts,
                                                      unsigned* __restrict__ jump_states) {
    __shared__ unsigned x[JHEAD_MAX];
    const int g = blockIdx.x / JSLICES, sl = blockIdx.x % JSLICES, tid = threadIdx.x;
    if (g == 0) return;                         // segment 0 continues the head
    {
        const uint4* src = (const uint4*)all;   // head_total is a multiple of 4
        uint4* dst = (uint4*)x;
        for (int i = tid; i < head_total / 4; i += 640) dst[i] = src[i];
    }
    __syncthreads();
    const int lo = jump_starts[g], cnt = jump_starts[g + 1] - lo;
    const int per = (cnt + JSLICES - 1) / JSLICES;
    const int k0 = lo + sl * per, k1 = min(lo + cnt, k0 + per);
    if (tid < MT_N) {
        unsigned acc = 0;
        const unsigned* xt = x + tid;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            const unsigned t0 = xt[jump_idx[k]], t1 = xt[jump_idx[k + 1]], t2 = xt[jump_idx[k + 2]],
                           t3 = xt[jump_idx[k + 3]], t4 = xt[jump_idx[k + 4]], t5 = xt[jump_idx[k + 5]],
                           t6 = xt[jump_idx[k + 6]], t7 = xt[jump_idx[k + 7]];
            acc ^= t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
        }
        for (; k < k1; ++k) acc ^= xt[jump_idx[k]];
        jump_states[(long)blockIdx.x * MT_N + tid] = acc;
    }
}

__global__ __launch_bounds__(640) void mt_segment_kernel(unsigned long long seed, const long long* __restrict__ d_step,
                                                         unsigned* __restrict__ all, long head_words, long seg_words,
                                                         long total_words,
                                                         const unsigned* __restrict__ jump_states) {
    __shared__ unsigned r[LINEAR];
    const int tid = threadIdx.x;
    const bool lane_on = tid < CHUNK;
    constexpr int BASE0 = LOOKBACK - MT_N;      // the 624-word start state lives at r[BASE0 .. LOOKBACK)
    long first, n_out;                          // generated-word range [first, first + n_out) of this workgroup
    if (!jump_states) {
        first = 0;
        n_out = head_words < total_words ? head_words : total_words;
        if (tid == 0) {
            unsigned s = (unsigned)((seed + (d_step ? (unsigned long long)*d_step : 0ull)) & 0xffffffffull);
            for (int pos = 0; pos < MT_N; ++pos) {          // numpy mt19937_seed == init_genrand
                r[BASE0 + pos] = s;
                all[pos] = s;
                s = 1812433253u * (s ^ (s >> 30)) + (unsigned)pos + 1u;
            }
        }
    } else {
        first = head_words + (long)blockIdx.x * seg_words;
        n_out = total_words - first;
        if (n_out > seg_words) n_out = seg_words;
        if (n_out <= 0) return;
        if (tid < MT_N) {
            unsigned acc;
            if (blockIdx.x == 0) {
                acc = all[first + tid];                     // state x[first .. first+624) in `all` indexing
            } else {
                acc = 0;                                    // XOR of the slices mt_jump_kernel reduced
                const unsigned* part = jump_states + (long)blockIdx.x * JSLICES * MT_N + tid;
#pragma unroll
                for (int sl = 0; sl < JSLICES; ++sl) acc ^= part[sl * MT_N];
            }
            r[BASE0 + tid] = acc;
        }
    }
    __syncthreads();
    unsigned* out = all + MT_N + first;
    // first chunk: x[n-227] may itself be new, so the plain recurrence runs in three dependent waves of <= 227
    // words (once per segment); it needs nothing older than the 624-word state
    {
        const int n = MT_N + tid;
        unsigned v = 0;
        for (int ph = 0; ph < 3; ++ph) {
            const int t = tid - ph * 227;
            if (t >= 0 && t < 227 && lane_on && tid < n_out) {
                v = r[BASE0 + n - 227] ^ twist(r[BASE0 + n - 624], r[BASE0 + n - 623]);
                r[BASE0 + n] = v;
            }
            __syncthreads();
        }
        if (lane_on && tid < n_out) out[tid] = v;
    }
    // steady state.  `cur` = LDS slot of the word this lane produces next; the chunk just produced sits at
    // [LOOKBACK, LOOKBACK + CHUNK), i.e. exactly one compaction phase into the buffer.
    int cur = LOOKBACK + CHUNK + tid;
    unsigned* o = out + CHUNK + tid;
    long left = n_out - CHUNK - tid;
    const long steps = n_out > CHUNK ? (n_out - CHUNK + CHUNK - 1) / CHUNK : 0;
    int in_span = 1;
    for (long st = 0; st < steps; ++st) {
        if (in_span == SPAN) {          // move the last LOOKBACK words to the front (640 threads, 2 passes)
            unsigned t0 = 0, t1 = 0;
            const int src = cur - tid - LOOKBACK;                   // first word to keep
            if (tid < LOOKBACK) t0 = r[src + tid];
            if (tid + 640 < LOOKBACK) t1 = r[src + tid + 640];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < LOOKBACK) r[tid] = t0;
            if (tid + 640 < LOOKBACK) r[tid + 640] = t1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur = LOOKBACK + tid;
            in_span = 0;
        }
        if (lane_on && left > 0) {
            const unsigned* q = r + cur;
            const unsigned v = q[-681] ^ twist(q[-624], q[-623]) ^ twist(q[-851], q[-850]) ^ twist(q[-1078], q[-1077]);
            r[cur] = v;
            *o = v;
        }
        cur += CHUNK;
        o += CHUNK;
        left -= CHUNK;
        ++in_span;
        // LDS-only barrier: __syncthreads() would also drain the global store above (vmcnt(0)) on every
        // step; the words are only read by later kernels
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// numpy legacy_double: (a >> 5, b >> 6) -> [0, 1)
__device__ __forceinline__ double legacy_double(unsigned a, unsigned b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
__device__ __forceinline__ bool polar_attempt(const unsigned* __restrict__ w, long i, double& x1, double& x2, double& r2) {
    const uint4 q = *reinterpret_cast<const uint4*>(w + 4 * i);
    x1 = 2.0 * legacy_double(temper(q.x), temper(q.y)) - 1.0;
    x2 = 2.0 * legacy_double(temper(q.z), temper(q.w)) - 1.0;
    r2 = x1 * x1 + x2 * x2;
    return !(r2 >= 1.0 || r2 == 0.0);
}

__global__ void polar_flags_kernel(const unsigned* __restrict__ w, long n_attempts, int* __restrict__ block_count) {
    __shared__ int cnt[FBLK / 64];
    const long i = (long)blockIdx.x * FBLK + threadIdx.x;
    double x1, x2, r2;
    const bool acc = i < n_attempts && polar_attempt(w, i, x1, x2, r2);
    const unsigned long long m = __ballot(acc);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// exclusive scan of block_count (single workgroup), total -> offsets[nb]
__global__ void block_offsets_kernel(const int* __restrict__ block_count, int nb, long* __restrict__ offsets) {
    __shared__ long part[1024];
    const int per = (nb + 1023) / 1024;
    long s = 0;
    for (int k = 0; k < per; ++k) {
        const int b = threadIdx.x * per + k;
        if (b < nb) s += block_count[b];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        long v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    long run = part[threadIdx.x] - s;           // exclusive prefix of my segment
    for (int k = 0; k < per; ++k) {
        const int b = threadIdx.x * per + k;
        if (b < nb) {
            offsets[b] = run;
            run += block_count[b];
        }
    }
    if (threadIdx.x == 1023) offsets[nb] = part[1023];
}

template <typename T>
__global__ void polar_emit_kernel(const unsigned* __restrict__ w, long n_attempts, const long* __restrict__ offsets,
                                  long first, long n_normals, double scale, T* __restrict__ noise,
                                  int* __restrict__ status) {
    __shared__ int cnt[FBLK / 64];
    const long i = (long)blockIdx.x * FBLK + threadIdx.x;
    double x1 = 0, x2 = 0, r2 = 1;
    const bool acc = i < n_attempts && polar_attempt(w, i, x1, x2, r2);
    const unsigned long long m = __ballot(acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) cnt[wv] = __popcll(m);
    __syncthreads();
    long k = offsets[blockIdx.x];
    for (int q = 0; q < wv; ++q) k += cnt[q];
    k += __popcll(m & ((1ull << lane) - 1ull));
    // normals [first, first + n_normals) of the stream land in noise[0 .. n_normals): a sharded run keeps its own
    // block of the one global stream
    const long last = first + n_normals;
    if (acc && 2 * k + 1 >= first && 2 * k < last) {
        const double f = sqrt(-2.0 * log(r2) / r2);                 // legacy_gauss
        if (2 * k >= first) noise[2 * k - first] = (T)((f * x2) * scale);              // returned first
        if (2 * k + 1 < last) noise[2 * k + 1 - first] = (T)((f * x1) * scale);        // the cached one
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const long need = (last + 1) / 2;
        if (status && offsets[gridDim.x] < need) *status = 1;     // sticky (the host clears what it reports): not enough attempts generated
    }
}

}  // namespace

long mt_attempts_for(long n_normals) {
    const long pairs = (n_normals + 1) / 2;
    return (long)((double)pairs / 0.7853981633974483 * 1.01) + 4096;          // acceptance pi/4, +1 % (> 14 sigma)
}

long mt_workspace_bytes(long n_normals) {
    const long na = mt_attempts_for(n_normals), nb = (na + FBLK - 1) / FBLK;
    return 4 * MT_N + 16 * na + 4 * nb + 8 * (nb + 1) + 64 + 4L * MT_MAX_SEGMENTS * JSLICES * MT_N;
}

template <typename T>
hipError_t sample_noise_mt19937(T* noise, long n_normals, double scale, unsigned long long seed, const long long* d_step,
                                void* ws, int* status, hipStream_t s, const int* jump_idx, const int* jump_starts,
                                long head_words, long seg_words, int n_segments, long first_normal) {
    if (n_normals <= 0) return hipSuccess;
    const long na = mt_attempts_for(first_normal + n_normals), nb = (na + FBLK - 1) / FBLK;
    unsigned* all = (unsigned*)ws;                                  // 624 + 4 * na words
    unsigned* words = all + MT_N;
    int* counts = (int*)(words + 4 * na);
    long* offsets = (long*)(((uintptr_t)(counts + nb) + 7) & ~(uintptr_t)7);
    unsigned* jstates = (unsigned*)(offsets + nb + 1);              // [n_segments][JSLICES][624]
    const long total = 4 * na;
    if (jump_idx && n_segments > 0) {
        // head (serial, short) then n_segments workgroups in parallel, each from its jumped-ahead state
        if (n_segments > MT_MAX_SEGMENTS || MT_N + head_words > JHEAD_MAX || (head_words & 3)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(mt_segment_kernel, dim3(1), dim3(640), 0, s, seed, d_step, all, head_words, 0L, total,
                           (const unsigned*)nullptr);
        hipLaunchKernelGGL(mt_jump_kernel, dim3(n_segments * JSLICES), dim3(640), 0, s, all, (int)(MT_N + head_words),
                           jump_idx, jump_starts, jstates);
        hipLaunchKernelGGL(mt_segment_kernel, dim3(n_segments), dim3(640), 0, s, seed, d_step, all, head_words,
                           seg_words, total, (const unsigned*)jstates);
    } else {
        hipLaunchKernelGGL(mt_segment_kernel, dim3(1), dim3(640), 0, s, seed, d_step, all, total, 0L, total,
                           (const unsigned*)nullptr);
    }
    hipLaunchKernelGGL(polar_flags_kernel, dim3((unsigned)nb), dim3(FBLK), 0, s, words, na, counts);
    hipLaunchKernelGGL(block_offsets_kernel, dim3(1), dim3(1024), 0, s, counts, (int)nb, offsets);
    hipLaunchKernelGGL(polar_emit_kernel<T>, dim3((unsigned)nb), dim3(FBLK), 0, s, words, na, offsets, first_normal,
                       n_normals, scale, noise, status);
    return hipGetLastError();
}

template hipError_t sample_noise_mt19937<float>(float*, long, double, unsigned long long, const long long*, void*, int*,
                                                hipStream_t, const int*, const int*, long, long, int, long);
template hipError_t sample_noise_mt19937<double>(double*, long, double, unsigned long long, const long long*, void*, int*,
                                                 hipStream_t, const int*, const int*, long, long, int, long);

}  // namespace mjmpc

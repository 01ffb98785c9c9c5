// extern "C" entry points declared in include/mjmpc_amd.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mjmpc_amd.h"
#include "arm_model.h"
#include "analytic_rollout.h"
#include "arm_rollout.h"
#include "noise_mt.h"
#include "tree_model.h"
#include "tree_rollout.h"
#include "update.h"

// failure counter (one unsigned) followed by the phase-clock slots of developer builds (arm_rollout.hip, Stamps)
constexpr size_t MJMPC_DIAG_BYTES = 8 * (2 + 64);      // counters, then the developer clocks of -DMJMPC_STAMPS builds (4 waves x 16)

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace
namespace mjmpc {
// (comm.hip reports through the same thread-local message as the entry points of this file)
int set_error(int code, const char* what, const char* detail) { return fail(code, "%s: %s", what, detail); }
}  // namespace mjmpc
namespace {

int hip_fail(hipError_t e, const char* what) {
    return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(expr)                                \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

}  // namespace

struct mjmpc_arm_s {
    int device = 0;
    int nv = 0, nu = 0, d_obs = 0;
    float* model_f32 = nullptr;
    double* model_f64 = nullptr;
    double* state = nullptr;        // MJMPC_ARM_STATE_LEN
    unsigned* diag = nullptr;
    double* pinned = nullptr;       // host staging for set_state: a ring of STAGE_SLOTS vectors, one event each
    hipEvent_t staged[4] = {nullptr, nullptr, nullptr, nullptr};
    int stage_next = 0;
    int n_shards = 1;               // > 1: model_f32 / model_f64 hold one block per shard
    double* zero_action = nullptr;  // [32] zeros (the kinematics-only launch of mjmpc_tree_rollout_cl)
    double* scratch = nullptr;      // [8] a place for that launch's cost
    double* shard_states = nullptr; // n_state_shards state vectors (per-shard start states)
    int n_state_shards = 0;
    // mjmpc_arm_mppi_step: the rollout workgroups' records.  The pointer travels to the kernels BY VALUE (MonoStep), so a
    // captured graph holds it: the buffer only ever GROWS, and a buffer it outgrew stays allocated until the handle is
    // destroyed (mono_retired) - a graph captured at one (P, H) survives later calls at another
    double* reset_rec = nullptr;    // n_shards records of ARM_RESET_LEN: MuJoCo's reset on instability (RolloutFusion::reset_rec)
    int inf_on_reset = 0;           // mjmpc_arm_set_reset_returns
    bool xj = false;                // slide joints / friction loss: launches go to the extended-joint build (arm_blob_is_xj)
    std::vector<double*> reset_retired;     // reset records that were replaced: bound launchers / captured graphs carry the
                                            // pointer by value (RolloutFusion, MonoStep), so they stay allocated until destroy
    double* mono_tree = nullptr;
    size_t mono_cap = 0;            // doubles
    std::vector<double*> mono_retired;
};

// which build of the arm kernels an engine's launches go to: the extended-joint one (arm_rollout_xj.hip) when some block of its
// model has a slide joint or a dof with friction loss
static bool arm_blob_is_xj(const double* blobs, int n_shards) {
    for (int s = 0; s < n_shards; ++s)
        for (int l = 0; l < mjmpc::LANES; ++l) {
            const double* b = blobs + (size_t)s * mjmpc::ARM_BLOB_LEN;
            if (b[mjmpc::O_JTYPE + l] != 0.0 || b[mjmpc::O_FLOSS + l] > 0.0) return true;
        }
    return false;
}
template <typename T, typename... A>
static hipError_t arm_rollout_launch(const mjmpc_arm_s* h, A&&... a) {
    return h->xj ? mjmpc::launch_arm_rollout_xj<T>(a...) : mjmpc::launch_arm_rollout<T>(a...);
}
template <typename T, typename... A>
static hipError_t arm_finish_launch(const mjmpc_arm_s* h, A&&... a) {
    return h->xj ? mjmpc::launch_arm_mppi_finish_xj<T>(a...) : mjmpc::launch_arm_mppi_finish<T>(a...);
}

struct mjmpc_tree_s {
    int device = 0;
    int nv = 0, nu = 0, d_obs = 0, max_path = 0, nq = 0;
    bool full = false;              // slide joints, springs, friction cones, > 8 contact points or a medium: the full kernel
    int gen = 0;                    // T_GEN: 1 ball / free joints, friction loss, boxes, equalities, tendon limits (the general
                                    // instantiation), 2 round 5's record kinds on top (instantiations of their own)
    int n_shards = 1;               // > 1: model_f32 / model_f64 hold one block per shard
    double* zero_action = nullptr;  // [32] zeros (the kinematics-only launch of mjmpc_tree_rollout_cl)
    double* scratch = nullptr;      // [8] a place for that launch's cost
    float* model_f32 = nullptr;
    double* model_f64 = nullptr;
    double* state = nullptr;        // MJMPC_TREE_DEVICE_STATE_LEN
    unsigned* diag = nullptr;
    double* shard_states = nullptr; // n_state_shards state vectors (per-shard start states)
    int n_state_shards = 0;
    double* pinned = nullptr;       // host staging for set_state: a ring of 4 vectors, one event each
    hipEvent_t staged[4] = {nullptr, nullptr, nullptr, nullptr};
    int stage_next = 0;
    std::vector<double> topo;       // create-time topology tables (shard blocks must match them)
    double* reset_rec = nullptr;    // n_shards records of TREE_RESET_LEN: MuJoCo's reset on instability (TreeFusion::reset_rec)
    int inf_on_reset = 0;           // mjmpc_tree_set_reset_returns
    std::vector<double*> reset_retired;     // (as mjmpc_arm_s::reset_retired)
};

extern "C" {

int mjmpc_abi_version(void) { return MJMPC_ABI_VERSION; }

const char* mjmpc_last_error(void) { return g_err; }

int mjmpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mjmpc_graph_kernel_nodes(void* hip_graph, int64_t* n_out) {
    if (!hip_graph || !n_out) return fail(MJMPC_E_BADARG, "null argument");
    size_t n = 0;
    HIP_TRY(hipGraphGetNodes((hipGraph_t)hip_graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) HIP_TRY(hipGraphGetNodes((hipGraph_t)hip_graph, nodes.data(), &n));
    int64_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        HIP_TRY(hipGraphNodeGetType(nodes[i], &t));
        k += t == hipGraphNodeTypeKernel;
    }
    n_out[0] = k;
    n_out[1] = (int64_t)n;
    return 0;
}

// An order-independent signature of a captured graph: n_out[0] = sum over its KERNEL nodes of a hash of (function, grid,
// block, dynamic LDS), n_out[1] = the same over every node's type.  (Argument VALUES are out of the runtime's reach - a
// kernel node's parameter array comes without sizes - but they are the recorded ones by construction; what can differ
// between an iteration and its tape is which kernels run and in what shape.)
int mjmpc_graph_signature(void* hip_graph, uint64_t* n_out) {
    if (!hip_graph || !n_out) return fail(MJMPC_E_BADARG, "null argument");
    size_t n = 0;
    HIP_TRY(hipGraphGetNodes((hipGraph_t)hip_graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) HIP_TRY(hipGraphGetNodes((hipGraph_t)hip_graph, nodes.data(), &n));
    auto mix = [](uint64_t h, uint64_t v) { return (h ^ v) * 1099511628211ull; };
    uint64_t ksum = 0, tsum = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        HIP_TRY(hipGraphNodeGetType(nodes[i], &t));
        tsum += mix(1469598103934665603ull, (uint64_t)t);
        if (t != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        HIP_TRY(hipGraphKernelNodeGetParams(nodes[i], &kp));
        uint64_t h = 1469598103934665603ull;
        h = mix(h, (uint64_t)(uintptr_t)kp.func);
        h = mix(h, ((uint64_t)kp.gridDim.x << 32) | kp.gridDim.y);
        h = mix(h, ((uint64_t)kp.gridDim.z << 32) | kp.blockDim.x);
        h = mix(h, ((uint64_t)kp.blockDim.y << 32) | kp.blockDim.z);
        h = mix(h, (uint64_t)kp.sharedMemBytes);
        ksum += h;
    }
    n_out[0] = ksum;
    n_out[1] = tsum;
    return 0;
}

static mjmpc::RolloutFusion arm_fuse(const mjmpc_arm_s* h) {
    mjmpc::RolloutFusion f;
    f.reset_rec = h->reset_rec;
    f.inf_on_reset = h->inf_on_reset;
    return f;
}

// The reset records of `n_shards` model blocks (host, ARM_BLOB_LEN each): per block ONE substep of the f64 kernel itself
// from the reset state (qpos0 = 0, zero velocity, zero controls) on a copy of the block with frame_skip 1 - state_out
// receives the state after it, the next observation's site entries are the site at the reset state.  Synchronous; called
// when the engine is created and when its model blocks are replaced.
// The records are returned in *out and the handle is NOT touched: the caller commits them together with the model blocks
// (mjmpc_arm_set_shard_models), so that a failure here leaves the engine as it was; the launches count into a scratch
// counter block of their own, not into the engine's live diagnostics.
static int arm_make_reset_records(mjmpc_arm_s* h, const double* blobs, int n_shards, double** out) {
    const size_t L = (size_t)mjmpc::ARM_BLOB_LEN, R = (size_t)mjmpc::ARM_RESET_LEN;
    const int dobs = 2 * h->nv + 6;
    double *rec = nullptr, *tmp = nullptr;
    HIP_TRY(hipMalloc(&rec, sizeof(double) * R * n_shards));
    // scratch: model block | state (19) | mean (8) | cost (1) | next observation (dobs) | counters (MJMPC_DIAG_BYTES)
    const size_t nscr = L + MJMPC_ARM_STATE_LEN + 8 + 1 + dobs + MJMPC_DIAG_BYTES / sizeof(double) + 1;
    hipError_t e = hipMalloc(&tmp, sizeof(double) * nscr);
    if (e == hipSuccess) e = hipMemset(tmp, 0, sizeof(double) * nscr);
    if (e == hipSuccess) e = hipMemset(rec, 0, sizeof(double) * R * n_shards);
    std::vector<double> b(L);
    double *st = tmp + L, *mean = st + MJMPC_ARM_STATE_LEN, *cost = mean + 8, *nobs = cost + 1;
    unsigned* sdiag = (unsigned*)(nobs + dobs);
    for (int k = 0; k < n_shards && e == hipSuccess; ++k) {
        std::memcpy(b.data(), blobs + (size_t)k * L, sizeof(double) * L);
        b[mjmpc::O_FRAME_SKIP] = 1.0;
        e = hipMemcpy(tmp, b.data(), sizeof(double) * L, hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
        double* rk = rec + (size_t)k * R;
        e = arm_rollout_launch<double>(h, tmp, st, 1, 1, h->nu, mean, nullptr, cost, nullptr, nullptr, nobs, rk, sdiag, nullptr);
        if (e == hipSuccess) e = hipMemcpy(rk + 2 * mjmpc::LANES, nobs + 2 * h->nv, sizeof(double) * 3, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) break;
        double host[mjmpc::ARM_RESET_LEN];          // sin / cos of the record's qpos (the kernels carry them beside q)
        e = hipMemcpy(host, rk, sizeof(double) * 19, hipMemcpyDeviceToHost);
        for (int l = 0; l < mjmpc::LANES; ++l) {
            const bool slide = b[mjmpc::O_JTYPE + l] != 0.0;     // (a slide joint's coordinate is a length: the kernels keep (0, 1))
            host[19 + l] = slide ? 0.0 : std::sin(host[l]);
            host[19 + mjmpc::LANES + l] = slide ? 1.0 : std::cos(host[l]);
        }
        if (e == hipSuccess) e = hipMemcpy(rk + 19, host + 19, sizeof(double) * 16, hipMemcpyHostToDevice);
    }
    hipFree(tmp);
    if (e != hipSuccess) {
        hipFree(rec);
        return hip_fail(e, "reset record");
    }
    *out = rec;
    return 0;
}

static int arm_create_impl(mjmpc_arm_s* h, const double* blob, int n_blob) {
    std::vector<float> f32(blob, blob + n_blob);
    h->xj = arm_blob_is_xj(blob, 1);
    HIP_TRY(hipMalloc(&h->model_f32, sizeof(float) * n_blob));
    HIP_TRY(hipMalloc(&h->model_f64, sizeof(double) * n_blob));
    HIP_TRY(hipMalloc(&h->state, sizeof(double) * MJMPC_ARM_STATE_LEN));
    HIP_TRY(hipMalloc(&h->diag, MJMPC_DIAG_BYTES));
    HIP_TRY(hipHostMalloc(&h->pinned, sizeof(double) * MJMPC_ARM_STATE_LEN * 4));
    for (int k = 0; k < 4; ++k) HIP_TRY(hipEventCreateWithFlags(&h->staged[k], hipEventDisableTiming));
    HIP_TRY(hipMemcpy(h->model_f32, f32.data(), sizeof(float) * n_blob, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->model_f64, blob, sizeof(double) * n_blob, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->state, 0, sizeof(double) * MJMPC_ARM_STATE_LEN));
    HIP_TRY(hipMemset(h->diag, 0, MJMPC_DIAG_BYTES));
    return arm_make_reset_records(h, blob, 1, &h->reset_rec);
}

int mjmpc_arm_create(const double* blob, int n_blob, int device, mjmpc_arm_t* out) {
    if (!blob || !out) return fail(MJMPC_E_BADARG, "null argument");
    if (n_blob != mjmpc::ARM_BLOB_LEN) return fail(MJMPC_E_BADMODEL, "model blob has %d scalars, expected %d", n_blob, (int)mjmpc::ARM_BLOB_LEN);
    const int nv = (int)blob[mjmpc::O_NV];
    if (nv < 1 || nv > mjmpc::MAX_LINKS) return fail(MJMPC_E_BADMODEL, "nv = %d outside 1..%d", nv, mjmpc::MAX_LINKS);
    if (mjmpc_device_count() <= device) return fail(MJMPC_E_NOGPU, "HIP device %d not present", device);
    HIP_TRY(hipSetDevice(device));
    mjmpc_arm_s* h = new mjmpc_arm_s();
    h->device = device;
    h->nv = nv;
    h->nu = (int)blob[mjmpc::O_NU] >= 1 && (int)blob[mjmpc::O_NU] <= nv ? (int)blob[mjmpc::O_NU] : nv;
    h->d_obs = 2 * nv + 6;
    if (int rc = arm_create_impl(h, blob, n_blob)) {        // a failed allocation leaves nothing behind
        mjmpc_arm_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int mjmpc_arm_set_shard_models(mjmpc_arm_t h, const double* blobs, int n_shards) {
    if (!h || !blobs || n_shards < 1) return fail(MJMPC_E_BADARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)n_shards * mjmpc::ARM_BLOB_LEN;
    for (int s = 0; s < n_shards; ++s)
        if ((int)blobs[(size_t)s * mjmpc::ARM_BLOB_LEN + mjmpc::O_NV] != h->nv)
            return fail(MJMPC_E_BADMODEL, "shard %d has a different nv", s);
    std::vector<float> f32(blobs, blobs + n);
    HIP_TRY(hipDeviceSynchronize());
    float* m32 = nullptr;
    double* m64 = nullptr;
    HIP_TRY(hipMalloc(&m32, sizeof(float) * n));
    if (hipError_t e = hipMalloc(&m64, sizeof(double) * n); e != hipSuccess) {
        hipFree(m32);
        return hip_fail(e, "hipMalloc");
    }
    hipError_t e = hipMemcpy(m32, f32.data(), sizeof(float) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m64, blobs, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(m32);
        hipFree(m64);
        return hip_fail(e, "hipMemcpy");
    }
    // the reset records of the NEW blocks first: if that fails the engine keeps its old models, shard count and records
    double* rec = nullptr;
    const bool xj_old = h->xj;
    h->xj = arm_blob_is_xj(blobs, n_shards);        // (the record launches run the build the NEW blocks need)
    if (int rc = arm_make_reset_records(h, blobs, n_shards, &rec); rc != 0) {
        h->xj = xj_old;
        hipFree(m32);
        hipFree(m64);
        return rc;
    }
    hipFree(h->model_f32);
    hipFree(h->model_f64);
    h->model_f32 = m32;
    h->model_f64 = m64;
    h->n_shards = n_shards;
    if (h->reset_rec) h->reset_retired.push_back(h->reset_rec);     // (launchers bound earlier still point at it)
    h->reset_rec = rec;
    return 0;
}

int mjmpc_arm_set_shard_states(mjmpc_arm_t h, const double* states, int n_shards, void* stream) {
    if (!h || (n_shards > 0 && !states) || n_shards < 0) return fail(MJMPC_E_BADARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (n_shards != h->n_state_shards) {
        HIP_TRY(hipDeviceSynchronize());
        hipFree(h->shard_states);
        h->shard_states = nullptr;
        if (n_shards > 0) HIP_TRY(hipMalloc(&h->shard_states, sizeof(double) * MJMPC_ARM_STATE_LEN * n_shards));
        h->n_state_shards = n_shards;
    }
    if (n_shards > 0) {
        HIP_TRY(hipMemcpyAsync(h->shard_states, states, sizeof(double) * MJMPC_ARM_STATE_LEN * n_shards,
                               hipMemcpyHostToDevice, (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));      // `states` is pageable host memory
    }
    return 0;
}

static int shard_fusion(mjmpc_arm_t h, int64_t P, mjmpc::RolloutFusion& fuse) {
    if (h->n_state_shards > 1) {
        if (P % h->n_state_shards != 0 || (P / h->n_state_shards) % mjmpc::LANES != 0)
            return fail(MJMPC_E_BADARG, "with per-shard start states P / n_shards must be a multiple of 8");
        fuse.state_shard_size = (long)(P / h->n_state_shards);
    }
    if (h->n_shards <= 1) return 0;
    if (P % h->n_shards != 0) return fail(MJMPC_E_BADARG, "P = %lld is not divisible by %d shards", (long long)P, h->n_shards);
    const long ss = (long)(P / h->n_shards);
    if (ss % mjmpc::LANES != 0) return fail(MJMPC_E_BADARG, "with per-shard models a shard must hold a multiple of 8 particles (got %ld)", ss);
    fuse.shard_size = ss;
    return 0;
}

int mjmpc_arm_destroy(mjmpc_arm_t h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipFree(h->model_f32);
    hipFree(h->model_f64);
    hipFree(h->state);
    hipFree(h->diag);
    hipFree(h->shard_states);
    hipFree(h->reset_rec);
    for (double* p : h->reset_retired) hipFree(p);
    hipFree(h->mono_tree);
    for (double* p : h->mono_retired) hipFree(p);
    hipHostFree(h->pinned);
    for (int k = 0; k < 4; ++k) if (h->staged[k]) hipEventDestroy(h->staged[k]);
    delete h;
    return 0;
}

int mjmpc_arm_dims(mjmpc_arm_t h, int* nv, int* nu, int* d_obs) {
    if (!h) return fail(MJMPC_E_BADARG, "null engine");
    if (nv) *nv = h->nv;
    if (nu) *nu = h->nu;
    if (d_obs) *d_obs = h->d_obs;
    return 0;
}

int mjmpc_arm_set_state(mjmpc_arm_t h, const double* qpos, const double* qvel, const double* target_pos,
                        void* stream) {
    if (!h || !qpos || !qvel || !target_pos) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(h->device));
    // Staging ring: wait only for the copy that last used THIS slot (four calls ago), never for the stream - a
    // captured control iteration still running on `s` keeps running while the next state is being staged.
    const int slot = h->stage_next;
    h->stage_next = (slot + 1) & 3;
    HIP_TRY(hipEventSynchronize(h->staged[slot]));
    double* stage = h->pinned + (size_t)slot * MJMPC_ARM_STATE_LEN;
    std::memset(stage, 0, sizeof(double) * MJMPC_ARM_STATE_LEN);
    std::memcpy(stage, qpos, sizeof(double) * h->nv);
    std::memcpy(stage + mjmpc::LANES, qvel, sizeof(double) * h->nv);
    std::memcpy(stage + 2 * mjmpc::LANES, target_pos, sizeof(double) * 3);
    HIP_TRY(hipMemcpyAsync(h->state, stage, sizeof(double) * MJMPC_ARM_STATE_LEN, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(h->staged[slot], s));
    return 0;
}

double* mjmpc_arm_state_ptr(mjmpc_arm_t h) { return h ? h->state : nullptr; }

int mjmpc_arm_rollout(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                      void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream) {
    if (!h || !d_mean || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    if (P < 0 || H < 0) return fail(MJMPC_E_BADARG, "negative size");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    mjmpc::RolloutFusion fuse = arm_fuse(h);
    if (int rc = shard_fusion(h, P, fuse)) return rc;
    const double* st = fuse.state_shard_size > 0 ? h->shard_states : h->state;
    if (dtype == MJMPC_F32) {
        e = arm_rollout_launch<float>(h, h->model_f32, st, (long)P, H, h->nu, d_mean, (const float*)d_noise,
                                             (float*)d_costs, (float*)d_actions, (float*)d_obs, (float*)d_next_obs,
                                             nullptr, h->diag, s, fuse);
    } else if (dtype == MJMPC_F64) {
        e = arm_rollout_launch<double>(h, h->model_f64, st, (long)P, H, h->nu, d_mean,
                                              (const double*)d_noise, (double*)d_costs, (double*)d_actions,
                                              (double*)d_obs, (double*)d_next_obs, nullptr, h->diag, s, fuse);
    } else {
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    }
    if (e != hipSuccess) return hip_fail(e, "arm_rollout launch");
    return 0;
}

int mjmpc_arm_rollout_cl(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_weights, const void* d_noise,
                         void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream) {
    if (!h || !d_weights || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    if (P < 0 || H < 0) return fail(MJMPC_E_BADARG, "negative size");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    mjmpc::RolloutFusion fuse = arm_fuse(h);
    if (int rc = shard_fusion(h, P, fuse)) return rc;
    const double* st = fuse.state_shard_size > 0 ? h->shard_states : h->state;
    fuse.clw = d_weights;
    hipError_t e;
    if (dtype == MJMPC_F32)
        e = arm_rollout_launch<float>(h, h->model_f32, st, (long)P, H, h->nu, d_weights, (const float*)d_noise,
                                             (float*)d_costs, (float*)d_actions, (float*)d_obs, (float*)d_next_obs,
                                             nullptr, h->diag, s, fuse);
    else if (dtype == MJMPC_F64)
        e = arm_rollout_launch<double>(h, h->model_f64, st, (long)P, H, h->nu, d_weights,
                                              (const double*)d_noise, (double*)d_costs, (double*)d_actions,
                                              (double*)d_obs, (double*)d_next_obs, nullptr, h->diag, s, fuse);
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "arm_rollout_cl launch");
    return 0;
}

int mjmpc_arm_rollout_fused(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                            const double* d_filter_coeffs, const double* d_gseq, void* d_costs, void* d_actions,
                            double* d_q0, void* stream) {
    if (!h || !d_mean || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    if ((d_q0 != nullptr) != (d_gseq != nullptr)) return fail(MJMPC_E_BADARG, "d_q0 and d_gseq go together");
    if (P < 0 || H < 0) return fail(MJMPC_E_BADARG, "negative size");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    mjmpc::RolloutFusion fuse = arm_fuse(h);
    if (int rc = shard_fusion(h, P, fuse)) return rc;
    const double* st = fuse.state_shard_size > 0 ? h->shard_states : h->state;
    fuse.filt = d_filter_coeffs;
    fuse.gseq = d_gseq;
    fuse.q0_out = d_q0;
    hipError_t e;
    if (dtype == MJMPC_F32)
        e = arm_rollout_launch<float>(h, h->model_f32, st, (long)P, H, h->nu, d_mean, (const float*)d_noise,
                                             (float*)d_costs, (float*)d_actions, nullptr, nullptr, nullptr, h->diag, s,
                                             fuse);
    else if (dtype == MJMPC_F64)
        e = arm_rollout_launch<double>(h, h->model_f64, st, (long)P, H, h->nu, d_mean,
                                              (const double*)d_noise, (double*)d_costs, (double*)d_actions, nullptr,
                                              nullptr, nullptr, h->diag, s, fuse);
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "arm_rollout_fused launch");
    return 0;
}

int mjmpc_arm_mppi_step(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, double* d_mean_out,
                        const double* d_gseq, const double* d_filter_coeffs, const double* d_chol, uint64_t seed,
                        uint64_t offset, int64_t particle_offset, int64_t* d_step_counter, double lam, double step_size,
                        int shift_mode, double* d_action_out, double* h_action_slots, double* d_record, int env_step,
                        void* d_step_cost, void* d_step_next_obs, void* d_costs, void* d_actions, double* d_q0, void* stream) {
    if (!h || !d_mean || !d_gseq || !d_chol) return fail(MJMPC_E_BADARG, "null argument");
    if (!d_record && shift_mode != -2 && (!d_mean_out || d_mean_out == d_mean))
        return fail(MJMPC_E_BADARG, "d_mean_out must be a buffer of its own (the finish launch reads d_mean while it writes)");
    if (P < 1 || H < 1) return fail(MJMPC_E_BADARG, "P and H must be positive");
    if (!(lam > 0) || shift_mode > 1) return fail(MJMPC_E_BADARG, "bad lam / shift_mode");
    const bool rollout_only = shift_mode == -2;         // (measurement: the first launch alone, records left in the engine)
    if (h->n_shards > 1 || h->n_state_shards > 1)
        return fail(MJMPC_E_BADARG, "the fused iteration runs one model and one start state (no per-shard blocks)");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const long groups = mjmpc::arm_rollout_groups((long)P);
    const size_t need = (size_t)mjmpc::mono_record_doubles(groups, H, h->nu);
    if (need > h->mono_cap) {
        // grow only, never under a capture (an allocation would invalidate it), and never free what an earlier graph
        // may still point at
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(MJMPC_E_BADARG, "the record buffer must grow for this (P, H): call once outside stream capture first");
        double* bigger = nullptr;
        HIP_TRY(hipMalloc(&bigger, sizeof(double) * need));
        if (h->mono_tree) h->mono_retired.push_back(h->mono_tree);
        h->mono_tree = bigger;
        h->mono_cap = need;
    }
    mjmpc::MonoStep mo;             // (travels to both kernels by value, as a kernel argument)
    mo.chol = d_chol;
    mo.seed = seed;
    mo.offset = offset;
    mo.particle_offset = (long)particle_offset;
    mo.d_step = (const long long*)d_step_counter;
    mo.lam = lam;
    mo.step_size = step_size;
    mo.shift_mode = shift_mode;
    mo.tree = h->mono_tree;
    mo.action_out = d_action_out;
    mo.action_host = h_action_slots;
    mo.step_counter = (long long*)d_step_counter;
    mo.record = d_record;
    mo.state_io = h->state;
    mo.step_cost = d_step_cost;
    mo.step_nobs = d_step_next_obs;
    mo.reset_rec = h->reset_rec;
    const int do_env = (env_step && !d_record) ? 1 : 0;
    mjmpc::RolloutFusion fuse = arm_fuse(h);
    fuse.filt = d_filter_coeffs;
    fuse.gseq = d_gseq;
    fuse.q0_out = d_q0;
    hipError_t e;
    if (dtype == MJMPC_F32) {
        e = arm_rollout_launch<float>(h, h->model_f32, h->state, (long)P, H, h->nu, d_mean, nullptr, (float*)d_costs,
                                             (float*)d_actions, nullptr, nullptr, nullptr, h->diag, s, fuse, &mo);
        if (e == hipSuccess && !rollout_only)
            e = arm_finish_launch<float>(h, h->model_f32, h->mono_tree, groups, H, h->nu, d_mean, d_mean_out, mo,
                                                     do_env, h->diag, s);
    } else if (dtype == MJMPC_F64) {
        e = arm_rollout_launch<double>(h, h->model_f64, h->state, (long)P, H, h->nu, d_mean, nullptr, (double*)d_costs,
                                              (double*)d_actions, nullptr, nullptr, nullptr, h->diag, s, fuse, &mo);
        if (e == hipSuccess && !rollout_only)
            e = arm_finish_launch<double>(h, h->model_f64, h->mono_tree, groups, H, h->nu, d_mean, d_mean_out, mo,
                                                      do_env, h->diag, s);
    } else {
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    }
    if (e != hipSuccess) return hip_fail(e, "arm_mppi_step launch");
    return 0;
}

int mjmpc_arm_rollout_sampled(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, const double* d_gseq,
                              const double* d_filter_coeffs, const double* d_chol, int chol_full, uint64_t seed, uint64_t offset,
                              int64_t particle_offset, const int64_t* d_step_counter, void* d_costs, void* d_actions,
                              double* d_q0, void* stream) {
    if (!h || !d_mean || !d_gseq || !d_chol) return fail(MJMPC_E_BADARG, "null argument");
    if (P < 1 || H < 1) return fail(MJMPC_E_BADARG, "P and H must be positive");
    if (h->n_shards > 1 || h->n_state_shards > 1)
        return fail(MJMPC_E_BADARG, "sampled rollouts run one model and one start state (no per-shard blocks)");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const long groups = mjmpc::arm_rollout_groups((long)P);
    const size_t need = (size_t)mjmpc::mono_record_doubles(groups, H, h->nu);      // (the launch also leaves its softmax partials)
    if (need > h->mono_cap) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(MJMPC_E_BADARG, "the record buffer must grow for this (P, H): call once outside stream capture first");
        double* bigger = nullptr;
        HIP_TRY(hipMalloc(&bigger, sizeof(double) * need));
        if (h->mono_tree) h->mono_retired.push_back(h->mono_tree);
        h->mono_tree = bigger;
        h->mono_cap = need;
    }
    mjmpc::MonoStep mo;
    mo.chol = d_chol;
    mo.chol_full = chol_full ? 1 : 0;
    mo.seed = seed;
    mo.offset = offset;
    mo.particle_offset = (long)particle_offset;
    mo.d_step = (const long long*)d_step_counter;
    mo.lam = 1.0;
    mo.shift_mode = -2;
    mo.tree = h->mono_tree;
    mjmpc::RolloutFusion fuse = arm_fuse(h);
    fuse.filt = d_filter_coeffs;
    fuse.gseq = d_gseq;
    fuse.q0_out = d_q0;
    hipError_t e;
    if (dtype == MJMPC_F32)
        e = arm_rollout_launch<float>(h, h->model_f32, h->state, (long)P, H, h->nu, d_mean, nullptr, (float*)d_costs,
                                             (float*)d_actions, nullptr, nullptr, nullptr, h->diag, s, fuse, &mo);
    else if (dtype == MJMPC_F64)
        e = arm_rollout_launch<double>(h, h->model_f64, h->state, (long)P, H, h->nu, d_mean, nullptr, (double*)d_costs,
                                              (double*)d_actions, nullptr, nullptr, nullptr, h->diag, s, fuse, &mo);
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "arm_rollout_sampled launch");
    return 0;
}

int mjmpc_arm_mppi_combine(mjmpc_arm_t h, int dtype, const double* d_records, int n_records, int H, const double* d_mean,
                           double* d_mean_out, int64_t* d_step_counter, double step_size, int shift_mode,
                           double* d_action_out, double* h_action_slots, int env_step, void* d_step_cost,
                           void* d_step_next_obs, void* stream) {
    if (!h || !d_records || !d_mean || !d_mean_out || d_mean_out == d_mean) return fail(MJMPC_E_BADARG, "null / aliased argument");
    if (n_records < 1 || H < 1 || shift_mode > 1 || shift_mode < -1) return fail(MJMPC_E_BADARG, "bad n_records / H / shift_mode");
    if (env_step && (h->n_shards > 1 || h->n_state_shards > 1))
        return fail(MJMPC_E_BADARG, "the fused env step runs one model and one state (no per-shard blocks)");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    mjmpc::MonoStep mo;
    mo.step_size = step_size;
    mo.shift_mode = shift_mode;
    mo.action_out = d_action_out;
    mo.action_host = h_action_slots;
    mo.step_counter = (long long*)d_step_counter;
    mo.state_io = h->state;
    mo.step_cost = d_step_cost;
    mo.step_nobs = d_step_next_obs;
    mo.reset_rec = h->reset_rec;
    hipError_t e;
    if (dtype == MJMPC_F32)
        e = arm_finish_launch<float>(h, h->model_f32, d_records, n_records, H, h->nu, d_mean, d_mean_out, mo,
                                                 env_step ? 1 : 0, h->diag, s);
    else if (dtype == MJMPC_F64)
        e = arm_finish_launch<double>(h, h->model_f64, d_records, n_records, H, h->nu, d_mean, d_mean_out, mo,
                                                  env_step ? 1 : 0, h->diag, s);
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "arm_mppi_combine launch");
    return 0;
}

int mjmpc_arm_step_state(mjmpc_arm_t h, int dtype, const double* d_action, void* d_cost, void* d_next_obs,
                         void* stream) {
    if (!h || !d_action || !d_cost) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (dtype == MJMPC_F32)
        e = arm_rollout_launch<float>(h, h->model_f32, h->state, 1, 1, h->nu, d_action, nullptr, (float*)d_cost,
                                             nullptr, nullptr, (float*)d_next_obs, h->state, h->diag, s, arm_fuse(h));
    else if (dtype == MJMPC_F64)
        e = arm_rollout_launch<double>(h, h->model_f64, h->state, 1, 1, h->nu, d_action, nullptr, (double*)d_cost,
                                              nullptr, nullptr, (double*)d_next_obs, h->state, h->diag, s, arm_fuse(h));
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "arm_step_state launch");
    return 0;
}

int mjmpc_arm_solver_failures(mjmpc_arm_t h, uint32_t* count) {
    if (!h || !count) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned c = 0;
    HIP_TRY(hipMemcpy(&c, h->diag, sizeof(unsigned), hipMemcpyDeviceToHost));
    *count = c;
    return 0;
}

int mjmpc_arm_diverged(mjmpc_arm_t h, uint32_t* count) {
    if (!h || !count) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned c = 0;
    HIP_TRY(hipMemcpy(&c, h->diag + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    *count = c;
    return 0;
}

#ifdef MJMPC_STAMPS
// developer builds only (not declared in include/mjmpc_amd.h): read and clear the phase clocks
extern "C" int mjmpc_debug_stamps(mjmpc_arm_t h, unsigned long long* out32) {
    if (!h || !out32) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out32, (char*)h->diag + 16, 8 * 64, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset((char*)h->diag + 16, 0, 8 * 64));
    return 0;
}
#endif

/* ---- tree engine ------------------------------------------------------------------------------------ */
static_assert(MJMPC_TREE_BLOB_LEN == mjmpc::TREE_BLOB_LEN && MJMPC_TREE_DEVICE_STATE_LEN == mjmpc::TREE_STATE_LEN &&
              MJMPC_TREE_STATE_LEN == mjmpc::TREE_PUBLIC_STATE_LEN, "include/mjmpc_amd.h and csrc/tree_model.h disagree");
#define MJMPC_TREE_DIAG_BYTES (8 + 8 * mjmpc::TREE_STAT_SLOTS + 8)  /* counters, the developer clocks of -DTREE_STATS builds, the real env's resets (TREE_DIAG_ENV_RESETS) */
#ifdef TREE_STATS
// developer builds only (not declared in include/mjmpc_amd.h): read and clear the phase clocks / iteration counts
extern "C" int mjmpc_debug_tree_stats(mjmpc_tree_t h, unsigned long long* out48) {
    if (!h || !out48) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out48, (char*)h->diag + 8, 8 * mjmpc::TREE_STAT_SLOTS, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset((char*)h->diag + 8, 0, 8 * mjmpc::TREE_STAT_SLOTS));
    return 0;
}
#endif
// MuJoCo's layout (qpos[nq], qvel[nv], target[3]) -> the device state vector: one coordinate per LINK, a ball joint's
// quaternion as x, y, z in its three links' entries and w in its first link's w entry, a free joint's translations relative
// to the body position (the links are slides from there)
static void tree_pack_state(const mjmpc_tree_s* h, const double* qpos, const double* qvel, const double* target, double* st) {
    std::memset(st, 0, sizeof(double) * MJMPC_TREE_DEVICE_STATE_LEN);
    const double* b = h->topo.data();
    for (int l = 0; l < h->nv; ++l) {
        const int kind = (int)b[mjmpc::T_JTYPE + l], adr = (int)b[mjmpc::T_QADR + l];
        st[mjmpc::TREE_QW + l] = 1.0;
        if (kind == mjmpc::LINK_BALL_X) {
            // the device quaternion is relative to the qpos0 pose: q_link = conj(q0) * qpos (a free joint's qpos is absolute)
            const double w0 = b[mjmpc::T_QW0 + l], x0 = -b[mjmpc::T_QOFF + l], y0 = -b[mjmpc::T_QOFF + l + 1], z0 = -b[mjmpc::T_QOFF + l + 2];
            const double w = qpos[adr], x = qpos[adr + 1], y = qpos[adr + 2], z = qpos[adr + 3];
            st[mjmpc::TREE_QW + l] = w0 * w - x0 * x - y0 * y - z0 * z;
            st[l] = w0 * x + x0 * w + y0 * z - z0 * y;
            st[l + 1] = w0 * y - x0 * z + y0 * w + z0 * x;
            st[l + 2] = w0 * z + x0 * y - y0 * x + z0 * w;
        } else if (kind <= mjmpc::LINK_SLIDE) {
            st[l] = qpos[adr] - b[mjmpc::T_QOFF + l];
        }
        st[mjmpc::TL + l] = qvel[l];
    }
    std::memcpy(st + 2 * mjmpc::TL, target, sizeof(double) * 3);
}
static void tree_unpack_state(const mjmpc_tree_s* h, const double* st, double* qpos, double* qvel) {
    const double* b = h->topo.data();
    for (int l = 0; l < h->nv; ++l) {
        const int kind = (int)b[mjmpc::T_JTYPE + l], adr = (int)b[mjmpc::T_QADR + l];
        if (kind == mjmpc::LINK_BALL_X) {          // qpos = q0 * q_link
            const double w0 = b[mjmpc::T_QW0 + l], x0 = b[mjmpc::T_QOFF + l], y0 = b[mjmpc::T_QOFF + l + 1], z0 = b[mjmpc::T_QOFF + l + 2];
            const double w = st[mjmpc::TREE_QW + l], x = st[l], y = st[l + 1], z = st[l + 2];
            qpos[adr] = w0 * w - x0 * x - y0 * y - z0 * z;
            qpos[adr + 1] = w0 * x + x0 * w + y0 * z - z0 * y;
            qpos[adr + 2] = w0 * y - x0 * z + y0 * w + z0 * x;
            qpos[adr + 3] = w0 * z + x0 * y - y0 * x + z0 * w;
        } else if (kind <= mjmpc::LINK_SLIDE) {
            qpos[adr] = st[l] + b[mjmpc::T_QOFF + l];
        }
        qvel[l] = st[mjmpc::TL + l];
    }
}

static bool tree_blob_is_full(const double* blob, int nv) {
    bool full = blob[mjmpc::T_ANY_FRICTION] != 0.0 || (int)blob[mjmpc::T_N_SPHERE] > 8 || blob[mjmpc::T_DENSITY] > 0.0 ||
                blob[mjmpc::T_VISCOSITY] > 0.0 || blob[mjmpc::T_GEN] != 0.0;
    for (int l = 0; l < nv; ++l) full = full || (int)blob[mjmpc::T_JTYPE + l] == 2 || blob[mjmpc::T_STIFFNESS + l] != 0.0;
    return full;
}

// the tables a shard block must share with the engine's create-time block: they fix the kernel instantiation, the
// launch shape and the factorisation schedule (dynamics randomization edits masses, inertias, damping, contact radii and
// friction - never the tree)
static bool tree_same_topology(const double* a, const double* b) {
    auto same = [&](int off, int n) { return std::memcmp(a + off, b + off, sizeof(double) * n) == 0; };
    return same(mjmpc::T_PARENT, mjmpc::TL) && same(mjmpc::T_EPARENT, mjmpc::TL) && same(mjmpc::T_SUBSIZE, mjmpc::TL) && same(mjmpc::T_ANC, 5 * mjmpc::TL) &&
           same(mjmpc::T_JTYPE, mjmpc::TL) && same(mjmpc::T_ACT, mjmpc::TL) && same(mjmpc::T_DEPTH, mjmpc::TL) &&
           same(mjmpc::T_N_ROUNDS, 1) && same(mjmpc::T_ELIM, (mjmpc::TL - 1) * mjmpc::TL) && same(mjmpc::T_NV, 1) &&
           same(mjmpc::T_NU, 1) && same(mjmpc::T_TASK, 1) && same(mjmpc::T_OBS_SKIP, 1) && same(mjmpc::T_JUMPS, 1) &&
           same(mjmpc::T_NQ, 1) && same(mjmpc::T_QADR, mjmpc::TL) && same(mjmpc::T_HAS_BALL, 1) && same(mjmpc::T_QW0, mjmpc::TL);
}

// what every rollout launch of the engine carries besides its own fusions: the reset records (one per model shard; shard >= 0:
// a launch that runs that one shard's model block)
static mjmpc::TreeFusion tree_fuse(const mjmpc_tree_s* h, int shard = -1) {
    mjmpc::TreeFusion f;
    if (h->reset_rec) {
        f.reset_rec = h->reset_rec + (shard > 0 && h->n_shards > 1 ? (size_t)shard * mjmpc::TREE_RESET_LEN : 0);
        f.reset_stride = (shard < 0 && h->n_shards > 1) ? mjmpc::TREE_RESET_LEN : 0;
    }
    f.inf_on_reset = h->inf_on_reset;
    return f;
}

// The reset records of `n_shards` model blocks (host, TREE_BLOB_LEN each): per block ONE substep of the f64 kernel itself
// from the reset state (qpos0 = the device's zero coordinates and identity quaternions, zero velocity, zero controls) on a
// copy of the block with frame_skip 1 - state_out receives the state after it, site_out / axis_out the site and the object
// axis at the reset state.  Synchronous; called when the engine is created and when its model blocks are replaced.
// (as arm_make_reset_records: the records come back in *out, `full` is the kernel choice of the NEW blocks, the launches count
// into a scratch counter block)
static int tree_make_reset_records(mjmpc_tree_s* h, const double* blobs, int n_shards, bool full, double** out) {
    const size_t L = (size_t)mjmpc::TREE_BLOB_LEN, R = (size_t)mjmpc::TREE_RESET_LEN;
    double *rec = nullptr, *tmp = nullptr, *st = nullptr;
    unsigned* sdiag = nullptr;
    HIP_TRY(hipMalloc(&rec, sizeof(double) * R * n_shards));
    hipError_t e = hipMalloc(&tmp, sizeof(double) * L);
    if (e == hipSuccess) e = hipMalloc(&sdiag, MJMPC_TREE_DIAG_BYTES);
    if (e == hipSuccess) e = hipMemset(sdiag, 0, MJMPC_TREE_DIAG_BYTES);
    if (e == hipSuccess) e = hipMalloc(&st, sizeof(double) * mjmpc::TREE_STATE_LEN);
    if (e == hipSuccess) e = hipMemset(rec, 0, sizeof(double) * R * n_shards);
    std::vector<double> b(L), s0(mjmpc::TREE_STATE_LEN, 0.0);
    for (int l = 0; l < mjmpc::TL; ++l) s0[mjmpc::TREE_QW + l] = 1.0;
    if (e == hipSuccess) e = hipMemcpy(st, s0.data(), sizeof(double) * s0.size(), hipMemcpyHostToDevice);
    for (int k = 0; k < n_shards && e == hipSuccess; ++k) {
        std::memcpy(b.data(), blobs + (size_t)k * L, sizeof(double) * L);
        b[mjmpc::T_FRAME_SKIP] = 1.0;
        e = hipMemcpy(tmp, b.data(), sizeof(double) * L, hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
        double* rk = rec + (size_t)k * R;
        mjmpc::TreeFusion f;
        f.axis_out = rk + mjmpc::TREE_STATE_LEN + 3;
        e = mjmpc::launch_tree_rollout<double>(tmp, 1, h->max_path, full, h->nv, st, 1, 1, h->nu, h->zero_action, nullptr,
                                               (double*)h->scratch, nullptr, nullptr, nullptr, sdiag, nullptr, rk, nullptr,
                                               rk + mjmpc::TREE_STATE_LEN, 1, h->gen, f);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    hipFree(tmp);
    hipFree(st);
    hipFree(sdiag);
    if (e != hipSuccess) {
        hipFree(rec);
        return hip_fail(e, "reset record");
    }
    *out = rec;
    return 0;
}

static int tree_create_impl(mjmpc_tree_s* h, const double* blob, int n_blob) {
    std::vector<float> f32(blob, blob + n_blob);
    HIP_TRY(hipMalloc(&h->model_f32, sizeof(float) * n_blob));
    HIP_TRY(hipMalloc(&h->model_f64, sizeof(double) * n_blob));
    HIP_TRY(hipMalloc(&h->state, sizeof(double) * MJMPC_TREE_DEVICE_STATE_LEN));
    HIP_TRY(hipMalloc(&h->diag, MJMPC_TREE_DIAG_BYTES));
    HIP_TRY(hipMemcpy(h->model_f32, f32.data(), sizeof(float) * n_blob, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->model_f64, blob, sizeof(double) * n_blob, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->state, 0, sizeof(double) * MJMPC_TREE_DEVICE_STATE_LEN));
    HIP_TRY(hipMemset(h->diag, 0, MJMPC_TREE_DIAG_BYTES));
    HIP_TRY(hipMalloc(&h->zero_action, sizeof(double) * 40));
    HIP_TRY(hipMemset(h->zero_action, 0, sizeof(double) * 40));
    h->scratch = h->zero_action + 32;
    HIP_TRY(hipHostMalloc(&h->pinned, sizeof(double) * MJMPC_TREE_DEVICE_STATE_LEN * 4));
    for (int k = 0; k < 4; ++k) HIP_TRY(hipEventCreateWithFlags(&h->staged[k], hipEventDisableTiming));
    return tree_make_reset_records(h, blob, 1, h->full, &h->reset_rec);
}

int mjmpc_tree_create(const double* blob, int n_blob, int device, mjmpc_tree_t* out) {
    if (!blob || !out) return fail(MJMPC_E_BADARG, "null argument");
    if (n_blob != mjmpc::TREE_BLOB_LEN)
        return fail(MJMPC_E_BADMODEL, "tree model blob has %d scalars, expected %d", n_blob, (int)mjmpc::TREE_BLOB_LEN);
    const int nv = (int)blob[mjmpc::T_NV];
    if (nv < 1 || nv > mjmpc::TL) return fail(MJMPC_E_BADMODEL, "nv = %d outside 1..%d", nv, mjmpc::TL);
    if ((int)blob[mjmpc::T_N_SPHERE] > mjmpc::TREE_MAX_SPHERES) return fail(MJMPC_E_BADMODEL, "too many contact points");
    const int nu = (int)blob[mjmpc::T_NU], task = (int)blob[mjmpc::T_TASK], skip = (int)blob[mjmpc::T_OBS_SKIP];
    if (nu < 1 || nu > nv || task < 0 || task > 2 || skip < 0 || skip >= nv)
        return fail(MJMPC_E_BADMODEL, "nu = %d, task = %d, obs_skip = %d do not fit nv = %d", nu, task, skip, nv);
    if (mjmpc_device_count() <= device) return fail(MJMPC_E_NOGPU, "HIP device %d not present", device);
    HIP_TRY(hipSetDevice(device));
    mjmpc_tree_s* h = new mjmpc_tree_s();
    h->device = device;
    h->nv = nv;
    h->nu = (int)blob[mjmpc::T_NU];
    h->nq = (int)blob[mjmpc::T_NQ];
    if (h->nq < nv || h->nq > mjmpc::TREE_NQ_MAX) {
        delete h;
        return fail(MJMPC_E_BADMODEL, "nq = %d does not fit nv = %d", (int)blob[mjmpc::T_NQ], nv);
    }
    h->d_obs = (int)blob[mjmpc::T_TASK] == 1 ? h->nq + nv - (int)blob[mjmpc::T_OBS_SKIP] : h->nq + nv + 6;
    h->full = tree_blob_is_full(blob, nv);
    h->gen = (int)blob[mjmpc::T_GEN];
    for (int l = 0; l < nv; ++l) h->max_path = std::max(h->max_path, (int)blob[mjmpc::T_DEPTH + l] + 1);
    h->topo.assign(blob, blob + n_blob);
    if (int rc = tree_create_impl(h, blob, n_blob)) {       // a failed allocation leaves nothing behind
        mjmpc_tree_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int mjmpc_tree_set_shard_models(mjmpc_tree_t h, const double* blobs, int n_shards) {
    if (!h || !blobs || n_shards < 1) return fail(MJMPC_E_BADARG, "bad argument");
    if (h->n_state_shards > 1 && n_shards > 1 && n_shards != h->n_state_shards)
        return fail(MJMPC_E_BADARG, "%d model shards but %d per-shard start states", n_shards, h->n_state_shards);
    HIP_TRY(hipSetDevice(h->device));
    const size_t L = (size_t)mjmpc::TREE_BLOB_LEN, n = (size_t)n_shards * L;
    bool full = false;
    for (int s = 0; s < n_shards; ++s) {
        const double* b = blobs + (size_t)s * L;
        if ((int)b[mjmpc::T_N_SPHERE] > mjmpc::TREE_MAX_SPHERES || !tree_same_topology(b, h->topo.data()))
            return fail(MJMPC_E_BADMODEL, "shard %d does not have the engine's topology / dimensions", s);
        full = full || tree_blob_is_full(b, h->nv);
        if ((int)b[mjmpc::T_GEN] != h->gen)
            return fail(MJMPC_E_BADMODEL, "shard %d needs a different kernel instantiation than the engine's model", s);
    }
    std::vector<float> f32(blobs, blobs + n);
    HIP_TRY(hipDeviceSynchronize());
    float* m32 = nullptr;
    double* m64 = nullptr;
    HIP_TRY(hipMalloc(&m32, sizeof(float) * n));
    if (hipError_t e = hipMalloc(&m64, sizeof(double) * n); e != hipSuccess) {
        hipFree(m32);
        return hip_fail(e, "hipMalloc");
    }
    hipError_t e = hipMemcpy(m32, f32.data(), sizeof(float) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m64, blobs, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(m32);
        hipFree(m64);
        return hip_fail(e, "hipMemcpy");
    }
    double* rec = nullptr;      // the reset records of the NEW blocks first: a failure leaves the engine as it was
    if (int rc = tree_make_reset_records(h, blobs, n_shards, full, &rec); rc != 0) {
        hipFree(m32);
        hipFree(m64);
        return rc;
    }
    hipFree(h->model_f32);
    hipFree(h->model_f64);
    h->model_f32 = m32;
    h->model_f64 = m64;
    h->n_shards = n_shards;
    h->full = full;             // of the NEW set of blocks (they replace the old ones)
    if (h->reset_rec) h->reset_retired.push_back(h->reset_rec);
    h->reset_rec = rec;
    return 0;
}

int mjmpc_tree_set_shard_states(mjmpc_tree_t h, const double* states, int n_shards, void* stream) {
    if (!h || (n_shards > 0 && !states) || n_shards < 0) return fail(MJMPC_E_BADARG, "bad argument");
    if (h->n_shards > 1 && n_shards > 1 && n_shards != h->n_shards)
        return fail(MJMPC_E_BADARG, "%d per-shard start states but %d model shards", n_shards, h->n_shards);
    HIP_TRY(hipSetDevice(h->device));
    if (n_shards != h->n_state_shards) {
        HIP_TRY(hipDeviceSynchronize());
        hipFree(h->shard_states);
        h->shard_states = nullptr;
        h->n_state_shards = 0;
        if (n_shards > 0) HIP_TRY(hipMalloc(&h->shard_states, sizeof(double) * MJMPC_TREE_DEVICE_STATE_LEN * n_shards));
        h->n_state_shards = n_shards;
    }
    if (n_shards > 0) {
        std::vector<double> packed((size_t)n_shards * MJMPC_TREE_DEVICE_STATE_LEN);
        for (int k = 0; k < n_shards; ++k) {
            const double* pub = states + (size_t)k * MJMPC_TREE_STATE_LEN;     // qpos[40] | qvel[32] | target[3] | -
            tree_pack_state(h, pub, pub + mjmpc::TREE_NQ_MAX, pub + mjmpc::TREE_NQ_MAX + mjmpc::TL,
                            packed.data() + (size_t)k * MJMPC_TREE_DEVICE_STATE_LEN);
        }
        HIP_TRY(hipMemcpyAsync(h->shard_states, packed.data(), sizeof(double) * packed.size(), hipMemcpyHostToDevice,
                               (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));      // `packed` is pageable host memory
    }
    return 0;
}

int mjmpc_tree_destroy(mjmpc_tree_t h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipFree(h->model_f32);
    hipFree(h->model_f64);
    hipFree(h->state);
    hipFree(h->diag);
    hipFree(h->zero_action);
    hipFree(h->shard_states);
    hipFree(h->reset_rec);
    for (double* p : h->reset_retired) hipFree(p);
    if (h->pinned) hipHostFree(h->pinned);
    for (int k = 0; k < 4; ++k) if (h->staged[k]) hipEventDestroy(h->staged[k]);
    delete h;
    return 0;
}

int mjmpc_tree_dims(mjmpc_tree_t h, int* nv, int* nu, int* d_obs) {
    if (!h) return fail(MJMPC_E_BADARG, "null engine");
    if (nv) *nv = h->nv;
    if (nu) *nu = h->nu;
    if (d_obs) *d_obs = h->d_obs;
    return 0;
}

int mjmpc_tree_nq(mjmpc_tree_t h) { return h ? h->nq : -1; }

int mjmpc_tree_set_state(mjmpc_tree_t h, const double* qpos, const double* qvel, const double* target_pos,
                         void* stream) {
    if (!h || !qpos || !qvel || !target_pos) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(h->device));
    // staging ring as mjmpc_arm_set_state: wait only for the copy that last used THIS slot (four calls ago), never for
    // the stream - a captured control iteration still running on `s` keeps running while the next state is staged
    const int slot = h->stage_next;
    h->stage_next = (slot + 1) & 3;
    HIP_TRY(hipEventSynchronize(h->staged[slot]));
    double* st = h->pinned + (size_t)slot * MJMPC_TREE_DEVICE_STATE_LEN;
    tree_pack_state(h, qpos, qvel, target_pos, st);
    HIP_TRY(hipMemcpyAsync(h->state, st, sizeof(double) * MJMPC_TREE_DEVICE_STATE_LEN, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(h->staged[slot], s));
    return 0;
}

int mjmpc_tree_rollout(mjmpc_tree_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                       void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream) {
    if (!h || !d_mean || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    if (P < 0 || H < 0) return fail(MJMPC_E_BADARG, "negative size");
    const int nss = h->n_state_shards > 1 ? h->n_state_shards : 1;
    if (P % h->n_shards != 0 || P % nss != 0)
        return fail(MJMPC_E_BADARG, "%lld particles do not divide into %d shards", (long long)P, std::max(h->n_shards, nss));
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    const double* st = nss > 1 ? h->shard_states : h->state;
    if (dtype == MJMPC_F32)
        e = mjmpc::launch_tree_rollout<float>(h->model_f32, h->n_shards, h->max_path, h->full, h->nv, st, (long)P, H, h->nu, d_mean,
                                              (const float*)d_noise, (float*)d_costs, (float*)d_actions, (float*)d_obs,
                                              (float*)d_next_obs, h->diag, s, nullptr, nullptr, nullptr, nss, h->gen, tree_fuse(h));
    else if (dtype == MJMPC_F64)
        e = mjmpc::launch_tree_rollout<double>(h->model_f64, h->n_shards, h->max_path, h->full, h->nv, st, (long)P, H, h->nu, d_mean,
                                               (const double*)d_noise, (double*)d_costs, (double*)d_actions,
                                               (double*)d_obs, (double*)d_next_obs, h->diag, s, nullptr, nullptr, nullptr, nss, h->gen, tree_fuse(h));
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "tree_rollout launch");
    return 0;
}

int mjmpc_tree_rollout_fused(mjmpc_tree_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                             const double* d_filter_coeffs, const double* d_gseq, void* d_costs, void* d_actions, double* d_q0,
                             void* stream) {
    if (!h || !d_mean || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    if ((d_q0 != nullptr) != (d_gseq != nullptr)) return fail(MJMPC_E_BADARG, "d_q0 and d_gseq go together");
    if (P < 0 || H < 0) return fail(MJMPC_E_BADARG, "negative size");
    const int nss = h->n_state_shards > 1 ? h->n_state_shards : 1;
    if (P % h->n_shards != 0 || P % nss != 0)
        return fail(MJMPC_E_BADARG, "%lld particles do not divide into %d shards", (long long)P, std::max(h->n_shards, nss));
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    mjmpc::TreeFusion fuse = tree_fuse(h);
    fuse.filt = d_filter_coeffs;
    fuse.gseq = d_gseq;
    fuse.q0_out = d_q0;
    hipError_t e;
    const double* st = nss > 1 ? h->shard_states : h->state;
    if (dtype == MJMPC_F32)
        e = mjmpc::launch_tree_rollout<float>(h->model_f32, h->n_shards, h->max_path, h->full, h->nv, st, (long)P, H, h->nu, d_mean,
                                              (const float*)d_noise, (float*)d_costs, (float*)d_actions, nullptr, nullptr, h->diag,
                                              s, nullptr, nullptr, nullptr, nss, h->gen, fuse);
    else if (dtype == MJMPC_F64)
        e = mjmpc::launch_tree_rollout<double>(h->model_f64, h->n_shards, h->max_path, h->full, h->nv, st, (long)P, H, h->nu, d_mean,
                                               (const double*)d_noise, (double*)d_costs, (double*)d_actions, nullptr, nullptr,
                                               h->diag, s, nullptr, nullptr, nullptr, nss, h->gen, fuse);
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "tree_rollout_fused launch");
    return 0;
}

int mjmpc_tree_rollout_cl(mjmpc_tree_t h, int dtype, int64_t P, int H, const double* d_weights, const void* d_noise,
                          void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream) {
    if (!h || !d_weights || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    if (P < 0 || H < 0) return fail(MJMPC_E_BADARG, "negative size");
    const int nss = h->n_state_shards > 1 ? h->n_state_shards : 1;
    if (P % h->n_shards != 0 || P % nss != 0)
        return fail(MJMPC_E_BADARG, "%lld particles do not divide into %d shards", (long long)P, std::max(h->n_shards, nss));
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    double* st = nss > 1 ? h->shard_states : h->state;
    // the first action depends on the fresh observation, whose tracked site comes out of a kinematics pass: a
    // one-particle, one-step launch per start state (its cost lands in the workspace and is discarded) leaves it in
    // the state vector
    for (int k = 0; k < nss && e == hipSuccess; ++k) {
        double* sk = st + (size_t)k * MJMPC_TREE_DEVICE_STATE_LEN;
        double* site0 = sk + 2 * mjmpc::TL + 3;
        if (dtype == MJMPC_F32)
            e = mjmpc::launch_tree_rollout<float>(h->model_f32 + (h->n_shards > 1 ? (size_t)k * mjmpc::TREE_BLOB_LEN : 0), 1, h->max_path,
                                                  h->full, h->nv, sk, 1, 1, h->nu, h->zero_action, nullptr, (float*)h->scratch,
                                                  nullptr, nullptr, nullptr, h->diag, s, nullptr, nullptr, site0, 1, h->gen, tree_fuse(h, k));
        else if (dtype == MJMPC_F64)
            e = mjmpc::launch_tree_rollout<double>(h->model_f64 + (h->n_shards > 1 ? (size_t)k * mjmpc::TREE_BLOB_LEN : 0), 1, h->max_path,
                                                   h->full, h->nv, sk, 1, 1, h->nu, h->zero_action, nullptr, (double*)h->scratch,
                                                   nullptr, nullptr, nullptr, h->diag, s, nullptr, nullptr, site0, 1, h->gen, tree_fuse(h, k));
        else
            return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    }
    if (e == hipSuccess) {
        if (dtype == MJMPC_F32)
            e = mjmpc::launch_tree_rollout<float>(h->model_f32, h->n_shards, h->max_path, h->full, h->nv, st, (long)P, H, h->nu,
                                                  d_weights, (const float*)d_noise, (float*)d_costs, (float*)d_actions, (float*)d_obs,
                                                  (float*)d_next_obs, h->diag, s, nullptr, d_weights, nullptr, nss, h->gen, tree_fuse(h));
        else
            e = mjmpc::launch_tree_rollout<double>(h->model_f64, h->n_shards, h->max_path, h->full, h->nv, st, (long)P, H, h->nu,
                                                   d_weights, (const double*)d_noise, (double*)d_costs, (double*)d_actions,
                                                   (double*)d_obs, (double*)d_next_obs, h->diag, s, nullptr, d_weights, nullptr, nss, h->gen, tree_fuse(h));
    }
    if (e != hipSuccess) return hip_fail(e, "tree_rollout_cl launch");
    return 0;
}

int mjmpc_tree_step_state(mjmpc_tree_t h, int dtype, const double* d_action, void* d_cost, void* d_next_obs, void* stream) {
    if (!h || !d_action || !d_cost) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    // one particle, one env step, no noise, shard 0's model; the state vector is advanced in place
    if (dtype == MJMPC_F32)
        e = mjmpc::launch_tree_rollout<float>(h->model_f32, 1, h->max_path, h->full, h->nv, h->state, 1, 1, h->nu, d_action, nullptr,
                                              (float*)d_cost, nullptr, nullptr, (float*)d_next_obs, h->diag, s, h->state, nullptr, nullptr, 1, h->gen, tree_fuse(h, 0));
    else if (dtype == MJMPC_F64)
        e = mjmpc::launch_tree_rollout<double>(h->model_f64, 1, h->max_path, h->full, h->nv, h->state, 1, 1, h->nu, d_action, nullptr,
                                               (double*)d_cost, nullptr, nullptr, (double*)d_next_obs, h->diag, s, h->state, nullptr, nullptr, 1, h->gen, tree_fuse(h, 0));
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "tree_step_state launch");
    return 0;
}

int mjmpc_tree_get_state(mjmpc_tree_t h, double* qpos, double* qvel, void* stream) {
    if (!h || !qpos || !qvel) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double st[MJMPC_TREE_DEVICE_STATE_LEN];
    HIP_TRY(hipMemcpyAsync(st, h->state, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    tree_unpack_state(h, st, qpos, qvel);
    return 0;
}

int mjmpc_tree_solver_failures(mjmpc_tree_t h, uint32_t* count) {
    if (!h || !count) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned c = 0;
    HIP_TRY(hipMemcpy(&c, h->diag, sizeof(unsigned), hipMemcpyDeviceToHost));
    *count = c;
    return 0;
}

int mjmpc_tree_diverged(mjmpc_tree_t h, uint32_t* count) {
    if (!h || !count) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned c = 0;
    HIP_TRY(hipMemcpy(&c, h->diag + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    *count = c;
    return 0;
}

int mjmpc_tree_env_resets(mjmpc_tree_t h, uint32_t* count) {
    if (!h || !count) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned c = 0;
    HIP_TRY(hipMemcpy(&c, h->diag + mjmpc::TREE_DIAG_ENV_RESETS, sizeof(unsigned), hipMemcpyDeviceToHost));
    *count = c;
    return 0;
}

int mjmpc_arm_env_resets(mjmpc_arm_t h, uint32_t* count) {
    if (!h || !count) return fail(MJMPC_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned c = 0;
    HIP_TRY(hipMemcpy(&c, h->diag + 2, sizeof(unsigned), hipMemcpyDeviceToHost));
    *count = c;
    return 0;
}

int mjmpc_tree_set_reset_returns(mjmpc_tree_t h, int inf_returns) {
    if (!h) return fail(MJMPC_E_BADARG, "null argument");
    h->inf_on_reset = inf_returns ? 1 : 0;
    return 0;
}

int mjmpc_arm_set_reset_returns(mjmpc_arm_t h, int inf_returns) {
    if (!h) return fail(MJMPC_E_BADARG, "null argument");
    h->inf_on_reset = inf_returns ? 1 : 0;
    return 0;
}

int mjmpc_analytic_rollout(int kind, const double* d_params, int n_state, int n_action, const double* d_state, int dtype,
                           int64_t P, int H, const double* d_mean, const void* d_noise, void* d_costs, void* d_actions,
                           void* d_obs, void* d_next_obs, int closed_loop_linear, void* stream) {
    if (!d_params || !d_state || !d_mean || !d_costs) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (dtype == MJMPC_F32)
        e = mjmpc::launch_analytic_rollout<float>(kind, d_params, n_state, n_action, d_state, (long)P, H, d_mean,
                                                  (const float*)d_noise, (float*)d_costs, (float*)d_actions,
                                                  (float*)d_obs, (float*)d_next_obs, s, closed_loop_linear);
    else if (dtype == MJMPC_F64)
        e = mjmpc::launch_analytic_rollout<double>(kind, d_params, n_state, n_action, d_state, (long)P, H, d_mean,
                                                   (const double*)d_noise, (double*)d_costs, (double*)d_actions,
                                                   (double*)d_obs, (double*)d_next_obs, s, closed_loop_linear);
    else
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    if (e != hipSuccess) return hip_fail(e, "analytic rollout launch");
    return 0;
}

// ---- update / noise entry points -------------------------------------------------------------------
#define DISPATCH(dtype, CALL_F32, CALL_F64)                                   \
    do {                                                                      \
        hipError_t e_;                                                        \
        if ((dtype) == MJMPC_F32) e_ = (CALL_F32);                            \
        else if ((dtype) == MJMPC_F64) e_ = (CALL_F64);                       \
        else return fail(MJMPC_E_BADARG, "unknown dtype %d", (int)(dtype));   \
        if (e_ != hipSuccess) return hip_fail(e_, __func__);                  \
        return 0;                                                             \
    } while (0)
#define PLAIN(CALL)                                           \
    do {                                                      \
        hipError_t e_ = (CALL);                               \
        if (e_ != hipSuccess) return hip_fail(e_, __func__);  \
        return 0;                                             \
    } while (0)

int64_t mjmpc_update_workspace_bytes(int64_t P, int H, int A) {
    return (int64_t)sizeof(double) * mjmpc::update_workspace_doubles((long)P, H, A);
}

int mjmpc_softmax_record_len(int H, int A, int tbw) { return 2 * (tbw ? H : 1) + H * A + A * A; }

int mjmpc_traj_cost(int dtype, int64_t P, int H, int A, const void* d_costs, const double* d_gseq, int gamma_zero,
                    void* d_ws, void* stream) {
    if (!d_costs || !d_gseq || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::traj_cost<float>((const float*)d_costs, nullptr, nullptr, nullptr, d_gseq, gamma_zero, 1.0, 1, 0,
                                     (long)P, H, A, (double*)d_ws, s),
             mjmpc::traj_cost<double>((const double*)d_costs, nullptr, nullptr, nullptr, d_gseq, gamma_zero, 1.0, 1, 0,
                                      (long)P, H, A, (double*)d_ws, s));
}

double* mjmpc_workspace_q0(void* d_ws, int64_t P, int H, int A) {
    return mjmpc::workspace_q0((double*)d_ws, (long)P, H, A);
}

int mjmpc_td_lambda_returns(int dtype, int64_t P, int H, int A, const void* d_costs, const void* d_actions,
                            const void* d_qvals, const double* d_mean, const double* d_covinv, const double* d_wseq,
                            int wseq_has_zero, double beta, int alpha, double gamma, double td_lam, void* d_returns,
                            void* d_ws, void* stream) {
    if (!d_costs || !d_returns || !d_ws || (H > 1 && !d_wseq)) return fail(MJMPC_E_BADARG, "null argument");
    if (alpha == 0 && (!d_actions || !d_mean || !d_covinv)) return fail(MJMPC_E_BADARG, "control cost needs actions, mean, cov^-1");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::td_lambda_returns<float>((const float*)d_costs, (const float*)d_actions, (const float*)d_qvals, d_mean,
                                             d_covinv, d_wseq, wseq_has_zero, beta, alpha, gamma, td_lam, (long)P, H, A,
                                             (float*)d_returns, (double*)d_ws, s),
             mjmpc::td_lambda_returns<double>((const double*)d_costs, (const double*)d_actions, (const double*)d_qvals,
                                              d_mean, d_covinv, d_wseq, wseq_has_zero, beta, alpha, gamma, td_lam,
                                              (long)P, H, A, (double*)d_returns, (double*)d_ws, s));
}

int mjmpc_softmax_stats(int dtype, int64_t P, int H, int A, const void* d_costs, const void* d_actions,
                        const double* d_mean, const double* d_covinv, const double* d_gseq, int gamma_zero,
                        double lam, int alpha, int tbw, int want_cov, double* d_record, void* d_ws, void* stream) {
    if (!d_costs || !d_actions || !d_mean || !d_gseq || !d_record || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    if (alpha == 0 && !d_covinv) return fail(MJMPC_E_BADARG, "alpha == 0 needs d_covinv");
    if (tbw && want_cov) return fail(MJMPC_E_BADARG, "time_based_weights and want_cov are exclusive");
    if (!(lam > 0)) return fail(MJMPC_E_BADARG, "lam must be positive");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::softmax_stats<float>((const float*)d_costs, (const float*)d_actions, d_mean, d_covinv, d_gseq,
                                         gamma_zero, lam, alpha, tbw, want_cov, (long)P, H, A, d_record,
                                         (double*)d_ws, s),
             mjmpc::softmax_stats<double>((const double*)d_costs, (const double*)d_actions, d_mean, d_covinv, d_gseq,
                                          gamma_zero, lam, alpha, tbw, want_cov, (long)P, H, A, d_record,
                                          (double*)d_ws, s));
}

int mjmpc_softmax_combine(const double* d_records, int G, int H, int A, int tbw, double lam, double step_size,
                          int cov_mode, double P_total, double* d_mean, double* d_cov, double* d_value,
                          double* d_wnorm, void* stream) {
    if (!d_records || !d_mean || G < 1) return fail(MJMPC_E_BADARG, "bad argument");
    if (cov_mode && !d_cov) return fail(MJMPC_E_BADARG, "cov_mode needs d_cov");
    PLAIN(mjmpc::softmax_combine(d_records, G, H, A, tbw, lam, step_size, cov_mode, P_total, d_mean, d_cov, d_value,
                                 d_wnorm, (hipStream_t)stream));
}

int mjmpc_softmax_weights(int64_t P, int H, int A, const double* d_wnorm, void* d_ws, double* d_weights,
                          void* stream) {
    if (!d_wnorm || !d_ws || !d_weights) return fail(MJMPC_E_BADARG, "null argument");
    PLAIN(mjmpc::softmax_weights((long)P, d_wnorm, (double*)d_ws, H, A, d_weights, (hipStream_t)stream));
}

int mjmpc_cem_elite_sums(int dtype, int64_t P, int H, int A, const void* d_actions, const double* d_q_all,
                         int64_t P_all, int64_t offset, int64_t k, double* d_sum_record, void* d_ws, void* stream) {
    if (!d_actions || !d_sum_record || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::cem_elite_sums<float>((const float*)d_actions, d_q_all, (long)P_all, (long)offset, (long)k, (long)P,
                                          H, A, d_sum_record, (double*)d_ws, s),
             mjmpc::cem_elite_sums<double>((const double*)d_actions, d_q_all, (long)P_all, (long)offset, (long)k,
                                           (long)P, H, A, d_sum_record, (double*)d_ws, s));
}

int mjmpc_cem_elite_cov(int dtype, int64_t P, int H, int A, const void* d_actions, const double* d_mean,
                        const double* d_sum_records, int G, double* d_cov_record, void* d_ws, void* stream) {
    if (!d_actions || !d_mean || !d_sum_records || !d_cov_record || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::cem_elite_cov<float>((const float*)d_actions, d_mean, d_sum_records, G, (long)P, H, A, d_cov_record,
                                         (double*)d_ws, s),
             mjmpc::cem_elite_cov<double>((const double*)d_actions, d_mean, d_sum_records, G, (long)P, H, A,
                                          d_cov_record, (double*)d_ws, s));
}

int mjmpc_cem_final(const double* d_cov_records, int G, int64_t P, int H, int A, double n_elite, int full_cov,
                    double step_size, double* d_mean, double* d_cov, void* d_ws, void* stream) {
    if (!d_cov_records || !d_mean || !d_cov || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    PLAIN(mjmpc::cem_final(d_cov_records, G, (long)P, H, A, n_elite, full_cov, step_size, d_mean, d_cov,
                           (double*)d_ws, (hipStream_t)stream));
}

int mjmpc_cem_combine(const double* d_records, int G, int H, int A, double n_elite, int full_cov, double step_size,
                      double* d_mean, double* d_cov, void* stream) {
    if (!d_records || !d_mean || !d_cov) return fail(MJMPC_E_BADARG, "null argument");
    PLAIN(mjmpc::cem_combine(d_records, G, H, A, n_elite, full_cov, step_size, d_mean, d_cov, (hipStream_t)stream));
}

int mjmpc_cem_fused_supported(int64_t P_all, int64_t P, int64_t k, int H, int A) {
    return mjmpc::cem_fused_supported((long)P_all, (long)P, (long)k, H, A) ? 1 : 0;
}

int mjmpc_cem_select_moments(int dtype, int64_t P, int H, int A, const void* d_actions, const double* d_q_all, int64_t P_all,
                             int64_t offset, int64_t k, const double* d_mean, const double* d_cov,
                             const int64_t* d_step_counter, void* d_ws, void* stream) {
    if (!d_actions || !d_mean || !d_cov || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    if (!mjmpc::cem_fused_supported(d_q_all ? (long)P_all : (long)P, (long)P, (long)k, H, A))
        return fail(MJMPC_E_BADARG, "shape outside the fused CEM step (mjmpc_cem_fused_supported)");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::cem_select_moments<float>((const float*)d_actions, d_q_all, (long)P_all, (long)offset, (long)k, (long)P, H, A,
                                              d_mean, d_cov, (const long long*)d_step_counter, (double*)d_ws, s),
             mjmpc::cem_select_moments<double>((const double*)d_actions, d_q_all, (long)P_all, (long)offset, (long)k, (long)P, H,
                                               A, d_mean, d_cov, (const long long*)d_step_counter, (double*)d_ws, s));
}

int mjmpc_cem_record(int64_t P, int H, int A, int64_t k, const double* d_mean, double* d_record, void* d_ws, void* stream) {
    if (!d_mean || !d_record || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    PLAIN(mjmpc::cem_record((long)k, (long)P, H, A, d_mean, d_record, (double*)d_ws, (hipStream_t)stream));
}

int mjmpc_cem_finish(int dtype, int64_t P, int H, int A, int64_t k, const double* d_records, int G, double n_elite,
                     int full_cov, double step_size, int shift_mode, double* d_mean, double* d_cov, double* d_chol,
                     int* d_status, const double* d_grow_diag, double grow_scale, double* d_action_out,
                     double* h_action_pinned, int64_t* d_step_counter, void* d_next_noise, uint64_t seed, uint64_t offset,
                     int64_t particle_offset, void* d_ws, void* stream) {
    if (!d_mean || !d_cov || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    if (shift_mode > 1) return fail(MJMPC_E_BADARG, "shift_mode must be < 0 (none), 0 (null) or 1 (repeat)");
    if (!mjmpc::cem_fused_supported((long)P, (long)P, (long)k, H, A))
        return fail(MJMPC_E_BADARG, "shape outside the fused CEM step (mjmpc_cem_fused_supported)");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::cem_finish<float>(d_records, G, (long)k, (long)P, H, A, n_elite, full_cov, step_size, shift_mode, d_mean, d_cov,
                                      d_chol, d_status, d_grow_diag, grow_scale, d_action_out, h_action_pinned,
                                      (long long*)d_step_counter, (float*)d_next_noise, seed, offset, (long)particle_offset,
                                      (double*)d_ws, s),
             mjmpc::cem_finish<double>(d_records, G, (long)k, (long)P, H, A, n_elite, full_cov, step_size, shift_mode, d_mean, d_cov,
                                       d_chol, d_status, d_grow_diag, grow_scale, d_action_out, h_action_pinned,
                                       (long long*)d_step_counter, (double*)d_next_noise, seed, offset, (long)particle_offset,
                                       (double*)d_ws, s));
}

int mjmpc_rs_best(int dtype, int64_t P, int H, int A, const void* d_actions, int64_t offset, double* d_record,
                  void* d_ws, void* stream) {
    if (!d_actions || !d_record || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::rs_best<float>((const float*)d_actions, (long)offset, (long)P, H, A, d_record, (double*)d_ws, s),
             mjmpc::rs_best<double>((const double*)d_actions, (long)offset, (long)P, H, A, d_record, (double*)d_ws, s));
}

int mjmpc_rs_combine(const double* d_records, int G, int H, int A, double step_size, double* d_mean, void* stream) {
    if (!d_records || !d_mean || G < 1) return fail(MJMPC_E_BADARG, "bad argument");
    PLAIN(mjmpc::rs_combine(d_records, G, H, A, step_size, d_mean, (hipStream_t)stream));
}

static int fused_update(int dtype, int64_t P, int H, int A, const double* d_q0, const void* d_actions, double lam,
                        double step_size, int shift_mode, double* d_mean, double* d_action_out, double* d_record,
                        double* d_value, double* h_action_mapped, int64_t* d_step_counter, void* d_ws, void* stream,
                        const mjmpc::NextNoise* next) {
    if (!d_actions || !d_mean || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    if (!(lam > 0) || shift_mode > 1) return fail(MJMPC_E_BADARG, "bad lam / shift_mode");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::mppi_fused_update<float>(d_q0, (const float*)d_actions, lam, step_size, shift_mode, (long)P, H, A,
                                             d_mean, d_action_out, d_record, d_value, (double*)d_ws, s, h_action_mapped,
                                             (long long*)d_step_counter, next),
             mjmpc::mppi_fused_update<double>(d_q0, (const double*)d_actions, lam, step_size, shift_mode, (long)P, H, A,
                                              d_mean, d_action_out, d_record, d_value, (double*)d_ws, s,
                                              h_action_mapped, (long long*)d_step_counter, next));
}

int mjmpc_mppi_fused_update(int dtype, int64_t P, int H, int A, const double* d_q0, const void* d_actions, double lam,
                            double step_size, int shift_mode, double* d_mean, double* d_action_out, double* d_record,
                            double* d_value, double* h_action_mapped, int64_t* d_step_counter, void* d_ws,
                            void* stream) {
    return fused_update(dtype, P, H, A, d_q0, d_actions, lam, step_size, shift_mode, d_mean, d_action_out, d_record,
                        d_value, h_action_mapped, d_step_counter, d_ws, stream, nullptr);
}

int mjmpc_mppi_fused_update_draw_next(int dtype, int64_t P, int H, int A, const double* d_q0, const void* d_actions,
                                      double lam, double step_size, int shift_mode, double* d_mean,
                                      double* d_action_out, double* d_record, double* d_value, double* h_action_mapped,
                                      int64_t* d_step_counter, void* d_ws, void* d_next_noise, const double* d_chol,
                                      uint64_t seed, uint64_t offset, int64_t particle_offset, const int64_t* d_step,
                                      int chol_is_diagonal, void* stream) {
    if (!d_next_noise || !d_chol) return fail(MJMPC_E_BADARG, "null noise buffer or Cholesky factor");
    mjmpc::NextNoise nn{d_next_noise, d_chol, seed, offset, (long)particle_offset, (const long long*)d_step,
                        chol_is_diagonal};
    return fused_update(dtype, P, H, A, d_q0, d_actions, lam, step_size, shift_mode, d_mean, d_action_out, d_record,
                        d_value, h_action_mapped, d_step_counter, d_ws, stream, &nn);
}

int mjmpc_mppi_fused_combine(const double* d_records, int G, double P_total, int H, int A, double lam, double step_size,
                             int shift_mode, double* d_mean, double* d_action_out, double* d_value,
                             double* h_action_mapped, int64_t* d_step_counter, void* stream) {
    if (!d_records || !d_mean || G < 1) return fail(MJMPC_E_BADARG, "null argument");
    if (!(lam > 0) || shift_mode > 1) return fail(MJMPC_E_BADARG, "bad lam / shift_mode");
    PLAIN(mjmpc::mppi_fused_combine(d_records, G, P_total, lam, step_size, shift_mode, H, A, d_mean, d_action_out, d_value,
                                    h_action_mapped, (long long*)d_step_counter, (hipStream_t)stream));
}

int mjmpc_q0_sum(int64_t P, int H, int A, double* d_out, void* d_ws, void* stream) {
    if (!d_out || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    PLAIN(mjmpc::q0_sum((long)P, H, A, d_out, (double*)d_ws, (hipStream_t)stream));
}

int mjmpc_shift_mean(double* d_mean, int H, int A, int mode, const double* d_row, void* stream) {
    if (!d_mean || (mode == 2 && !d_row) || mode < 0 || mode > 2) return fail(MJMPC_E_BADARG, "bad argument");
    PLAIN(mjmpc::shift_mean(d_mean, H, A, mode, d_row, (hipStream_t)stream));
}

int mjmpc_step_tail(double* d_mean, int H, int A, int shift_mode, const double* d_row, double* d_action_out,
                    double* h_action_mapped, int64_t* d_step_counter, double* d_cov, const double* d_cov_diag,
                    double cov_scale, void* stream) {
    if (!d_mean || (shift_mode == 2 && !d_row) || shift_mode < 0 || shift_mode > 2 || A < 1 || A > 64 || H < 1)
        return fail(MJMPC_E_BADARG, "bad argument");
    PLAIN(mjmpc::step_tail(d_mean, H, A, shift_mode, d_row, d_action_out, h_action_mapped, (long long*)d_step_counter,
                           d_cov, d_cov_diag, cov_scale, (hipStream_t)stream));
}

int mjmpc_cholesky_lower(const double* d_cov, int A, double* d_chol, int* d_status, void* stream) {
    if (!d_cov || !d_chol || A < 1 || A > 64) return fail(MJMPC_E_BADARG, "bad argument (A <= 64)");
    PLAIN(mjmpc::cholesky_lower(d_cov, A, d_chol, d_status, (hipStream_t)stream));
}

int mjmpc_cov_add_diag(double* d_cov, int A, const double* d_diag, double scale, void* stream) {
    if (!d_cov || A < 1 || A > 64) return fail(MJMPC_E_BADARG, "bad argument (A <= 64)");
    PLAIN(mjmpc::cov_add_diag(d_cov, A, d_diag, scale, (hipStream_t)stream));
}

int mjmpc_filter_noise(int dtype, void* d_noise, int64_t P, int H, int A, const double* d_coeffs, void* stream) {
    if (!d_noise || !d_coeffs) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype, mjmpc::filter_noise<float>((float*)d_noise, (long)P, H, A, d_coeffs, s),
             mjmpc::filter_noise<double>((double*)d_noise, (long)P, H, A, d_coeffs, s));
}

int mjmpc_color_noise(int dtype, void* d_noise, int64_t rows, int A, const double* d_B, void* stream) {
    if (!d_noise || !d_B) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype, mjmpc::color_rows<float>((float*)d_noise, (long)rows, A, d_B, s),
             mjmpc::color_rows<double>((double*)d_noise, (long)rows, A, d_B, s));
}

int64_t mjmpc_mt19937_workspace_bytes(int64_t n_normals) { return (int64_t)mjmpc::mt_workspace_bytes((long)n_normals); }

int mjmpc_sample_noise_mt19937(int dtype, void* d_noise, int64_t n_normals, double scale, uint64_t seed,
                               const int64_t* d_step, void* d_ws, int* d_status, void* stream) {
    if (!d_noise || !d_ws) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::sample_noise_mt19937<float>((float*)d_noise, (long)n_normals, scale, seed, (const long long*)d_step,
                                                d_ws, d_status, s),
             mjmpc::sample_noise_mt19937<double>((double*)d_noise, (long)n_normals, scale, seed,
                                                 (const long long*)d_step, d_ws, d_status, s));
}

int64_t mjmpc_mt19937_stream_words(int64_t n_normals) { return 4 * (int64_t)mjmpc::mt_attempts_for((long)n_normals); }

int mjmpc_sample_noise_mt19937_jump(int dtype, void* d_noise, int64_t n_normals, double scale, uint64_t seed,
                                    const int64_t* d_step, const int32_t* d_jump_idx, const int32_t* d_jump_starts,
                                    int64_t head_words, int64_t seg_words, int n_segments, int64_t first_normal,
                                    void* d_ws, int* d_status, void* stream) {
    if (!d_noise || !d_ws || first_normal < 0) return fail(MJMPC_E_BADARG, "null argument");
    if (n_segments > 0) {
        if (!d_jump_idx || !d_jump_starts) return fail(MJMPC_E_BADARG, "jump tables missing");
        if (head_words < 19936 || head_words > 19968 || (head_words & 3) || seg_words < 2 * 624)
            return fail(MJMPC_E_BADARG, "head_words must be a multiple of 4 in [19936, 19968], seg_words >= 1248");
        if (n_segments > mjmpc::MT_MAX_SEGMENTS) return fail(MJMPC_E_BADARG, "too many segments (max 64)");
        if (head_words + seg_words * (int64_t)n_segments < mjmpc_mt19937_stream_words(first_normal + n_normals))
            return fail(MJMPC_E_BADARG, "segments do not cover the stream");
    }
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::sample_noise_mt19937<float>((float*)d_noise, (long)n_normals, scale, seed, (const long long*)d_step,
                                                d_ws, d_status, s, d_jump_idx, d_jump_starts, (long)head_words,
                                                (long)seg_words, n_segments, (long)first_normal),
             mjmpc::sample_noise_mt19937<double>((double*)d_noise, (long)n_normals, scale, seed,
                                                 (const long long*)d_step, d_ws, d_status, s, d_jump_idx, d_jump_starts,
                                                 (long)head_words, (long)seg_words, n_segments, (long)first_normal));
}

int mjmpc_sample_noise(int dtype, void* d_noise, int64_t P, int H, int A, const double* d_chol,
                       const double* d_coeffs, uint64_t seed, uint64_t offset, int64_t particle_offset,
                       const int64_t* d_step, int chol_is_diagonal, void* stream) {
    if (!d_noise || !d_chol) return fail(MJMPC_E_BADARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype,
             mjmpc::sample_noise<float>((float*)d_noise, (long)P, H, A, d_chol, d_coeffs, seed, offset,
                                        (long)particle_offset, (const long long*)d_step, s, chol_is_diagonal),
             mjmpc::sample_noise<double>((double*)d_noise, (long)P, H, A, d_chol, d_coeffs, seed, offset,
                                         (long)particle_offset, (const long long*)d_step, s, chol_is_diagonal));
}

}  // extern "C"

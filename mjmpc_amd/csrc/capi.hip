// extern "C" entry points declared in include/mjmpc_amd.h.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mjmpc_amd.h"
#include "arm_model.h"
#include "arm_rollout.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

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
    double* pinned = nullptr;       // host staging for set_state
};

extern "C" {

int mjmpc_abi_version(void) { return MJMPC_ABI_VERSION; }

const char* mjmpc_last_error(void) { return g_err; }

int mjmpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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
    h->nu = nv;
    h->d_obs = 2 * nv + 6;
    std::vector<float> f32(blob, blob + n_blob);
    HIP_TRY(hipMalloc(&h->model_f32, sizeof(float) * n_blob));
    HIP_TRY(hipMalloc(&h->model_f64, sizeof(double) * n_blob));
    HIP_TRY(hipMalloc(&h->state, sizeof(double) * MJMPC_ARM_STATE_LEN));
    HIP_TRY(hipMalloc(&h->diag, sizeof(unsigned)));
    HIP_TRY(hipHostMalloc(&h->pinned, sizeof(double) * MJMPC_ARM_STATE_LEN));
    HIP_TRY(hipMemcpy(h->model_f32, f32.data(), sizeof(float) * n_blob, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->model_f64, blob, sizeof(double) * n_blob, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->state, 0, sizeof(double) * MJMPC_ARM_STATE_LEN));
    HIP_TRY(hipMemset(h->diag, 0, sizeof(unsigned)));
    *out = h;
    return 0;
}

int mjmpc_arm_destroy(mjmpc_arm_t h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipFree(h->model_f32);
    hipFree(h->model_f64);
    hipFree(h->state);
    hipFree(h->diag);
    hipHostFree(h->pinned);
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
    HIP_TRY(hipStreamSynchronize(s));            // the staging buffer may still be in flight
    std::memset(h->pinned, 0, sizeof(double) * MJMPC_ARM_STATE_LEN);
    std::memcpy(h->pinned, qpos, sizeof(double) * h->nv);
    std::memcpy(h->pinned + mjmpc::LANES, qvel, sizeof(double) * h->nv);
    std::memcpy(h->pinned + 2 * mjmpc::LANES, target_pos, sizeof(double) * 3);
    HIP_TRY(hipMemcpyAsync(h->state, h->pinned, sizeof(double) * MJMPC_ARM_STATE_LEN, hipMemcpyHostToDevice, s));
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
    if (dtype == MJMPC_F32) {
        e = mjmpc::launch_arm_rollout<float>(h->model_f32, h->state, (long)P, H, h->nu, d_mean, (const float*)d_noise,
                                             (float*)d_costs, (float*)d_actions, (float*)d_obs, (float*)d_next_obs,
                                             h->diag, s);
    } else if (dtype == MJMPC_F64) {
        e = mjmpc::launch_arm_rollout<double>(h->model_f64, h->state, (long)P, H, h->nu, d_mean,
                                              (const double*)d_noise, (double*)d_costs, (double*)d_actions,
                                              (double*)d_obs, (double*)d_next_obs, h->diag, s);
    } else {
        return fail(MJMPC_E_BADARG, "unknown dtype %d", dtype);
    }
    if (e != hipSuccess) return hip_fail(e, "arm_rollout launch");
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

}  // extern "C"

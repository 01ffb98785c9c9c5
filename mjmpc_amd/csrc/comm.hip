// The one exchange of a sharded control iteration (reference: the gather of the workers' results,
// mjmpc/envs/vec_env/subproc_vec_env.py:161-186) issued FROM THE LIBRARY: an RCCL all-gather of this rank's float64 record on
// the iteration's stream.  A torch.distributed call does the same collective but is not a library call, so an iteration
// that contains one can only be replayed as a hipGraph (9-13 us between replays on the device); with the
// all-gather behind the C ABI the sharded iteration runs from the launch tape / as direct launches like the one-GPU loop.
//
// RCCL is bound at RUN TIME (dlopen / dlsym): the library stays loadable on a box without RCCL or without a GPU, and the
// process uses the copy of librccl that is already mapped (PyTorch ships its own) rather than a second one.
// For the same reason the handful of RCCL declarations this file needs are written out here instead of taken from
// <rccl/rccl.h>: the build does not depend on RCCL's development headers being installed, and cannot pick up a header that
// disagrees with the librccl the process has mapped.  They are the stable NCCL 2 C API (ncclUniqueId = 128 opaque bytes, the
// ncclDataType_t numbering); ncclGetVersion is checked at bind time (major version 2).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;                       // (0 = success; every other value is an error code)
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
}

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/mjmpc_amd.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    char why[256] = "";
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy the process already holds (RTLD_NOLOAD), else the system's
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "librccl.so not found (%s)", dlerror());
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        r.GetVersion = (decltype(r.GetVersion))dlsym(r.handle, "ncclGetVersion");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString || !r.GetVersion) {
            snprintf(r.why, sizeof(r.why), "librccl.so lacks one of the six entry points");
            return;
        }
        int v = 0;          // NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch since 2.9 (major * 1000 + ... before)
        if (r.GetVersion(&v) != ncclSuccess || !((v >= 20000 && v < 30000) || (v >= 2000 && v < 3000))) {
            snprintf(r.why, sizeof(r.why), "librccl.so reports version code %d: the declarations in comm.hip are NCCL 2's", v);
            r.GetUniqueId = nullptr;
        }
    });
    return r;
}

}  // namespace

namespace mjmpc {
int set_error(int code, const char* what, const char* detail);      // capi.hip: the message mjmpc_last_error() returns
}
static int comm_fail(int code, const char* what, const char* detail) { return mjmpc::set_error(code, what, detail); }

struct mjmpc_comm_s {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
};

extern "C" {

int mjmpc_comm_unique_id(void* id_out) {
    if (!id_out) return comm_fail(MJMPC_E_BADARG, "mjmpc_comm_unique_id", "null argument");
    Rccl& r = rccl();
    if (r.why[0]) return comm_fail(MJMPC_E_NOGPU, "RCCL", r.why);
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return comm_fail(1000 + (int)e, "ncclGetUniqueId", r.GetErrorString(e));
    static_assert(sizeof(id) == MJMPC_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int mjmpc_comm_create(const void* id_bytes, int world_size, int rank, int device, mjmpc_comm_t* out) {
    if (!id_bytes || !out || world_size < 1 || rank < 0 || rank >= world_size)
        return comm_fail(MJMPC_E_BADARG, "mjmpc_comm_create", "bad argument");
    Rccl& r = rccl();
    if (r.why[0]) return comm_fail(MJMPC_E_NOGPU, "RCCL", r.why);
    if (hipSetDevice(device) != hipSuccess) return comm_fail(MJMPC_E_NOGPU, "mjmpc_comm_create", "hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    mjmpc_comm_s* c = new mjmpc_comm_s();
    c->world = world_size;
    c->rank = rank;
    c->device = device;
    const ncclResult_t e = r.CommInitRank(&c->comm, world_size, id, rank);
    if (e != ncclSuccess) {
        delete c;
        return comm_fail(1000 + (int)e, "ncclCommInitRank", r.GetErrorString(e));
    }
    *out = c;
    return 0;
}

int mjmpc_comm_all_gather_f64(mjmpc_comm_t c, const double* d_send, double* d_recv, int64_t count, void* stream) {
    if (!c || !d_send || !d_recv || count < 0) return comm_fail(MJMPC_E_BADARG, "mjmpc_comm_all_gather_f64", "bad argument");
    if (count == 0) return 0;
    Rccl& r = rccl();
    const ncclResult_t e = r.AllGather(d_send, d_recv, (size_t)count, ncclFloat64, c->comm, (hipStream_t)stream);
    if (e != ncclSuccess) return comm_fail(1000 + (int)e, "ncclAllGather", r.GetErrorString(e));
    return 0;
}

int mjmpc_comm_destroy(mjmpc_comm_t c) {
    if (!c) return 0;
    Rccl& r = rccl();
    if (c->comm && r.CommDestroy) {
        hipSetDevice(c->device);
        r.CommDestroy(c->comm);
    }
    delete c;
    return 0;
}

}  // extern "C"

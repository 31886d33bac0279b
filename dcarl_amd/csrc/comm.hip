// The one collective of the path: an all-gather of the per-state summaries over RCCL / xGMI.
// Thin calls into librccl.so.1, resolved with dlopen at first use: a process that already carries torch's copy of RCCL
// gets that same copy (the loader matches by soname), a plain C caller gets the ROCm one.  The communicator is the
// caller's opaque handle; this library keeps no state beyond the resolved function pointers.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "common.h"

namespace dcarl {

int comm_fail(int code, const char* fmt, ...);   // abi.hip: sets the thread-local message, returns code

namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { snprintf(r.why, sizeof(r.why), "dlopen(librccl.so.1): %s", dlerror()); return; }
#define DCARL_SYM(field, name)                                                            \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                        \
    if (!r.field) { snprintf(r.why, sizeof(r.why), "librccl.so.1 lacks %s", name); return; }
        DCARL_SYM(GetUniqueId, "ncclGetUniqueId");
        DCARL_SYM(CommInitRank, "ncclCommInitRank");
        DCARL_SYM(AllGather, "ncclAllGather");
        DCARL_SYM(CommDestroy, "ncclCommDestroy");
        DCARL_SYM(GetErrorString, "ncclGetErrorString");
#undef DCARL_SYM
        r.ok = true;
    });
    return r;
}
static_assert(sizeof(ncclUniqueId) == DCARL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
}  // namespace

int comm_unique_id(uint8_t* id) {
    Rccl& r = rccl();
    if (!r.ok) return comm_fail(DCARL_ECOMM, "%s", r.why);
    ncclUniqueId u;
    const ncclResult_t rc = r.GetUniqueId(&u);
    if (rc != ncclSuccess) return comm_fail(DCARL_ECOMM, "ncclGetUniqueId: %s", r.GetErrorString(rc));
    memcpy(id, &u, sizeof(u));
    return DCARL_OK;
}

int comm_init(int nranks, int rank, const uint8_t* id, void** comm) {
    Rccl& r = rccl();
    if (!r.ok) return comm_fail(DCARL_ECOMM, "%s", r.why);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const ncclResult_t rc = r.CommInitRank(&c, nranks, u, rank);
    if (rc != ncclSuccess) return comm_fail(DCARL_ECOMM, "ncclCommInitRank(%d of %d): %s", rank, nranks, r.GetErrorString(rc));
    *comm = c;
    return DCARL_OK;
}

int comm_allgather(void* comm, const void* send, void* recv, int64_t bytes, hipStream_t st) {
    Rccl& r = rccl();
    if (!r.ok) return comm_fail(DCARL_ECOMM, "%s", r.why);
    const ncclResult_t rc = r.AllGather(send, recv, (size_t)bytes, ncclInt8, static_cast<ncclComm_t>(comm), st);
    if (rc != ncclSuccess) return comm_fail(DCARL_ECOMM, "ncclAllGather(%lld B): %s", (long long)bytes, r.GetErrorString(rc));
    return DCARL_OK;
}

int comm_destroy(void* comm) {
    Rccl& r = rccl();
    if (!r.ok) return comm_fail(DCARL_ECOMM, "%s", r.why);
    const ncclResult_t rc = r.CommDestroy(static_cast<ncclComm_t>(comm));
    if (rc != ncclSuccess) return comm_fail(DCARL_ECOMM, "ncclCommDestroy: %s", r.GetErrorString(rc));
    return DCARL_OK;
}

}  // namespace dcarl

// RCCL behind the C ABI: the gradient exchange of patch-level data parallelism, one process per GPU over xGMI.
//
// Replaces the reference's only multi-GPU mechanism, single-process nn.DataParallel (ctunet/pytorch/Model.py:481-487:
// replicate / scatter / gather / reduce-add inside one process, which cannot even split the batch of 1 every example
// ini uses) by ONE collective per gradient bucket: ncclAllReduce(avg, fp32) on a caller-provided stream.
//
// librccl is bound at run time (dlopen by soname), so libctunet_hip.so has no link-time dependency on it: the library
// loads on a box without RCCL, and in a process where PyTorch-ROCm has already mapped its own librccl.so.1 the same copy
// is used (one RCCL per process).  Nothing here synchronises the stream; the communicator is the caller's to destroy.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* err = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.err = "librccl.so.1 not found (dlopen)"; return; }
#define CTU_SYM(field, sym)                                                      \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));     \
        if (!r.field) { r.err = "librccl: missing symbol " sym; return; }
        CTU_SYM(GetUniqueId, "ncclGetUniqueId")
        CTU_SYM(CommInitRank, "ncclCommInitRank")
        CTU_SYM(CommDestroy, "ncclCommDestroy")
        CTU_SYM(AllReduce, "ncclAllReduce")
        CTU_SYM(GroupStart, "ncclGroupStart")
        CTU_SYM(GroupEnd, "ncclGroupEnd")
        CTU_SYM(GetErrorString, "ncclGetErrorString")
#undef CTU_SYM
    });
    return r;
}

#define CTU_RCCL(call, what)                                                            \
    do {                                                                                \
        ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess) {                                                        \
            ctu_set_error("%s: %s", what, rccl().GetErrorString(r_));                   \
            return CTU_ELAUNCH;                                                         \
        }                                                                               \
    } while (0)

}  // namespace

static_assert(sizeof(ncclUniqueId) == CTU_COMM_ID_BYTES, "ctunet_hip.h: CTU_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int ctu_comm_available(void) { return rccl().err == nullptr ? 1 : 0; }

extern "C" int ctu_comm_unique_id(void* id_host) {
    CTU_REQUIRE(id_host, "comm_unique_id: null pointer");
    CTU_REQUIRE(rccl().err == nullptr, "comm_unique_id: %s", rccl().err);
    CTU_RCCL(rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_host)), "ncclGetUniqueId");
    return CTU_OK;
}

extern "C" int ctu_comm_init(void** comm, int world, int rank, const void* id_host) {
    CTU_REQUIRE(comm && id_host, "comm_init: null pointer");
    CTU_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    CTU_REQUIRE(rccl().err == nullptr, "comm_init: %s", rccl().err);
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t c = nullptr;
    CTU_RCCL(rccl().CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    *comm = c;
    return CTU_OK;
}

extern "C" int ctu_comm_destroy(void* comm) {
    if (!comm) return CTU_OK;
    CTU_REQUIRE(rccl().err == nullptr, "comm_destroy: %s", rccl().err);
    CTU_RCCL(rccl().CommDestroy(reinterpret_cast<ncclComm_t>(comm)), "ncclCommDestroy");
    return CTU_OK;
}

extern "C" int ctu_comm_allreduce_f32(void* comm, const float* send, float* recv, size_t count, int average, void* stream) {
    CTU_REQUIRE(comm && send && recv, "comm_allreduce_f32: null pointer");
    CTU_REQUIRE(rccl().err == nullptr, "comm_allreduce_f32: %s", rccl().err);
    if (count == 0) return CTU_OK;
    CTU_RCCL(rccl().AllReduce(send, recv, count, ncclFloat32, average ? ncclAvg : ncclSum, reinterpret_cast<ncclComm_t>(comm),
                              (hipStream_t)stream),
             "ncclAllReduce");
    return CTU_OK;
}

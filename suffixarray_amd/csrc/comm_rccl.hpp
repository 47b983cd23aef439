// comm_rccl.hpp -- multi-GPU lifecycle in the C ABI (no counterpart in the reference, which has
// no distributed code at all): RCCL communicator init / teardown, index replication and the gather of the 8-byte range
// pairs, without PyTorch.  One process per GPU; the caller ships the 128-byte unique id from rank 0 to the other ranks
// by whatever it has (MPI, a file, a socket) -- exactly RCCL's own bootstrap contract.
//
// RCCL is loaded at run time (dlopen): libsa_hip.so carries no link-time dependency on it, a single-GPU user never
// touches it, and a process that already holds a copy (PyTorch ships its own librccl.so) gets THAT copy instead of a
// second one.  xGMI is point to point, so the replication is one large broadcast per buffer (RCCL pipelines it over
// the links), straight out of the builder's buffers into buffers the replica has reserved (sa_hip_index_replica_*).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace sa {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

inline RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) { a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (a.handle) break; }   // a copy the process already holds
        if (!a.handle) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.handle) break; }
        if (!a.handle) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(a.handle, "ncclBroadcast"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.Broadcast && a.AllGather && a.AllReduce && a.GetErrorString;
        return a;
    }();
    return api;
}

#define SA_RCCL_CHECK(expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess) return ::sa::fail(SA_HIP_EHIP, #expr, ::sa::rccl_api().GetErrorString(_r));         \
    } while (0)

}  // namespace sa

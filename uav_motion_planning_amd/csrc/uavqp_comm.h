// uavqp_comm.h -- multi-GPU entry points of the C ABI (include/uavqp.h): contiguous shards of the trajectory batch, one
// process (or thread) per GPU, no data-path collective in the solve, ONE exchange step: the all-gather of the solved
// coefficient shards over xGMI through RCCL.  The reference has no multi-device code at all (its only
// remark on parallelism is test_minimum_jerk.cpp:73-74); this is the north star's "RCCL all-gather of solved coefficients".
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME both /opt/rocm and the PyTorch wheel ship): a process that
// already holds a RCCL -- e.g. torch.distributed's -- shares it, and libuavqp.so carries no link-time dependency for the
// single-GPU user.  The communicator belongs to the ctx and runs on the ctx stream, i.e. ordered behind the solve.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

namespace uavqp {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // what RCCL itself says the communicator is (uavqp_comm_info)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;

    // one thread per GPU is the documented use: the first callers may race here, so the binding happens exactly once
    std::once_flag once;
    bool loaded = false;
    bool load() {
        std::call_once(once, [this] { loaded = load_once(); });
        return loaded;
    }
    bool load_once() {
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            error = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed");
            return false;
        }
        bool ok = true;
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(handle, name);
            if (!p) { ok = false; error = std::string("RCCL symbol missing: ") + name; }
            return p;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        CommUserRank = (decltype(CommUserRank))sym("ncclCommUserRank");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(handle); handle = nullptr; }
        return ok;
    }
};

static RcclApi& rccl() {
    static RcclApi api;
    return api;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

}  // namespace uavqp

// TEST-ONLY stand-in for librccl.so.1 (tests/test_gpu_multirank_fake_rccl.py): the eleven symbols libuavqp.so binds at run time
// (csrc/uavqp_comm.h), implemented for ranks that are THREADS OF ONE PROCESS sharing one GPU -- the one configuration a single-GPU box
// offers (real RCCL refuses two ranks on one device).  Purpose: execute the world > 1 branches of the C ABI (the grouped
// ncclSend / ncclRecv all-gather-v of allgather_shards, in-place aliasing, zero-sized shards) before the driver's 8-GPU run does.
// NOT a communication library: every collective synchronises the calling rank's stream, meets the other ranks at a barrier, copies
// device-to-device with hipMemcpy and meets them again.  Semantics kept from RCCL: ops are matched by (source, destination) in
// posting order inside a group, counts of a matched pair must agree (else ncclInvalidArgument), a group is executed at ncclGroupEnd.
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct FakeComm;
typedef FakeComm* ncclComm_t;
}

namespace {
size_t dt_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
struct Op { bool send; const void* src; void* dst; size_t bytes; int peer; };
struct World {
    int world = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<std::vector<Op>> posted;   // [rank]: the ops of the group being executed
    bool failed = false;
    void barrier() {
        std::unique_lock<std::mutex> l(m);
        const unsigned long long g = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(l, [&] { return generation != g; });
    }
};
std::mutex g_m;
std::map<std::string, World*> g_worlds;
unsigned long long g_next_id = 1;
}  // namespace

struct FakeComm {
    World* w;
    int rank;
    hipStream_t stream = nullptr;
    std::vector<Op> pending;
};
static thread_local int t_group_depth = 0;
static thread_local std::vector<FakeComm*> t_group_comms;

static ncclResult_t execute(FakeComm* c) {
    World* w = c->w;
    if (c->stream && hipStreamSynchronize(c->stream) != hipSuccess) return ncclUnhandledCudaError;   // this rank's buffers are final
    { std::lock_guard<std::mutex> l(w->m); w->posted[c->rank] = c->pending; }
    w->barrier();                                  // every rank has posted
    ncclResult_t rc = ncclSuccess;
    std::vector<size_t> cursor(w->world, 0);       // per peer: how many of its sends to me have been consumed
    for (const Op& r : c->pending) {
        if (r.send) continue;
        const std::vector<Op>& theirs = w->posted[r.peer];
        size_t k = cursor[r.peer], seen = 0, idx = theirs.size();
        for (size_t i = 0; i < theirs.size(); ++i)
            if (theirs[i].send && theirs[i].peer == c->rank) { if (seen == k) { idx = i; break; } ++seen; }
        cursor[r.peer] = k + 1;
        if (idx == theirs.size() || theirs[idx].bytes != r.bytes) { rc = ncclInvalidArgument; continue; }
        if (r.bytes && hipMemcpy(r.dst, theirs[idx].src, r.bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    if (hipDeviceSynchronize() != hipSuccess) rc = ncclUnhandledCudaError;
    w->barrier();                                  // nobody reuses a send buffer before every receiver has copied
    c->pending.clear();
    return rc;
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> l(g_m);
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "fake-rccl-%llu", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    World* w;
    {
        std::lock_guard<std::mutex> l(g_m);
        World*& slot = g_worlds[std::string(id.internal)];
        if (!slot) { slot = new World(); slot->world = nranks; slot->posted.resize(nranks); }
        w = slot;
        if (w->world != nranks) return ncclInvalidArgument;
        ++w->joined;
    }
    FakeComm* c = new FakeComm();
    c->w = w;
    c->rank = rank;
    *comm = c;
    w->barrier();   // like the real thing: returns once every rank has joined
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->w->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

static ncclResult_t post(FakeComm* c, const Op& op, hipStream_t s) {
    c->stream = s;
    c->pending.push_back(op);
    if (t_group_depth > 0) {
        bool known = false;
        for (FakeComm* k : t_group_comms) known = known || k == c;
        if (!known) t_group_comms.push_back(c);
        return ncclSuccess;
    }
    return execute(c);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
    if (!comm || peer < 0 || peer >= comm->w->world) return ncclInvalidArgument;
    return post(comm, Op{true, buf, nullptr, count * dt_size(dt), peer}, s);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
    if (!comm || peer < 0 || peer >= comm->w->world) return ncclInvalidArgument;
    return post(comm, Op{false, nullptr, buf, count * dt_size(dt), peer}, s);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t s) {
    if (!comm) return ncclInvalidArgument;
    const size_t bytes = count * dt_size(dt);
    const bool grouped = t_group_depth > 0;
    if (!grouped) ++t_group_depth;
    for (int g = 0; g < comm->w->world; ++g) {      // every rank sends its block to every rank (itself included) and receives all blocks
        post(comm, Op{true, send, nullptr, bytes, g}, s);
        post(comm, Op{false, nullptr, (char*)recv + (size_t)g * bytes, bytes, g}, s);
    }
    if (grouped) return ncclSuccess;
    --t_group_depth;
    t_group_comms.clear();
    return execute(comm);
}

ncclResult_t ncclGroupStart() {
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (FakeComm* c : t_group_comms) { const ncclResult_t r = execute(c); if (r != ncclSuccess) rc = r; }
    t_group_comms.clear();
    return rc;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) { case ncclSuccess: return "no error"; case ncclInvalidArgument: return "invalid argument (fake rccl)"; case ncclInvalidUsage: return "invalid usage (fake rccl)";
                 case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)"; default: return "error (fake rccl)"; }
}

}  // extern "C"

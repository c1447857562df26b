// comm.hpp -- RCCL through dlopen: libbella_hip.so has no link-time dependency on librccl; the communicator entry points of
// include/bella_hip.h resolve it on first use (the copy a host program such as PyTorch already loaded, else /opt/rocm's).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace bella {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional: tears a stalled communicator down (comm_sync's deadline)
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return h && GetUniqueId && CommInitRank && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd; }
};

inline const Rccl& rccl() {
    static const Rccl api = [] {
        Rccl r;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {                          // already in the process?
            r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (r.h) break;
        }
        for (int i = 0; !r.h && i < 3; ++i) r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) return r;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.h, "ncclCommAbort");
        r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
        r.Send = (decltype(r.Send))dlsym(r.h, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.h, "ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        return r;
    }();
    return api;
}

// ---- in-process transport with RCCL's call shapes -----------------------------------------------------------------------------
// The "ranks" are contexts of ONE process, each driven by its own host thread (the shim's numGPU contexts; tests that exercise the
// N > 1 send/recv offset logic of the collective entry points on a single GPU).  A communicator is a seat in a group keyed by the
// 128-byte id; all-gather and grouped send/recv rendezvous on the host and move the bytes with device-to-device copies on the
// caller's stream.  Same semantics as the RCCL calls the library makes: grouped point-to-point operations between a pair of ranks
// match in program order; every call returns when the caller's buffers may be reused.
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0;
    int seated = 0;
    uint64_t bar_gen = 0;
    int bar_count = 0;
    std::vector<const void*> pub;                               // all-gather: every rank's send buffer
    struct Msg { const void* p; size_t bytes; };
    std::vector<std::deque<Msg>> mail;                          // [src * nranks + dst]
    std::vector<uint64_t> taken;                                // [src * nranks + dst] messages the receiver has copied out
    std::vector<uint64_t> sent;                                 // [src * nranks + dst] messages the sender has posted
};
struct LoopComm {
    std::shared_ptr<LoopGroup> g;
    int rank = 0;
};
struct LoopOp { bool send; void* p; size_t bytes; int peer; LoopComm* c; hipStream_t st; };

inline std::mutex& loop_registry_mu() { static std::mutex m; return m; }
inline std::map<std::string, std::shared_ptr<LoopGroup>>& loop_registry() { static std::map<std::string, std::shared_ptr<LoopGroup>> r; return r; }
inline int& loop_group_depth() { static thread_local int d = 0; return d; }
inline std::vector<LoopOp>& loop_pending() { static thread_local std::vector<LoopOp> v; return v; }

inline size_t loop_type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclUint8: case ncclInt8: return 1;
        case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
        case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
        default: return 2;
    }
}
inline void loop_barrier(LoopGroup& g, std::unique_lock<std::mutex>& lk) {
    const uint64_t gen = g.bar_gen;
    if (++g.bar_count == g.nranks) { g.bar_count = 0; ++g.bar_gen; g.cv.notify_all(); }
    else g.cv.wait(lk, [&] { return g.bar_gen != gen; });
}
inline ncclResult_t loop_GetUniqueId(ncclUniqueId* id) {
    static std::atomic<uint64_t> next{1};
    std::memset(id, 0, sizeof(*id));
    const uint64_t v = next.fetch_add(1);
    std::memcpy(id->internal, "bella-loop", 10);
    std::memcpy(id->internal + 16, &v, 8);
    const uint64_t pid = (uint64_t)getpid();
    std::memcpy(id->internal + 24, &pid, 8);
    return ncclSuccess;
}
inline ncclResult_t loop_CommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    std::shared_ptr<LoopGroup> g;
    {
        std::lock_guard<std::mutex> lk(loop_registry_mu());
        auto& slot = loop_registry()[std::string(id.internal, sizeof(id.internal))];
        if (!slot) {
            slot = std::make_shared<LoopGroup>();
            slot->nranks = nranks;
            slot->pub.assign((size_t)nranks, nullptr);
            slot->mail.resize((size_t)nranks * nranks);
            slot->taken.assign((size_t)nranks * nranks, 0);
            slot->sent.assign((size_t)nranks * nranks, 0);
        }
        g = slot;
    }
    if (g->nranks != nranks) return ncclInvalidArgument;
    LoopComm* c = new LoopComm();
    c->g = g;
    c->rank = rank;
    {   // like ncclCommInitRank: returns when every rank has joined
        std::unique_lock<std::mutex> lk(g->mu);
        ++g->seated;
        g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->seated >= g->nranks; });
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}
inline ncclResult_t loop_CommDestroy(ncclComm_t comm) {
    LoopComm* c = (LoopComm*)comm;
    {
        std::lock_guard<std::mutex> lk(loop_registry_mu());
        for (auto it = loop_registry().begin(); it != loop_registry().end(); ++it)
            if (it->second == c->g && it->second.use_count() <= 2) { loop_registry().erase(it); break; }
    }
    delete c;
    return ncclSuccess;
}
inline ncclResult_t loop_AllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
    LoopComm* c = (LoopComm*)comm;
    LoopGroup& g = *c->g;
    const size_t bytes = count * loop_type_bytes(t);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;   // my contribution is complete
    {
        std::unique_lock<std::mutex> lk(g.mu);
        g.pub[(size_t)c->rank] = send;
        loop_barrier(g, lk);
    }
    for (int r = 0; r < g.nranks; ++r)
        if (bytes && hipMemcpyAsync((char*)recv + (size_t)r * bytes, g.pub[(size_t)r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(g.mu);
    loop_barrier(g, lk);                                         // nobody's send buffer changes before everybody has read it
    return ncclSuccess;
}
inline ncclResult_t loop_run(std::vector<LoopOp>& ops) {
    if (ops.empty()) return ncclSuccess;
    for (auto& o : ops)
        if (o.send && hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    for (auto& o : ops) {
        if (!o.send) continue;
        LoopGroup& g = *o.c->g;
        std::lock_guard<std::mutex> lk(g.mu);
        const size_t x = (size_t)o.c->rank * g.nranks + o.peer;
        g.mail[x].push_back({o.p, o.bytes});
        ++g.sent[x];
        g.cv.notify_all();
    }
    ncclResult_t rc = ncclSuccess;
    for (auto& o : ops) {
        if (o.send) continue;
        LoopGroup& g = *o.c->g;
        LoopGroup::Msg m;
        {
            std::unique_lock<std::mutex> lk(g.mu);
            const size_t x = (size_t)o.peer * g.nranks + o.c->rank;
            g.cv.wait(lk, [&] { return !g.mail[x].empty(); });
            m = g.mail[x].front();
            g.mail[x].pop_front();
        }
        if (m.bytes != o.bytes) rc = ncclInvalidArgument;          // sizes of a matched pair must agree
        else if (m.bytes && hipMemcpyAsync(o.p, m.p, m.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    for (auto& o : ops)
        if (!o.send && hipStreamSynchronize(o.st) != hipSuccess) rc = ncclUnhandledCudaError;
    for (auto& o : ops) {
        if (o.send) continue;
        LoopGroup& g = *o.c->g;
        std::lock_guard<std::mutex> lk(g.mu);
        ++g.taken[(size_t)o.peer * g.nranks + o.c->rank];
        g.cv.notify_all();
    }
    for (auto& o : ops) {                                        // my send buffers are free once every receiver has copied them out
        if (!o.send) continue;
        LoopGroup& g = *o.c->g;
        std::unique_lock<std::mutex> lk(g.mu);
        const size_t x = (size_t)o.c->rank * g.nranks + o.peer;
        g.cv.wait(lk, [&] { return g.taken[x] >= g.sent[x]; });
    }
    ops.clear();
    return rc;
}
inline ncclResult_t loop_GroupStart() { ++loop_group_depth(); return ncclSuccess; }
inline ncclResult_t loop_GroupEnd() {
    if (loop_group_depth() > 0 && --loop_group_depth() == 0) return loop_run(loop_pending());
    return ncclSuccess;
}
inline ncclResult_t loop_Send(const void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    loop_pending().push_back({true, (void*)p, count * loop_type_bytes(t), peer, (LoopComm*)comm, st});
    return loop_group_depth() ? ncclSuccess : loop_run(loop_pending());
}
inline ncclResult_t loop_Recv(void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    loop_pending().push_back({false, p, count * loop_type_bytes(t), peer, (LoopComm*)comm, st});
    return loop_group_depth() ? ncclSuccess : loop_run(loop_pending());
}
inline const char* loop_GetErrorString(ncclResult_t) { return "in-process transport error"; }

inline const Rccl& loopback() {
    static const Rccl api = [] {
        Rccl r;
        r.h = (void*)1;
        r.GetUniqueId = loop_GetUniqueId; r.CommInitRank = loop_CommInitRank; r.CommDestroy = loop_CommDestroy; r.AllGather = loop_AllGather;
        r.Send = loop_Send; r.Recv = loop_Recv; r.GroupStart = loop_GroupStart; r.GroupEnd = loop_GroupEnd; r.GetErrorString = loop_GetErrorString;
        return r;
    }();
    return api;
}

}  // namespace bella

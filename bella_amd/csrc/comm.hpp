// comm.hpp -- RCCL through dlopen: libbella_hip.so has no link-time dependency on librccl; the communicator entry points of
// include/bella_hip.h resolve it on first use (the copy a host program such as PyTorch already loaded, else /opt/rocm's).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
// The "ranks" are contexts of ONE process, each driven by its own host thread (the host programs' -g N contexts; tests that exercise
// the N > 1 send/recv offset logic of the collective entry points on a single GPU).  A communicator is a seat in a group keyed by the
// 128-byte id; all-gather and grouped send/recv rendezvous on the host and move the bytes with device-to-device copies on the
// caller's stream.  Same semantics as the RCCL calls the library makes: grouped point-to-point operations between a pair of ranks
// match in program order; every call returns when the caller's buffers may be reused.
// Seats on DIFFERENT devices (a real -g N run): every seat records its device at init; once all have joined, a seat enables peer
// access to every other device it can reach (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess: xGMI between the GPUs of a node) and
// the copies between two devices are hipMemcpyPeerAsync -- which the runtime stages through the host where there is no peer path, so
// a pair without one is slow, not wrong (said once on stderr).
// Every rendezvous wait has a deadline (BELLA_HIP_COMM_TIMEOUT_S, default 120 s, as the stream waits of bella_hip.hip: comm_sync): a
// seat that waits longer marks the group broken and leaves with an error; every other seat's waits end at once with the same error.
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0;
    int seated = 0;
    bool broken = false;                                        // a seat gave up waiting (or failed): every wait of the group ends
    uint64_t bar_gen = 0;
    int bar_count = 0;
    std::vector<const void*> pub;                               // all-gather: every rank's send buffer
    std::vector<int> dev;                                       // every rank's device
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
inline double loop_timeout_s() {
    const char* e = getenv("BELLA_HIP_COMM_TIMEOUT_S");          // (read at every wait: a host program may set it per phase)
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
}
// cv.wait with the group's deadline: false = the deadline passed or the group is broken (the group is marked broken either way)
template <class Pred>
inline bool loop_wait(LoopGroup& g, std::unique_lock<std::mutex>& lk, Pred pred) {
    const bool ok = g.cv.wait_for(lk, std::chrono::duration<double>(loop_timeout_s()), [&] { return g.broken || pred(); });
    if (!ok || (g.broken && !pred())) { g.broken = true; g.cv.notify_all(); return false; }
    return true;
}

inline size_t loop_type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclUint8: case ncclInt8: return 1;
        case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
        case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
        default: return 2;
    }
}
inline bool loop_barrier(LoopGroup& g, std::unique_lock<std::mutex>& lk) {
    const uint64_t gen = g.bar_gen;
    if (++g.bar_count == g.nranks) { g.bar_count = 0; ++g.bar_gen; g.cv.notify_all(); return !g.broken; }
    return loop_wait(g, lk, [&] { return g.bar_gen != gen; });
}
// bytes from a buffer on device sdev to one on device ddev, on the receiver's stream
inline hipError_t loop_copy(void* dst, int ddev, const void* src, int sdev, size_t bytes, hipStream_t st) {
    if (ddev == sdev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, st);
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
            slot->dev.assign((size_t)nranks, -1);
            slot->mail.resize((size_t)nranks * nranks);
            slot->taken.assign((size_t)nranks * nranks, 0);
            slot->sent.assign((size_t)nranks * nranks, 0);
        }
        g = slot;
    }
    if (g->nranks != nranks) return ncclInvalidArgument;
    int mydev = 0;
    if (hipGetDevice(&mydev) != hipSuccess) return ncclUnhandledCudaError;
    LoopComm* c = new LoopComm();
    c->g = g;
    c->rank = rank;
    {   // like ncclCommInitRank: returns when every rank has joined
        std::unique_lock<std::mutex> lk(g->mu);
        g->dev[(size_t)rank] = mydev;
        ++g->seated;
        g->cv.notify_all();
        if (!loop_wait(*g, lk, [&] { return g->seated >= g->nranks; })) { delete c; return ncclSystemError; }
    }
    // peer access from this seat's device to every other device of the group (enabled once per device pair and process)
    for (int r = 0; r < nranks; ++r) {
        const int pd = g->dev[(size_t)r];
        if (pd == mydev || pd < 0) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, mydev, pd) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(pd, 0);
            if (e != hipSuccess) (void)hipGetLastError();        // (hipErrorPeerAccessAlreadyEnabled: another seat of this device was first)
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
        }
        if (!can) {
            static std::once_flag once;
            std::call_once(once, [&] { fprintf(stderr, "bella_hip: no peer access between devices %d and %d: the in-process transport stages their copies through the host\n", mydev, pd); });
        }
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
// a seat that cannot go on (a failed copy, a deadline): nobody waits for it
inline ncclResult_t loop_break(LoopGroup& g, ncclResult_t rc) {
    std::lock_guard<std::mutex> lk(g.mu);
    g.broken = true;
    g.cv.notify_all();
    return rc;
}
inline ncclResult_t loop_CommAbort(ncclComm_t comm) {
    LoopComm* c = (LoopComm*)comm;
    (void)loop_break(*c->g, ncclSuccess);
    return loop_CommDestroy(comm);
}
inline ncclResult_t loop_AllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
    LoopComm* c = (LoopComm*)comm;
    LoopGroup& g = *c->g;
    const size_t bytes = count * loop_type_bytes(t);
    if (hipStreamSynchronize(st) != hipSuccess) return loop_break(g, ncclUnhandledCudaError);   // my contribution is complete
    {
        std::unique_lock<std::mutex> lk(g.mu);
        g.pub[(size_t)c->rank] = send;
        if (!loop_barrier(g, lk)) return ncclSystemError;
    }
    const int mydev = g.dev[(size_t)c->rank];
    for (int r = 0; r < g.nranks; ++r)
        if (bytes && loop_copy((char*)recv + (size_t)r * bytes, mydev, g.pub[(size_t)r], g.dev[(size_t)r], bytes, st) != hipSuccess) return loop_break(g, ncclUnhandledCudaError);
    if (hipStreamSynchronize(st) != hipSuccess) return loop_break(g, ncclUnhandledCudaError);
    std::unique_lock<std::mutex> lk(g.mu);
    if (!loop_barrier(g, lk)) return ncclSystemError;           // nobody's send buffer changes before everybody has read it
    return ncclSuccess;
}
inline ncclResult_t loop_run(std::vector<LoopOp>& ops) {
    if (ops.empty()) return ncclSuccess;
    struct Clear { std::vector<LoopOp>& v; ~Clear() { v.clear(); } } clear{ops};
    for (auto& o : ops)
        if (o.send && hipStreamSynchronize(o.st) != hipSuccess) return loop_break(*o.c->g, ncclUnhandledCudaError);
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
            if (!loop_wait(g, lk, [&] { return !g.mail[x].empty(); })) return ncclSystemError;
            m = g.mail[x].front();
            g.mail[x].pop_front();
        }
        if (m.bytes != o.bytes) rc = ncclInvalidArgument;          // sizes of a matched pair must agree
        else if (m.bytes && loop_copy(o.p, g.dev[(size_t)o.c->rank], m.p, g.dev[(size_t)o.peer], m.bytes, o.st) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    for (auto& o : ops)
        if (!o.send && hipStreamSynchronize(o.st) != hipSuccess) rc = ncclUnhandledCudaError;
    if (rc != ncclSuccess) return loop_break(*ops[0].c->g, rc);
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
        if (!loop_wait(g, lk, [&] { return g.taken[x] >= g.sent[x]; })) return ncclSystemError;
    }
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
inline const char* loop_GetErrorString(ncclResult_t r) { return r == ncclSystemError ? "in-process transport: a peer did not arrive in time or left the group" : "in-process transport error"; }

inline const Rccl& loopback() {
    static const Rccl api = [] {
        Rccl r;
        r.h = (void*)1;
        r.GetUniqueId = loop_GetUniqueId; r.CommInitRank = loop_CommInitRank; r.CommDestroy = loop_CommDestroy; r.AllGather = loop_AllGather;
        r.Send = loop_Send; r.Recv = loop_Recv; r.GroupStart = loop_GroupStart; r.GroupEnd = loop_GroupEnd; r.GetErrorString = loop_GetErrorString;
        r.CommAbort = loop_CommAbort;
        return r;
    }();
    return api;
}

}  // namespace bella

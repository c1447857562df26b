// comm.hpp -- RCCL through dlopen: libbella_hip.so has no link-time dependency on librccl; the communicator entry points of
// include/bella_hip.h resolve it on first use (the copy a host program such as PyTorch already loaded, else /opt/rocm's).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only

namespace bella {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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

}  // namespace bella

// ise_comm.hip -- the ONE collective of the sharded search, issued straight on the caller's HIP
// stream: an RCCL all-gather of the per-shard packed candidates (nq x k uint64 per rank, latency
// bound).  The reference has no collective (single in-RAM IndexFlat,
// backend/utils.py:327); this is the exchange step of the row-sharded index north_star asks for.
//
// Going through torch.distributed costs ~47 us of host time per collective (measured round 1,
// scripts/shard_host_probe.py) -- more than a 125k-row shard scan takes on the GPU.  Here the
// communicator belongs to the library: rank 0 draws an id (ise_comm_unique_id), the host side
// passes it to the other ranks by whatever channel it has (sharded.py: one torch.distributed
// broadcast at start-up), every rank calls ise_comm_create, and from then on a bucket's exchange is
// one ncclAllGather enqueued on the bucket's stream between its scans and its merge.
//
// librccl is resolved at run time (dlopen): the library loads, and every single-GPU entry point
// works, on a box without RCCL; PyTorch-ROCm's bundled copy is reused when it is already loaded.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "ise_common.hpp"

extern int ise_fail_(int code, const std::string& msg);  // ise_knn.hip: sets the thread-local message

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)  // a copy the process already holds (PyTorch-ROCm's) first
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) {
            api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
            return;
        }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString)
            api.err = "librccl lacks an expected symbol";
    });
    return &api;
}

int rccl_fail(const char* what, ncclResult_t r) {
    RcclApi* a = rccl();
    return ise_fail_(ISE_E_HIP, std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(r) : "?"));
}
}  // namespace

struct ise_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
};

extern "C" int ise_comm_precheck(int device) {
    RcclApi* a = rccl();
    if (!a->err.empty()) return ise_fail_(ISE_E_NODEVICE, a->err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return ise_fail_(ISE_E_NODEVICE, "device out of range (the collective needs a GPU)");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return ise_fail_(ISE_E_HIP, "hipSetDevice failed");
    void* probe = nullptr;  // the device really is usable from this process
    const hipError_t e = hipMalloc(&probe, 256);
    if (e == hipSuccess) (void)hipFree(probe);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return ise_fail_(ISE_E_HIP, std::string("device probe failed: ") + hipGetErrorString(e));
    return ISE_OK;
}

extern "C" int ise_comm_unique_id(void* id128) {
    if (!id128) return ise_fail_(ISE_E_INVALID, "id128 is NULL");
    RcclApi* a = rccl();
    if (!a->err.empty()) return ise_fail_(ISE_E_NODEVICE, a->err);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = a->GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id128, &id, sizeof(id));
    return ISE_OK;
}

extern "C" int ise_comm_create(ise_comm_t** out, const void* id128, int world, int rank, int device) {
    if (!out || !id128) return ise_fail_(ISE_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return ise_fail_(ISE_E_INVALID, "bad world / rank");
    RcclApi* a = rccl();
    if (!a->err.empty()) return ise_fail_(ISE_E_NODEVICE, a->err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return ise_fail_(ISE_E_NODEVICE, "device out of range (the collective needs a GPU)");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return ise_fail_(ISE_E_HIP, "hipSetDevice failed");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = a->CommInitRank(&c, world, id, rank);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    ise_comm* h = new (std::nothrow) ise_comm();
    if (!h) {
        (void)a->CommDestroy(c);
        return ise_fail_(ISE_E_NOMEM, "host allocation failed");
    }
    h->comm = c; h->world = world; h->rank = rank; h->device = device;
    *out = h;
    return ISE_OK;
}

extern "C" int ise_comm_allgather_keys(ise_comm_t* c, const uint64_t* send_dev, uint64_t* recv_dev, int64_t count,
                                       void* stream) {
    if (!c || !c->comm) return ise_fail_(ISE_E_INVALID, "communicator is NULL");
    if (count < 0 || (count > 0 && (!send_dev || !recv_dev))) return ise_fail_(ISE_E_INVALID, "bad buffer argument");
    if (count == 0) return ISE_OK;
    ncclResult_t r = rccl()->AllGather(send_dev, recv_dev, (size_t)count, ncclUint64, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    return ISE_OK;
}

extern "C" int ise_comm_destroy(ise_comm_t* c) {
    if (!c) return ISE_OK;
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    delete c;
    return ISE_OK;
}

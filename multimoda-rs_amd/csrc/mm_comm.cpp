// mm_comm.cpp -- the RCCL communicator behind the C ABI (include/mm_hausdorff.h, "multi-GPU").
//
// The per-shard best scores of a sharded search are all-reduced over the ranks of one node (one process
// per GPU, xGMI).  A host that is not Python -- the reference's Rust host binds this library through
// extern "C" -- gets the collective from the library itself: mm_comm_unique_id on rank 0, the 128 bytes
// handed to every rank by whatever the host uses to start its ranks, mm_comm_init_rank on each, then
// mm_within_plan_search_sharded / _run_sharded issue ncclAllReduce(MIN) on the engine's stream,
// stream-ordered behind the export kernels.
//
// RCCL is loaded at run time (dlopen "librccl.so.1"), not linked: a single-GPU host does not need it, and a
// process that already holds a copy (torch ships one under the same soname) shares that one instead of
// mapping a second.  MM_RCCL_LIB overrides the path.
#include "mm_engine.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace mm {

struct RcclApi {
    void* handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId)   GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)  CommInitRank = nullptr;
    decltype(&ncclCommDestroy)   CommDestroy = nullptr;
    decltype(&ncclAllReduce)     AllReduce = nullptr;
    decltype(&ncclBroadcast)     Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion)    GetVersion = nullptr;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static const RcclApi& rccl()
{
    std::call_once(g_rccl_once, [] {
        RcclApi& a = g_rccl;
        const char* env = std::getenv("MM_RCCL_LIB");
        // a library named by MM_RCCL_LIB is the only one tried: an override that silently fell back would hide a typo
        const bool named = env && *env;
        const char* names[] = {named ? env : "librccl.so.1", named ? nullptr : "librccl.so", named ? nullptr : "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
            a.error = dlerror();
        }
        if (!a.handle) { a.error = "RCCL not found (librccl.so.1; set MM_RCCL_LIB): " + a.error; return; }
        auto sym = [&](const char* s) { void* p = dlsym(a.handle, s); if (!p && a.error.empty()) a.error = std::string("RCCL symbol missing: ") + s; return p; };
        a.error.clear();
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        a.GetVersion = (decltype(a.GetVersion))sym("ncclGetVersion");
        if (!a.error.empty()) { dlclose(a.handle); a.handle = nullptr; }
    });
    return g_rccl;
}

struct Comm {
    ncclComm_t c = nullptr;
    int rank = 0, world = 1, device = 0;
};

static int rccl_error(const RcclApi& a, ncclResult_t r, const char* what)
{
    return set_error(MM_ERR_COMM, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}

// all-reduce(MIN), in place on device memory, enqueued on `st`
int comm_all_reduce_min(Comm* c, void* dev, int64_t n, bool is_f64, hipStream_t st)
{
    const RcclApi& a = rccl();
    if (!a.handle) return set_error(MM_ERR_COMM, a.error);
    if (n <= 0) return MM_OK;
    const ncclResult_t r = a.AllReduce(dev, dev, (size_t)n, is_f64 ? ncclFloat64 : ncclInt64, ncclMin, c->c, st);
    return r == ncclSuccess ? MM_OK : rccl_error(a, r, is_f64 ? "ncclAllReduce(MIN, f64)" : "ncclAllReduce(MIN, i64)");
}

int comm_rank(const Comm* c) { return c->rank; }
int comm_world(const Comm* c) { return c->world; }

}  // namespace mm

using namespace mm;

int mm_comm_unique_id(void* id)
{
    if (!id) return set_error(MM_ERR_INVALID, "mm_comm_unique_id: id == NULL");
    static_assert(sizeof(ncclUniqueId) == MM_COMM_ID_BYTES, "RCCL's unique id is 128 bytes");
    const RcclApi& a = rccl();
    if (!a.handle) return set_error(MM_ERR_COMM, a.error);
    ncclUniqueId u;
    const ncclResult_t r = a.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_error(a, r, "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof u);
    return MM_OK;
}

int mm_comm_init_rank(const void* id, int rank, int world, int device, mm_comm** out)
{
    if (!out) return set_error(MM_ERR_INVALID, "mm_comm_init_rank: out == NULL");
    *out = nullptr;
    if (!id || world <= 0 || rank < 0 || rank >= world) return set_error(MM_ERR_INVALID, "mm_comm_init_rank: bad id / rank / world");
    const RcclApi& a = rccl();
    if (!a.handle) return set_error(MM_ERR_COMM, a.error);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(MM_ERR_NO_DEVICE, "no HIP device available: this engine has no CPU fallback");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return set_error(MM_ERR_NO_DEVICE, "hipGetDevice failed"); }
    if (device >= n) return set_error(MM_ERR_INVALID, "device index out of range");
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return hip_error(he, "hipSetDevice");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    Comm* c = new Comm();
    c->rank = rank; c->world = world; c->device = device;
    const ncclResult_t r = a.CommInitRank(&c->c, world, u, rank);   // collective: every rank of the job calls it
    if (r != ncclSuccess) { delete c; return rccl_error(a, r, "ncclCommInitRank"); }
    *out = reinterpret_cast<mm_comm*>(c);
    return MM_OK;
}

void mm_comm_destroy(mm_comm* h)
{
    Comm* c = reinterpret_cast<Comm*>(h);
    if (!c) return;
    const RcclApi& a = rccl();
    if (a.handle && c->c) { (void)hipSetDevice(c->device); (void)a.CommDestroy(c->c); }
    delete c;
}

int mm_comm_rank(const mm_comm* h) { return h ? reinterpret_cast<const Comm*>(h)->rank : -1; }
int mm_comm_world(const mm_comm* h) { return h ? reinterpret_cast<const Comm*>(h)->world : 0; }

int mm_comm_version(void)
{
    const RcclApi& a = rccl();
    if (!a.handle) { set_error(MM_ERR_COMM, a.error); return -1; }
    int v = 0;
    return a.GetVersion(&v) == ncclSuccess ? v : -1;
}

int mm_comm_all_reduce_min_f64(mm_comm* h, double* dev, int64_t n, void* stream)
{
    if (!h || (n > 0 && !dev) || n < 0) return set_error(MM_ERR_INVALID, "mm_comm_all_reduce_min_f64: bad argument");
    Comm* c = reinterpret_cast<Comm*>(h);
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return hip_error(he, "hipSetDevice");
    return comm_all_reduce_min(c, dev, n, true, (hipStream_t)stream);
}

int mm_comm_broadcast(mm_comm* h, void* dev, int64_t bytes, int root, void* stream)
{
    if (!h || (bytes > 0 && !dev) || bytes < 0) return set_error(MM_ERR_INVALID, "mm_comm_broadcast: bad argument");
    Comm* c = reinterpret_cast<Comm*>(h);
    if (root < 0 || root >= c->world) return set_error(MM_ERR_INVALID, "mm_comm_broadcast: root is not a rank of the communicator");
    if (bytes == 0) return MM_OK;
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return hip_error(he, "hipSetDevice");
    const RcclApi& a = rccl();
    if (!a.handle) return set_error(MM_ERR_COMM, a.error);
    const ncclResult_t r = a.Broadcast(dev, dev, (size_t)bytes, ncclUint8, root, c->c, (hipStream_t)stream);
    return r == ncclSuccess ? MM_OK : rccl_error(a, r, "ncclBroadcast");
}

int mm_comm_all_reduce_min_i64(mm_comm* h, int64_t* dev, int64_t n, void* stream)
{
    if (!h || (n > 0 && !dev) || n < 0) return set_error(MM_ERR_INVALID, "mm_comm_all_reduce_min_i64: bad argument");
    Comm* c = reinterpret_cast<Comm*>(h);
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return hip_error(he, "hipSetDevice");
    return comm_all_reduce_min(c, dev, n, false, (hipStream_t)stream);
}

// nmpc_comm.hip -- multi-GPU result exchange for C / C++ callers of the C ABI (include/alore_nmpc.h, "comm" section).
//
// The north star's "RCCL all-gather of converged trajectories over xGMI" for a caller that is not Python: one process
// per GPU, each rank owns B problems (block partition of the global index, no data-path collective inside a solve), and
// alore_nmpc_comm_all_gather puts the trajectories (x, u), status and KKT value of all ranks on every rank with ONE
// grouped RCCL call (4 all-gathers, one launch) on the caller's stream -- so it overlaps with the next solve enqueued on
// another stream.  RCCL is loaded lazily (dlopen of librccl.so.1): the single-GPU library has no link-time dependency
// on it.  bench.py keeps using torch.distributed (the driver's launch contract); the two paths move the same bytes.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/alore_nmpc.h"

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId :43, ncclAllGather :678, data types :460-466)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int Result;
enum { kUint8 = 1, kInt32 = 2, kFloat32 = 7 };

struct Rccl {
    void* so = nullptr;
    Result (*GetUniqueId)(UniqueId*) = nullptr;
    Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    Result (*GroupStart)() = nullptr;
    Result (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(Result) = nullptr;
    bool loaded = false; // set only once every symbol has resolved
    bool load(std::string& err)
    {
        if (loaded) return true;
        so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!so) { err = std::string("dlopen librccl: ") + dlerror(); return false; }
#define SYM(field, name)                                                          \
    field = reinterpret_cast<decltype(field)>(dlsym(so, name));                   \
    if (!field) { err = std::string("librccl lacks ") + name; dlclose(so); so = nullptr; return false; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        loaded = true;
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_err;

} // namespace

struct alore_nmpc_comm {
    Comm comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
};

extern "C" {

const char* alore_nmpc_comm_last_error(void) { return g_err.c_str(); }

int alore_nmpc_comm_unique_id(char id[128])
{
    if (!id) return ALORE_NMPC_E_INVALID;
    if (!g_rccl.load(g_err)) return ALORE_NMPC_E_HIP;
    UniqueId u;
    const Result r = g_rccl.GetUniqueId(&u);
    if (r != 0) { g_err = g_rccl.GetErrorString(r); return ALORE_NMPC_E_HIP; }
    std::memcpy(id, u.internal, 128);
    return ALORE_NMPC_OK;
}

int alore_nmpc_comm_create(int n_ranks, int rank, const char id[128], int device, alore_nmpc_comm_handle* out)
{
    if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return ALORE_NMPC_E_INVALID;
    *out = nullptr;
    if (!g_rccl.load(g_err)) return ALORE_NMPC_E_HIP;
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice"; return ALORE_NMPC_E_NO_DEVICE; }
    alore_nmpc_comm* c = new (std::nothrow) alore_nmpc_comm;
    if (!c) return ALORE_NMPC_E_NOMEM;
    c->n_ranks = n_ranks; c->rank = rank; c->device = device;
    UniqueId u;
    std::memcpy(u.internal, id, 128);
    const Result r = g_rccl.CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != 0) { g_err = g_rccl.GetErrorString(r); delete c; return ALORE_NMPC_E_HIP; }
    *out = c;
    return ALORE_NMPC_OK;
}

int alore_nmpc_comm_destroy(alore_nmpc_comm_handle c)
{
    if (!c) return ALORE_NMPC_E_INVALID;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return ALORE_NMPC_OK;
}

int alore_nmpc_comm_all_gather(alore_nmpc_comm_handle c, const alore_nmpc_batch* local, int B, int N, const alore_nmpc_batch* all,
                               void* stream)
{
    if (!c || !local || !all || B <= 0 || N <= 0) return ALORE_NMPC_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) { g_err = "hipSetDevice"; return ALORE_NMPC_E_NO_DEVICE; }
    hipStream_t s = (hipStream_t)stream;
    struct { const void* src; void* dst; size_t count; int type; } m[4] = {
        {local->x, all->x, (size_t)B * 3 * (N + 1), kFloat32}, {local->u, all->u, (size_t)B * 2 * N, kFloat32},
        {local->status, all->status, (size_t)B, kInt32},       {local->kkt, all->kkt, (size_t)B, kFloat32}};
    // every rank must pass the same set of non-NULL members (a member is skipped when it is NULL on either side):
    // ranks that disagree would issue different collectives and hang
    if (!g_rccl.load(g_err)) return ALORE_NMPC_E_HIP;
    Result r = g_rccl.GroupStart();
    if (r != 0) { g_err = g_rccl.GetErrorString(r); return ALORE_NMPC_E_HIP; } // no group was opened: nothing to close
    for (int i = 0; i < 4 && r == 0; ++i)
        if (m[i].src && m[i].dst) r = g_rccl.AllGather(m[i].src, m[i].dst, m[i].count, m[i].type, c->comm, s);
    const Result e = g_rccl.GroupEnd();
    if (r == 0) r = e;
    if (r != 0) { g_err = g_rccl.GetErrorString(r); return ALORE_NMPC_E_HIP; }
    return ALORE_NMPC_OK;
}

} // extern "C"

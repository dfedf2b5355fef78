// RCCL behind the C-ABI: the one exchange of the data-parallel step is a sum all-reduce of
// the fp32 gradient buckets over xGMI.  The reference has no distributed code (GPU_COUNT = 0, config.py:47); these entry
// points are what a host in any language binds to drive the exchange without torch.distributed:
//     rank 0:     myolo_comm_unique_id(id)            -> ship the 128 bytes to the other ranks (any side channel)
//     every rank: myolo_comm_init(rank, n, id, &comm) -> ncclCommInitRank on the CURRENT HIP device
//                 myolo_allreduce_sum_f32(buf, n, comm, stream)   (in place, asynchronous on `stream`)
//                 myolo_comm_destroy(comm)
// librccl is resolved lazily with dlopen, so libmyolo_hip.so itself has no link-time dependency on it: a single-GPU
// user never loads RCCL, and inside a PyTorch process the already-loaded librccl.so is reused (same SONAME).
#include "myolo_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[192] = "no dlopen attempted";      // dlerror() text captured once, at load time (a later dlerror() may return NULL)
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl()
{
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
        const char* e = dlerror();
        snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen failed");
    }
    if (!g_rccl.h) return;
    snprintf(g_rccl.why, sizeof(g_rccl.why), "symbols missing");
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.h, "ncclCommCount");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy && g_rccl.CommCount;
}

int need_rccl(const char* who)
{
    std::call_once(g_once, load_rccl);
    if (!g_rccl.ok) {
        myolo_set_error("%s: librccl.so could not be loaded (%s)", who, g_rccl.why);
        return MYOLO_ECOMM;
    }
    return MYOLO_OK;
}

int fail(const char* who, ncclResult_t r)
{
    myolo_set_error("%s: RCCL error %d (%s)", who, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return MYOLO_ECOMM;
}
}  // namespace

extern "C" {

int myolo_comm_unique_id(void* id_out_128_bytes)
{
    MYOLO_REQUIRE(id_out_128_bytes, "comm_unique_id: null output");
    if (int rc = need_rccl("comm_unique_id")) return rc;
    static_assert(sizeof(ncclUniqueId) == MYOLO_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail("comm_unique_id", r);
    memcpy(id_out_128_bytes, &id, sizeof(id));
    return MYOLO_OK;
}

int myolo_comm_init(int rank, int nranks, const void* unique_id_128_bytes, void** comm_out)
{
    MYOLO_REQUIRE(unique_id_128_bytes && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments (rank %d of %d)", rank, nranks);
    if (int rc = need_rccl("comm_init")) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id_128_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, nranks, id, rank);
    if (r != ncclSuccess) return fail("comm_init", r);
    *comm_out = (void*)c;
    return MYOLO_OK;
}

int myolo_comm_size(void* comm, int* nranks_out)
{
    MYOLO_REQUIRE(comm && nranks_out, "comm_size: bad arguments");
    if (int rc = need_rccl("comm_size")) return rc;
    const ncclResult_t r = g_rccl.CommCount((ncclComm_t)comm, nranks_out);
    if (r != ncclSuccess) return fail("comm_size", r);
    return MYOLO_OK;
}

int myolo_allreduce_sum_f32(float* buf, int64_t n, void* comm, void* stream)
{
    MYOLO_REQUIRE(buf && comm && n >= 0, "allreduce_sum_f32: bad arguments");
    if (n == 0) return MYOLO_OK;
    if (int rc = need_rccl("allreduce_sum_f32")) return rc;
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail("allreduce_sum_f32", r);
    return MYOLO_OK;
}

int myolo_comm_destroy(void* comm)
{
    if (!comm) return MYOLO_OK;
    if (int rc = need_rccl("comm_destroy")) return rc;
    const ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)comm);
    if (r != ncclSuccess) return fail("comm_destroy", r);
    return MYOLO_OK;
}

}  // extern "C"

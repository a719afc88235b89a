// lipvq_comm.hip -- the path's only cross-GPU exchange: RCCL all-reduce of the per-batch code-usage histogram
// (and of fp32 payloads for the EMA extension / data-parallel gradients).  Host code only.
//
// RCCL is bound at run time: PyTorch ships its own librccl.so (SONAME librccl.so.1) and has it mapped in every
// process that imported torch; two RCCL copies in one process would each keep their own communicator state and
// topology caches, so the already-mapped copy is taken first (RTLD_NOLOAD) and the system copy only otherwise.
// A host without RCCL still loads the tokenizer library; these entries then report LIPVQ_EUNSUPPORTED.
#include <dlfcn.h>

#include <mutex>

#include "lipvq_common.h"

namespace {

// the part of rccl.h this file needs (an opaque communicator, two enums, five functions: a stable public ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[LIPVQ_COMM_ID_BYTES]; } ncclUniqueId;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_SUM = 0 };
enum { NCCL_INT64 = 4, NCCL_FLOAT32 = 7 };

struct Rccl {
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    const char* why = "not loaded";
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g_rccl.why = "librccl.so.1 not found"; return; }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GetErrorString) {
        g_rccl.why = "librccl.so.1 lacks an expected symbol";
        return;
    }
    g_rccl.ok = true;
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int nccl_fail(const Rccl* r, const char* what, int rc) {
    return fail(LIPVQ_EHIP, "%s: RCCL error %d (%s)", what, rc, r->GetErrorString(rc));
}

}  // namespace

extern "C" int lipvq_comm_unique_id(void* id128) {
    if (!id128) return fail(LIPVQ_EINVAL, "comm_unique_id: null pointer");
    const Rccl* r = rccl();
    if (!r) return fail(LIPVQ_EUNSUPPORTED, "comm_unique_id: RCCL unavailable (%s)", g_rccl.why);
    ncclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) return nccl_fail(r, "comm_unique_id", rc);
    memcpy(id128, id.internal, LIPVQ_COMM_ID_BYTES);
    return LIPVQ_OK;
}

// Collective over the `world` ranks (blocks until all have called it); the communicator belongs to the CURRENT device.
extern "C" int lipvq_comm_init(void** comm, const void* id128, int rank, int world) {
    if (!comm || !id128 || world <= 0 || rank < 0 || rank >= world) return fail(LIPVQ_EINVAL, "comm_init: bad argument");
    const Rccl* r = rccl();
    if (!r) return fail(LIPVQ_EUNSUPPORTED, "comm_init: RCCL unavailable (%s)", g_rccl.why);
    ncclUniqueId id;
    memcpy(id.internal, id128, LIPVQ_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    const int rc = r->CommInitRank(&c, world, id, rank);
    if (rc != NCCL_SUCCESS) return nccl_fail(r, "comm_init", rc);
    *comm = (void*)c;
    return LIPVQ_OK;
}

extern "C" int lipvq_comm_destroy(void* comm) {
    if (!comm) return LIPVQ_OK;
    const Rccl* r = rccl();
    if (!r) return fail(LIPVQ_EUNSUPPORTED, "comm_destroy: RCCL unavailable (%s)", g_rccl.why);
    const int rc = r->CommDestroy((ncclComm_t)comm);
    return rc == NCCL_SUCCESS ? LIPVQ_OK : nccl_fail(r, "comm_destroy", rc);
}

// int lipvq_allreduce_counts(int64_t* counts, int K, ncclComm_t, hipStream_t) -- in place, sum.
// 8 KiB at K = 1024: latency bound; RCCL picks its low-latency protocol for messages of this size by itself.
extern "C" int lipvq_allreduce_counts(int64_t* counts, int K, void* comm, void* stream) {
    if (!counts || K <= 0 || !comm) return fail(LIPVQ_EINVAL, "allreduce_counts: bad argument");
    const Rccl* r = rccl();
    if (!r) return fail(LIPVQ_EUNSUPPORTED, "allreduce_counts: RCCL unavailable (%s)", g_rccl.why);
    const int rc = r->AllReduce(counts, counts, (size_t)K, NCCL_INT64, NCCL_SUM, (ncclComm_t)comm, (hipStream_t)stream);
    return rc == NCCL_SUCCESS ? LIPVQ_OK : nccl_fail(r, "allreduce_counts", rc);
}

extern "C" int lipvq_allreduce_f32(float* buf, int64_t n, void* comm, void* stream) {
    if (!buf || n <= 0 || !comm) return fail(LIPVQ_EINVAL, "allreduce_f32: bad argument");
    const Rccl* r = rccl();
    if (!r) return fail(LIPVQ_EUNSUPPORTED, "allreduce_f32: RCCL unavailable (%s)", g_rccl.why);
    const int rc = r->AllReduce(buf, buf, (size_t)n, NCCL_FLOAT32, NCCL_SUM, (ncclComm_t)comm, (hipStream_t)stream);
    return rc == NCCL_SUCCESS ? LIPVQ_OK : nccl_fail(r, "allreduce_f32", rc);
}

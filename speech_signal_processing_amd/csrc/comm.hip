// Collectives of the C-ABI: the one exchange step of the hot path is an all-gather of the compact per-utterance
// decision records after scoring; centroid building / EM statistics over utterance shards would add one sum all-reduce.  RCCL
// (librccl.so: the ROCm build of the NCCL API, rings over xGMI inside a node) is resolved at RUN time with dlopen the first time a
// communicator is asked for — libsspgpu.so has no link-time dependency on it, and single-GPU callers never load it.
// Replaces nothing in the reference (it is single-process); it is what lets a non-Python caller of include/ssp.h run configs[3]:
// one process per GPU, utterances sharded by the caller, models replicated, ssp_allgather of the (argmax, best, ubm) records
// (the loops at GMM_UBM.py:183-197 and d_vector.py:315-318 have no cross-utterance term).
#include "common.hpp"

#include <dlfcn.h>

#include <mutex>

namespace ssp {
namespace {

// the slice of the NCCL C API this file uses (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE)
struct NcclUniqueId {
    char internal[128];
};
typedef void* nccl_comm_t;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_INT8 = 0, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0 };
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, NcclUniqueId, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    char why[256] = "";
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("SSP_RCCL_PATH");
        const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            snprintf(r.why, sizeof(r.why), "%s", dlerror());
        }
        if (!r.handle) return;
        r.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(r.handle, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(r.handle, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(r.handle, "ncclCommDestroy"));
        r.all_gather = reinterpret_cast<fn_all_gather>(dlsym(r.handle, "ncclAllGather"));
        r.all_reduce = reinterpret_cast<fn_all_reduce>(dlsym(r.handle, "ncclAllReduce"));
        r.error_string = reinterpret_cast<fn_error_string>(dlsym(r.handle, "ncclGetErrorString"));
        if (!(r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather && r.all_reduce)) {
            snprintf(r.why, sizeof(r.why), "librccl.so lacks a symbol of the NCCL API");
            r.handle = nullptr;
        }
    });
    return r;
}

int need_rccl(Rccl** out) {
    Rccl& r = rccl();
    if (!r.handle) SSP_FAIL(SSP_ERR_UNSUPPORTED, "RCCL is not available: %s (set SSP_RCCL_PATH to librccl.so)", r.why);
    *out = &r;
    return SSP_OK;
}

#define SSP_NCCL(r, expr)                                                                                              \
    do {                                                                                                               \
        const int e_ = (expr);                                                                                         \
        if (e_ != NCCL_SUCCESS) SSP_FAIL(SSP_ERR_HIP, "%s failed: %s", #expr, (r)->error_string ? (r)->error_string(e_) : "rccl error"); \
    } while (0)

}  // namespace
}  // namespace ssp

using namespace ssp;

extern "C" {

int ssp_comm_unique_id(void* id_out) {
    if (!id_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_comm_unique_id: null output");
    Rccl* r;
    SSP_TRY(need_rccl(&r));
    NcclUniqueId id;
    SSP_NCCL(r, r->get_unique_id(&id));
    memcpy(id_out, id.internal, SSP_COMM_ID_BYTES);
    return SSP_OK;
}

int ssp_comm_init(ssp_ctx* ctx, int rank, int nranks, const void* unique_id) {
    SSP_TRY(use_ctx(ctx));
    if (nranks < 1 || rank < 0 || rank >= nranks || !unique_id) SSP_FAIL(SSP_ERR_INVALID, "ssp_comm_init: rank %d of %d / null id", rank, nranks);
    if (ctx->comm) SSP_FAIL(SSP_ERR_INVALID, "ssp_comm_init: this ctx already has a communicator (ssp_comm_destroy first)");
    Rccl* r;
    SSP_TRY(need_rccl(&r));
    NcclUniqueId id;
    memcpy(id.internal, unique_id, SSP_COMM_ID_BYTES);
    nccl_comm_t c = nullptr;
    SSP_NCCL(r, r->comm_init_rank(&c, nranks, id, rank));  // (blocks until every rank of the id has arrived)
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    return SSP_OK;
}

int ssp_comm_destroy(ssp_ctx* ctx) {
    if (!ctx || !ctx->comm) return SSP_OK;
    SSP_TRY(use_ctx(ctx));
    SSP_HIP(hipStreamSynchronize(ctx->stream));
    Rccl* r;
    SSP_TRY(need_rccl(&r));
    SSP_NCCL(r, r->comm_destroy(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_size = 1;
    return SSP_OK;
}

int ssp_comm_info(const ssp_ctx* ctx, int* rank, int* nranks) {
    if (!ctx) SSP_FAIL(SSP_ERR_INVALID, "null ssp_ctx");
    if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
    if (nranks) *nranks = ctx->comm ? ctx->comm_size : 1;
    return SSP_OK;
}

int ssp_allgather(ssp_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank) {
    ssp::TraceRange trace_("ssp_allgather");
    SSP_TRY(use_ctx(ctx));
    if (bytes_per_rank == 0) return SSP_OK;
    if (!send || !recv) SSP_FAIL(SSP_ERR_INVALID, "ssp_allgather: null buffer");
    if (!ctx->comm) {  // a ctx without a communicator is a world of one: the gather is a copy
        if (send != recv) SSP_HIP(hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return SSP_OK;
    }
    Rccl* r;
    SSP_TRY(need_rccl(&r));
    SSP_NCCL(r, r->all_gather(send, recv, bytes_per_rank, NCCL_INT8, ctx->comm, ctx->stream));
    return SSP_OK;
}

int ssp_allreduce_sum(ssp_ctx* ctx, void* buf, size_t count, int is_f64) {
    ssp::TraceRange trace_("ssp_allreduce_sum");
    SSP_TRY(use_ctx(ctx));
    if (count == 0 || !ctx->comm) return SSP_OK;
    if (!buf) SSP_FAIL(SSP_ERR_INVALID, "ssp_allreduce_sum: null buffer");
    Rccl* r;
    SSP_TRY(need_rccl(&r));
    SSP_NCCL(r, r->all_reduce(buf, buf, count, is_f64 ? NCCL_FLOAT64 : NCCL_FLOAT32, NCCL_SUM, ctx->comm, ctx->stream));
    return SSP_OK;
}

}  // extern "C"

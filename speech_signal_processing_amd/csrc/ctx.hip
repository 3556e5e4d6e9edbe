// Context, error reporting and segment handles of libsspgpu.so.
#include "common.hpp"

#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <unordered_set>

namespace ssp {

static thread_local char g_err[512] = "";

void ctx_register(const ssp_ctx* c);
void ctx_unregister(const ssp_ctx* c);

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- roctx ranges (TraceRange, common.hpp): resolved once per process, only when SSP_ROCTX is set
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)();
struct Roctx {
    roctx_push_t push = nullptr;
    roctx_pop_t pop = nullptr;
    Roctx() {
        const char* e = getenv("SSP_ROCTX");
        if (!e || !*e || *e == '0') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}
}  // namespace

TraceRange::TraceRange(const char* name) : on(false) {
    const Roctx& r = roctx();
    if (r.push) {
        r.push(name);
        on = true;
    }
}
TraceRange::~TraceRange() {
    if (on) roctx().pop();
}

// ---- registry of live contexts (see quiesce_ctx in common.hpp)
namespace {
std::mutex g_live_mu;
std::unordered_set<const ssp_ctx*>& live_set() {
    static auto* s = new std::unordered_set<const ssp_ctx*>;  // (never destroyed: handles may be finalized during process exit)
    return *s;
}
}  // namespace
void ctx_register(const ssp_ctx* c) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    live_set().insert(c);
}
void ctx_unregister(const ssp_ctx* c) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    live_set().erase(c);
}
// hipStreamSynchronize on a stream its owner has already destroyed (a borrowed stream at process exit) does not return an error on
// ROCm 7.2 — it throws std::bad_variant_access out of the C API; nothing may escape a destroy function
static void sync_quietly(hipStream_t s) {
    try {
        (void)hipStreamSynchronize(s);
    } catch (...) {
    }
}
void quiesce_ctx(const ssp_ctx* ctx) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (!live_set().count(ctx)) return;
    (void)hipSetDevice(ctx->device);
    sync_quietly(ctx->stream);
}

int segments_make(ssp_ctx* ctx, const int64_t* offsets, int64_t n, ssp_segments** out) {
    if (!out) SSP_FAIL(SSP_ERR_INVALID, "segments: null out");
    *out = nullptr;
    SSP_TRY(use_ctx(ctx));
    if (n < 0 || (!offsets)) SSP_FAIL(SSP_ERR_INVALID, "segments: null offsets or negative count");
    if (offsets[0] < 0) SSP_FAIL(SSP_ERR_INVALID, "segments: offsets[0] < 0");
    for (int64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) SSP_FAIL(SSP_ERR_INVALID, "segments: offsets decrease at %lld", (long long)i);
    ssp_segments* s = new (std::nothrow) ssp_segments;
    if (!s) SSP_FAIL(SSP_ERR_NOMEM, "segments: host alloc");
    static std::atomic<uint64_t> next_serial{1};
    s->ctx = ctx;
    s->serial = next_serial.fetch_add(1);
    s->n = n;
    s->host.assign(offsets, offsets + n + 1);
    int rc = s->dev.alloc(sizeof(int64_t) * (size_t)(n + 1));
    if (rc == SSP_OK) {
        hipError_t e = hipMemcpyAsync(s->dev.p, s->host.data(), sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice,
                                      ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("segments upload failed: %s", hipGetErrorString(e));
            rc = SSP_ERR_HIP;
        }
    }
    if (rc != SSP_OK) {
        delete s;
        return rc;
    }
    *out = s;
    return SSP_OK;
}

// fills the whole LDS of the CU it lands on with one 32-bit pattern (test aid: a kernel that reads LDS it never wrote then reads
// this pattern — e.g. a NaN — instead of whatever the previous kernel left behind)
__global__ __launch_bounds__(256) void poison_lds_kernel(uint32_t pattern, uint32_t* sink) {
    extern __shared__ uint32_t lds_all[];
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += 256) lds_all[i] = pattern;
    __syncthreads();
    if (lds_all[(threadIdx.x * 97) % n] != pattern) sink[0] = 1;  // (keeps the stores alive)
}

}  // namespace ssp

extern "C" {

int ssp_debug_poison_lds(ssp_ctx* ctx, uint32_t pattern) {
    SSP_TRY(ssp::use_ctx(ctx));
    SSP_TRY(ctx->scratch[5].reserve(16));
    SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ssp::poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // one workgroup owns a CU's whole LDS: a few rounds over the CUs reach every one of them
    hipLaunchKernelGGL(ssp::poison_lds_kernel, dim3(4 * ctx->num_cu), dim3(256), 160 * 1024, ctx->stream, pattern, ctx->scratch[5].as<uint32_t>());
    SSP_HIP(hipGetLastError());
    SSP_HIP(hipStreamSynchronize(ctx->stream));
    return SSP_OK;
}

int ssp_abi_version(void) { return SSP_ABI_VERSION; }

const char* ssp_last_error(void) { return ssp::g_err; }

int ssp_ctx_create(int device, void* stream, int borrow_stream, ssp_ctx** out) {
    if (!out) SSP_FAIL(SSP_ERR_INVALID, "ssp_ctx_create: null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) SSP_FAIL(SSP_ERR_NODEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) SSP_FAIL(SSP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    SSP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SSP_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    ssp_ctx* c = new (std::nothrow) ssp_ctx;
    if (!c) SSP_FAIL(SSP_ERR_NOMEM, "ctx: host alloc");
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (borrow_stream) {
        c->stream = static_cast<hipStream_t>(stream);
        c->owns_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            SSP_FAIL(SSP_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        c->owns_stream = true;
    }
    ssp::ctx_register(c);
    *out = c;
    return SSP_OK;
}

int ssp_ctx_destroy(ssp_ctx* ctx) {
    if (!ctx) return SSP_OK;
    ssp::ctx_unregister(ctx);
    (void)hipSetDevice(ctx->device);
    ssp::sync_quietly(ctx->stream);
    (void)ssp_comm_destroy(ctx);
    if (ctx->pipe) {
        ssp::sync_quietly(ctx->pipe->h2d);
        ssp::sync_quietly(ctx->pipe->d2h);
        delete ctx->pipe;
        ctx->pipe = nullptr;
    }
    if (ctx->pinned_words) (void)hipHostFree(ctx->pinned_words);
    for (hipEvent_t& ev : ctx->order_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SSP_OK;
}

int ssp_ctx_sync(ssp_ctx* ctx) {
    SSP_TRY(ssp::use_ctx(ctx));
    SSP_HIP(hipStreamSynchronize(ctx->stream));
    return SSP_OK;
}

// stream ordering without a host wait: an event recorded on one stream, waited for by the other.  One event per ctx and direction,
// created on first use and destroyed with the ctx (an event may be recorded again once the wait that used it has been queued)
static int order_streams(ssp_ctx* ctx, int which, hipStream_t first, hipStream_t then) {
    if (first == then) return SSP_OK;
    hipEvent_t& ev = ctx->order_ev[which];
    if (!ev) SSP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    SSP_HIP(hipEventRecord(ev, first));
    SSP_HIP(hipStreamWaitEvent(then, ev, 0));
    return SSP_OK;
}

int ssp_ctx_wait_stream(ssp_ctx* ctx, void* other_stream) {
    SSP_TRY(ssp::use_ctx(ctx));
    return order_streams(ctx, 0, static_cast<hipStream_t>(other_stream), ctx->stream);
}

int ssp_ctx_signal_stream(ssp_ctx* ctx, void* other_stream) {
    SSP_TRY(ssp::use_ctx(ctx));
    return order_streams(ctx, 1, ctx->stream, static_cast<hipStream_t>(other_stream));
}

int ssp_segments_create(ssp_ctx* ctx, const int64_t* offsets, int64_t n_seg, ssp_segments** out) {
    return ssp::segments_make(ctx, offsets, n_seg, out);
}

int ssp_segments_destroy(ssp_segments* seg) {
    if (!seg) return SSP_OK;
    ssp::quiesce_ctx(seg->ctx);
    delete seg;
    return SSP_OK;
}

int ssp_segments_count(const ssp_segments* seg, int64_t* n_seg, int64_t* total) {
    if (!seg) SSP_FAIL(SSP_ERR_INVALID, "null segments");
    if (n_seg) *n_seg = seg->n;
    if (total) *total = seg->total();
    return SSP_OK;
}

int ssp_segments_read(const ssp_segments* seg, int64_t* offsets_out) {
    if (!seg || !offsets_out) SSP_FAIL(SSP_ERR_INVALID, "null segments / output");
    memcpy(offsets_out, seg->host.data(), sizeof(int64_t) * seg->host.size());
    return SSP_OK;
}

}  // extern "C"

// Host-side staging of SSP_HOST calls (included twice by common.hpp: part 1 before ssp_ctx — the pool it holds —, part 2 behind it — the
// per-operand helper).  No device code here: the file is deliberately NOT part of bench.py's kernel-source hash.
#if SSP_STAGING_PART == 1
#include <mutex>
namespace ssp {
// Staging buffers of SSP_HOST calls, kept by the ctx between calls: the reference's callers loop over utterances in Python, and a
// hipMalloc + hipFree pair per staged operand costs up to 0.7 ms each once the allocator has no small block at hand (GMM_UBM.delta
// took 1.4 ms a call in such a state, 0.06 ms with the buffers kept).  A slot is taken for the life of one Staged; all work of a
// ctx is ordered on its one stream, so the next call may overwrite a slot without a host wait.  Buffers above KEEP_MAX are not kept.
// (a template over the buffer type so that tests/native/stagepool_threads.cpp can drive the slot logic from several threads under
//  ThreadSanitizer with a malloc-backed buffer, without a HIP runtime)
template <class Buf>
struct StagePoolT {
    static constexpr int SLOTS = 8;
    static constexpr size_t KEEP_MAX = (size_t)64 << 20;
    Buf slot[SLOTS];
    bool busy[SLOTS] = {};
    std::mutex mu;  // (a ctx is not thread-safe, but before the pool two host-pointer calls on one ctx never shared a staging buffer: keep it so)
    void give_back(int i) {
        std::lock_guard<std::mutex> g(mu);
        busy[i] = false;
    }
    // a free slot of at least n bytes: the smallest that fits, else the smallest free one grown to n; -1 = none free or n too large
    int take(size_t n, int* rc) {
        *rc = SSP_OK;
        if (n > KEEP_MAX) return -1;
        std::lock_guard<std::mutex> g(mu);
        int fit = -1, spare = -1;
        for (int i = 0; i < SLOTS; ++i) {
            if (busy[i]) continue;
            if (slot[i].p && slot[i].bytes >= n && (fit < 0 || slot[i].bytes < slot[fit].bytes)) fit = i;
            if (spare < 0 || slot[i].bytes < slot[spare].bytes) spare = i;
        }
        if (fit < 0) {
            if (spare < 0) return -1;
            size_t want = n < 4096 ? 4096 : n + n / 4;  // headroom: utterance lengths vary from call to call
            if (want > KEEP_MAX) want = KEEP_MAX;
            *rc = slot[spare].alloc(want);
            if (*rc != SSP_OK) return -1;
            fit = spare;
        }
        busy[fit] = true;
        return fit;
    }
};
#ifndef SSP_STAGING_NO_HIP
using StagePool = StagePoolT<DevBuf>;
// Pipeline of large SSP_HOST MFCC batches (mfcc_plan.hip, mfcc_run_host_sliced): the batch goes through RING slots of slice size — slice
// i + 1 is copied in (own stream) while slice i computes (the ctx stream) and slice i - 1's features are copied back (third stream).
// The slots, streams and events live on the ctx, are made on first use and kept (grow-only); ssp_ctx_destroy frees them.
struct HostPipe {
    static constexpr int RING = 3;
    hipStream_t h2d = nullptr, d2h = nullptr;
    DevBuf in[RING], raw[RING], out[RING];          // fp32 samples | int16 samples as copied in (widened into `in`) | features
    hipEvent_t in_ready[RING] = {}, computed[RING] = {}, out_done[RING] = {};
    int init() {
        if (h2d) return SSP_OK;
        SSP_HIP(hipStreamCreateWithFlags(&h2d, hipStreamNonBlocking));
        SSP_HIP(hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking));
        for (int i = 0; i < RING; ++i) {
            SSP_HIP(hipEventCreateWithFlags(&in_ready[i], hipEventDisableTiming));
            SSP_HIP(hipEventCreateWithFlags(&computed[i], hipEventDisableTiming));
            SSP_HIP(hipEventCreateWithFlags(&out_done[i], hipEventDisableTiming));
        }
        return SSP_OK;
    }
    ~HostPipe() {
        for (int i = 0; i < RING; ++i) {
            if (in_ready[i]) (void)hipEventDestroy(in_ready[i]);
            if (computed[i]) (void)hipEventDestroy(computed[i]);
            if (out_done[i]) (void)hipEventDestroy(out_done[i]);
        }
        if (h2d) (void)hipStreamDestroy(h2d);
        if (d2h) (void)hipStreamDestroy(d2h);
    }
};
#endif  // SSP_STAGING_NO_HIP
}  // namespace ssp
#elif SSP_STAGING_PART == 2
#include <algorithm>
#include <cstdlib>
namespace ssp {
// Staging helper for SSP_HOST calls: device copy of a host input / device scratch for an output.
struct Staged {
    DevBuf own;  // operands too large for the ctx's pool (or when all its slots are taken)
    const ssp_ctx* pool = nullptr;
    int slot = -1;
    void* p = nullptr;
    Staged() = default;
    Staged(const Staged&) = delete;
    Staged& operator=(const Staged&) = delete;
    ~Staged() {
        if (slot >= 0) pool->stage.give_back(slot);
    }
    int get(const ssp_ctx* ctx, size_t bytes) {
        int rc;
        if (slot >= 0) ctx->stage.give_back(slot);
        slot = ctx->stage.take(bytes, &rc);  // (a slot that could not grow is no error: the operand gets a buffer of its own for the call)
        if (slot >= 0) {
            pool = ctx;
            p = ctx->stage.slot[slot].p;
            return SSP_OK;
        }
        rc = own.alloc(bytes);
        p = own.p;
        return rc;
    }
    const void* in(const ssp_ctx* ctx, const void* host, size_t bytes, int where, int* rc) {
        *rc = SSP_OK;
        if (where == SSP_DEVICE || host == nullptr) return host;
        *rc = get(ctx, bytes);
        if (*rc != SSP_OK) return nullptr;
        hipError_t e = hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            set_error("hipMemcpyAsync H2D failed: %s", hipGetErrorString(e));
            *rc = SSP_ERR_HIP;
            return nullptr;
        }
        return p;
    }
    void* out(const ssp_ctx* ctx, void* host, size_t bytes, int where, int* rc) {
        *rc = SSP_OK;
        if (where == SSP_DEVICE || host == nullptr) return host;
        *rc = get(ctx, bytes);
        return *rc == SSP_OK ? p : nullptr;
    }
    int back(const ssp_ctx* ctx, void* host, size_t bytes, int where) {
        if (where == SSP_DEVICE || host == nullptr) return SSP_OK;
        SSP_HIP(hipMemcpyAsync(host, p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return SSP_OK;
    }
};

// feed_rows — a HOST matrix through the ctx's ring of device slots, copied in ahead of its consumer (the scorers' host-fed batches: the
// rows of slice i + 1 and i + 2 cross PCIe on the copy stream while `consume` has slice i's kernels on the ctx stream).  `cuts`: row
// indices 0 = c0 < c1 < ... < cn of the slices; consume(i, dev) launches slice i's work on ctx->stream, dev = device copy of rows
// [cuts[i], cuts[i + 1]).  The host waits for slice i's kernels before it issues the copy that reuses a slot (and consumers may
// therefore reuse host-side tables they upload per slice); on an error everything in flight is drained before it goes up.
template <class F>
static int feed_rows(ssp_ctx* ctx, const void* host, size_t row_bytes, const std::vector<int64_t>& cuts, F&& consume) {
    const int n = (int)cuts.size() - 1;
    if (n <= 0) return SSP_OK;
    if (!ctx->pipe) {
        ctx->pipe = new (std::nothrow) HostPipe;
        if (!ctx->pipe) SSP_FAIL(SSP_ERR_NOMEM, "host alloc (pipeline)");
    }
    HostPipe& hp = *ctx->pipe;
    SSP_TRY(hp.init());
    hipStream_t cs = ctx->stream;
    size_t max_bytes = 0;
    for (int i = 0; i < n; ++i) max_bytes = std::max(max_bytes, (size_t)(cuts[(size_t)i + 1] - cuts[(size_t)i]) * row_bytes);
    bool grow = false;
    for (int k = 0; k < HostPipe::RING; ++k) grow = grow || hp.in[k].bytes < max_bytes + 256;
    if (grow) {  // (a slot that has to grow may still be read by work of an earlier call: everything is drained first)
        SSP_HIP(hipStreamSynchronize(cs));
        SSP_HIP(hipStreamSynchronize(hp.h2d));
        SSP_HIP(hipStreamSynchronize(hp.d2h));
        for (int k = 0; k < HostPipe::RING; ++k) SSP_TRY(hp.in[k].reserve(max_bytes + 256));
    }
    auto body = [&]() -> int {
        auto issue = [&](int i) -> int {
            const int k = i % HostPipe::RING;
            SSP_HIP(hipMemcpyAsync(hp.in[k].p, static_cast<const char*>(host) + (size_t)cuts[(size_t)i] * row_bytes,
                                   (size_t)(cuts[(size_t)i + 1] - cuts[(size_t)i]) * row_bytes, hipMemcpyHostToDevice, hp.h2d));
            SSP_HIP(hipEventRecord(hp.in_ready[k], hp.h2d));
            return SSP_OK;
        };
        SSP_HIP(hipEventRecord(hp.computed[0], cs));   // the copy stream starts behind what the ctx stream holds (the slots' last readers)
        SSP_HIP(hipStreamWaitEvent(hp.h2d, hp.computed[0], 0));
        for (int i = 0; i < n && i < HostPipe::RING - 1; ++i) SSP_TRY(issue(i));
        for (int i = 0; i < n; ++i) {
            if (i + HostPipe::RING - 1 < n) SSP_TRY(issue(i + HostPipe::RING - 1));   // its slot's last reader was slice i - 1: waited for below
            SSP_HIP(hipStreamWaitEvent(cs, hp.in_ready[i % HostPipe::RING], 0));
            SSP_TRY(consume(i, hp.in[i % HostPipe::RING].p));
            SSP_HIP(hipStreamSynchronize(cs));
        }
        return SSP_OK;
    };
    const int rc = body();
    if (rc != SSP_OK) {
        (void)hipStreamSynchronize(cs);
        (void)hipStreamSynchronize(hp.h2d);
    }
    return rc;
}

static inline size_t host_slice_bytes() {
    size_t mb = 64;  // ~1.2 ms of PCIe per slice: long against a launch's host cost, short against the batch (fill + drain = two slices)
    if (const char* e = getenv("SSP_HOST_SLICE_MB")) mb = (size_t)std::max(1, atoi(e));
    return mb << 20;
}
}  // namespace ssp
#endif

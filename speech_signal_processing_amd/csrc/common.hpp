// Shared host-side plumbing for libsspgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/ssp.h"

namespace ssp {

// numpy.maximum / ndarray.max(): a NaN on either side is the result (fmaxf returns the other operand).  The librosa dialect's floor
// max(amin, S) and its top_db clamp max(S_db, S_db.max() - top_db) (MFCC_DTW.py:28-31 through librosa.power_to_db) are such maxima: one
// NaN sample makes the whole utterance's features NaN there, and the kernels must not hand back finite numbers instead.
#if defined(__HIPCC__)
__device__ __forceinline__ float nanmax(float a, float b) {
    const float r = fmaxf(a, b);
    return (a != a || b != b) ? __builtin_nanf("") : r;
}
#endif


void set_error(const char* fmt, ...);

#define SSP_FAIL(code, ...)            \
    do {                               \
        ::ssp::set_error(__VA_ARGS__); \
        return (code);                 \
    } while (0)

#define SSP_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ::ssp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                      \
            return e_ == hipErrorOutOfMemory ? SSP_ERR_NOMEM : SSP_ERR_HIP;                  \
        }                                                                                    \
    } while (0)

#define SSP_TRY(expr)          \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != SSP_OK) return rc_; \
    } while (0)

// roctx range around a C-ABI call (tracing): a named range on the calling thread while SSP_ROCTX=1 is in the
// environment (`rocprofv3 --marker-trace --kernel-trace -- python3 ...` then shows which entry point launched which kernels).  The roctx
// library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so): no link-time dependency, no cost when off.
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    bool on;
};

template <class T>
static inline T ceil_div(T a, T b) {
    return (a + b - 1) / b;
}

// RAII device buffer (used for per-call staging in SSP_HOST mode and for plan tables).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        SSP_HIP(hipMalloc(&p, n));
        bytes = n;
        return SSP_OK;
    }
    int reserve(size_t n) { return (p && bytes >= n) ? SSP_OK : alloc(n); }  // grow-only scratch
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

}  // namespace ssp
#define SSP_STAGING_PART 1
#include "staging.hpp"  // StagePool (host side only: kept out of this header, whose hash ties PMC files to the kernels' source)
#undef SSP_STAGING_PART
namespace ssp {

// hipEvent pair around a region on the ctx stream.
struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    bool on = false;
    int start(bool enable, hipStream_t s) {
        on = enable;
        if (!on) return SSP_OK;
        SSP_HIP(hipEventCreate(&a));
        SSP_HIP(hipEventCreate(&b));
        SSP_HIP(hipEventRecord(a, s));
        return SSP_OK;
    }
    int stop(hipStream_t s, float* ms) {
        if (!on) return SSP_OK;
        SSP_HIP(hipEventRecord(b, s));
        SSP_HIP(hipEventSynchronize(b));
        SSP_HIP(hipEventElapsedTime(ms, a, b));
        return SSP_OK;
    }
    ~Timer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

}  // namespace ssp

struct ssp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cu = 256;
    // grow-only device scratch shared by the entry points that are called in a loop (one EM iteration per call): a ctx is
    // not thread-safe and its work is ordered on one stream, so consecutive calls may reuse the same buffers
    ssp::DevBuf scratch[6];
    // RCCL communicator of this ctx (comm.hip; null = a world of one)
    void* comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    hipEvent_t order_ev[2] = {nullptr, nullptr};  // ssp_ctx_wait_stream / ssp_ctx_signal_stream
    mutable int32_t cos_last_rescored = 0;  // rows the last ssp_cosine_identify2(precision >= 1) call scored again in fp32
    mutable int32_t cos_last_split = 0;     // ... rows its precision-2 cascade handed from the bf16 sweep to the bf16x3 sweep
    mutable int32_t cos_auto_choice = -1, cos_auto_pilot_rows = 0, cos_auto_to_x3 = 0, cos_auto_to_f32 = 0;  // ssp_cosine_identify2(precision = 3): the pilot's verdict
    mutable bool cos_counts_pending = false;  // the two counts still sit in cos_count on the device (a device-pointer call does not wait for them)
    // cosine scorer's scratch, kept between calls (the reference calls it in a loop): packed centroid images, the lists of close calls, counts
    ssp::DevBuf cos_img16, cos_img, cos_list1, cos_list2, cos_count, cos_inc;
    mutable ssp::StagePool stage;  // staging buffers of SSP_HOST calls
    int32_t* pinned_words = nullptr;  // 64 bytes of pinned host memory for small read-backs (precision-auto pilots)
    ssp::HostPipe* pipe = nullptr;  // slots / streams of the sliced host-fed MFCC path (staging.hpp; made on first use)
};

struct ssp_segments {
    ssp_ctx* ctx = nullptr;
    uint64_t serial = 0;        // unique per handle (work-table caches key on it, not on the address)
    int64_t n = 0;
    std::vector<int64_t> host;  // n+1
    ssp::DevBuf dev;            // int64[n+1]
    int64_t total() const { return host.empty() ? 0 : host.back() - host.front(); }
    int64_t max_len() const {
        int64_t m = 0;
        for (int64_t i = 0; i < n; ++i) m = host[i + 1] - host[i] > m ? host[i + 1] - host[i] : m;
        return m;
    }
};

namespace ssp {
// make `dev` the current device for this thread
static inline int use_ctx(const ssp_ctx* ctx) {
    if (!ctx) SSP_FAIL(SSP_ERR_INVALID, "null ssp_ctx");
    SSP_HIP(hipSetDevice(ctx->device));
    return SSP_OK;
}
int segments_make(ssp_ctx* ctx, const int64_t* offsets, int64_t n, ssp_segments** out);
// Handles (plans, scorers, networks, segments) keep a pointer to their ctx and may be destroyed AFTER it (garbage collectors run
// finalizers of dead object groups in any order — CPython at interpreter exit does).  Their destroy functions call this instead of
// touching the ctx: it waits for the ctx's stream when the ctx is still alive and does nothing when it is gone (ssp_ctx_destroy
// has already drained the stream); device buffers are then freed without reference to the ctx.
void quiesce_ctx(const ssp_ctx* ctx);

}  // namespace ssp
#define SSP_STAGING_PART 2
#include "staging.hpp"  // Staged
#undef SSP_STAGING_PART

// Calibration kernels of the bench line (ssp_calibrate): what THIS box sustains on two textbook loads, measured in the process that
// measures the hot path, so that two bench lines from different boxes (clock / power-cap state differs by a few per cent between the
// boxes of a pool) can be compared after division by these figures.
//   * copy:  float4 grid-stride copy of a buffer far larger than the caches -> GB/s (read + write); the guide's reference is 6.29 TB/s
//   * fma:   eight independent packed-fp32 FMA chains per lane (v_pk_fma_f32) -> TFLOP/s; the reference is the 157.3 TF vector peak
// Both run ~20 ms by default: long enough for the power governor to settle into the state the hot path sees.
#include "common.hpp"

namespace ssp {
namespace {
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_copy_kernel(const v4f_t* __restrict__ src, v4f_t* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
        // four independent 16-byte loads in flight per lane
        const size_t i1 = i + 256, i2 = i + 512, i3 = i + 768;
        const v4f_t a = src[i];
        const v4f_t b = i1 < n4 ? src[i1] : a;
        const v4f_t c = i2 < n4 ? src[i2] : a;
        const v4f_t d = i3 < n4 ? src[i3] : a;
        dst[i] = a;
        if (i1 < n4) dst[i1] = b;
        if (i2 < n4) dst[i2] = c;
        if (i3 < n4) dst[i3] = d;
    }
}

__global__ __launch_bounds__(256) void calib_fma_kernel(float* out, int iters, float seed) {
    v2f_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = v2f_t{seed + (float)k, seed - (float)(threadIdx.x & 7)};
    const v2f_t m = v2f_t{0.999f, 1.001f}, c = v2f_t{1e-3f, -1e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_fma(a[k], m, c);
    }
    v2f_t s = a[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += a[k];
    if (s.x + s.y == 12345.678f) out[0] = s.x;  // (keeps the chains alive; never true)
}
}  // namespace
}  // namespace ssp

extern "C" int ssp_calibrate(ssp_ctx* ctx, double target_ms, double* copy_gbs, double* fma_tflops, double* copy_ms, double* fma_ms) {
    using namespace ssp;
    SSP_TRY(use_ctx(ctx));
    if (!copy_gbs || !fma_tflops) SSP_FAIL(SSP_ERR_INVALID, "ssp_calibrate: null output");
    if (!(target_ms > 0.0)) target_ms = 20.0;
    hipStream_t s = ctx->stream;
    hipEvent_t e0, e1;
    SSP_HIP(hipEventCreate(&e0));
    SSP_HIP(hipEventCreate(&e1));
    int rc = SSP_OK;
    float ms = 0.f;
    // ---- copy: 1 GiB -> 1 GiB per launch (2 GiB of traffic, ~0.35 ms at 6.3 TB/s); launches back to back for ~target_ms
    {
        const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
        DevBuf src, dst;
        if ((rc = src.alloc(bytes)) != SSP_OK || (rc = dst.alloc(bytes)) != SSP_OK) goto done;
        if (hipMemsetAsync(src.p, 0x11, bytes, s) != hipSuccess) { rc = SSP_ERR_HIP; set_error("ssp_calibrate: memset"); goto done; }
        const int grid = ctx->num_cu * 8;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(calib_copy_kernel, dim3(grid), dim3(256), 0, s, src.as<v4f_t>(), dst.as<v4f_t>(), n4);
        const int reps = std::max(4, (int)(target_ms / 0.36));
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_copy_kernel, dim3(grid), dim3(256), 0, s, src.as<v4f_t>(), dst.as<v4f_t>(), n4);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = SSP_ERR_HIP; set_error("ssp_calibrate: copy timing"); goto done; }
        *copy_gbs = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
        if (copy_ms) *copy_ms = ms;
    }
    // ---- fma: every SIMD busy with packed FMAs; flop = threads x iters x 8 chains x 2 lanes-of-the-pair x 2
    {
        const int grid = ctx->num_cu * 16, iters_probe = 4096;
        DevBuf out;
        if ((rc = out.alloc(64)) != SSP_OK) goto done;
        hipLaunchKernelGGL(calib_fma_kernel, dim3(grid), dim3(256), 0, s, out.as<float>(), iters_probe, 1.0f);
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(calib_fma_kernel, dim3(grid), dim3(256), 0, s, out.as<float>(), iters_probe, 1.0f);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = SSP_ERR_HIP; set_error("ssp_calibrate: fma probe"); goto done; }
        const int iters = (int)std::min(4.0e6, std::max(4096.0, iters_probe * target_ms / std::max(ms, 1e-3f)));
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(calib_fma_kernel, dim3(grid), dim3(256), 0, s, out.as<float>(), iters, 1.0f);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = SSP_ERR_HIP; set_error("ssp_calibrate: fma timing"); goto done; }
        *fma_tflops = (double)grid * 256.0 * (double)iters * 8.0 * 2.0 * 2.0 / (ms * 1e-3) / 1e12;
        if (fma_ms) *fma_ms = ms;
    }
done:
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

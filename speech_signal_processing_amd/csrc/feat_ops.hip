// Stand-alone delta (GMM_UBM.py:53-69 == d_vector.py:143-160) and per-utterance CMVN
// (sklearn.preprocessing.scale as called at GMM_UBM.py:93) on (frames x dim) feature matrices.
// Both are HBM-bound streaming kernels; the fused MFCC pass does the same work in LDS.
#include "common.hpp"

namespace ssp {

// one thread per output element; utterance found by binary search over frame offsets
__device__ __forceinline__ int find_segment(const int64_t* __restrict__ off, int n_seg, int64_t row) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= row) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// a workgroup owns DELTA_RB consecutive rows: one binary search for the block's first row, then every element walks forward from
// that segment (the first version searched 17 dependent levels per ELEMENT: 0.5 TB/s)
constexpr int DELTA_RB = 64;

template <int NC>  // NC > 0: compile-time half width (the reference's N = 2), all 2 N loads independent; 0: runtime N
__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                    const int64_t* __restrict__ off, int n_seg, int dim, int N_rt,
                                                    float inv_den, int64_t row0, int64_t n_rows) {
    const int N = NC > 0 ? NC : N_rt;
    const int64_t r0 = (int64_t)blockIdx.x * DELTA_RB;
    const int nr = (int)min<int64_t>(DELTA_RB, n_rows - r0);
    const int base = find_segment(off, n_seg, row0 + r0);
    for (int i = threadIdx.x; i < nr * dim; i += 256) {
        const int r = i / dim, d = i - r * dim;
        const int64_t row = row0 + r0 + r;
        int u = base;
        while (off[u + 1] <= row) ++u;  // (empty utterances are skipped too)
        const int64_t a = off[u], b = off[u + 1] - 1;  // edge padding inside the utterance (GMM_UBM.py:64)
        // (the n = 0 term of GMM_UBM.py:68's numpy.dot: 0 . c[t] — nothing for a finite cepstrum, NaN for a non-finite one, which is how
        //  the reference's delta comes out non-finite AT a silent frame whose neighbours are not)
        float acc = 0.f * in[row * dim + d];
        if (NC > 0) {
            float vp[NC > 0 ? NC : 1], vm[NC > 0 ? NC : 1];
#pragma unroll
            for (int m = 1; m <= NC; ++m) {
                const int64_t rp = row + m > b ? b : row + m, rm = row - m < a ? a : row - m;
                vp[m - 1] = in[rp * dim + d];
                vm[m - 1] = in[rm * dim + d];
            }
#pragma unroll
            for (int m = 1; m <= NC; ++m) acc += (float)m * (vp[m - 1] - vm[m - 1]);
        } else {
            for (int m = 1; m <= N; ++m) {
                const int64_t rp = row + m > b ? b : row + m, rm = row - m < a ? a : row - m;
                acc += (float)m * (in[rp * dim + d] - in[rm * dim + d]);
            }
        }
        out[row * dim + d] = acc * inv_den;
    }
}

// one workgroup per utterance: mean, then variance about the mean (two passes), then normalise.  Thread t owns column t % dim and
// row phase t / dim, so a sweep of the workgroup reads whole consecutive rows (coalesced); utterances that fit the LDS budget are
// read from HBM once.
__global__ __launch_bounds__(256) void cmvn_kernel(const float* in, float* out,  // (in == out allowed)
                                                   const int64_t* __restrict__ off, int dim, int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [R][dim] partials
    const int u = blockIdx.x;
    const int64_t a = off[u];
    const int T = (int)(off[u + 1] - a);
    if (T == 0) return;
    const int tid = threadIdx.x;
    const int R = dim <= 256 ? 256 / dim : 1;     // row phases (dim > 256: threads stride over columns)
    float* mean = red + R * (dim <= 256 ? dim : 256);
    float* istd = mean + dim;
    float* xl = istd + dim;                        // [lds_rows x dim] the utterance, when it fits
    const float* x = in + a * dim;
    const bool in_lds = T <= lds_rows;
    const int tot = T * dim;
    if (in_lds) {
        for (int i0 = tid; i0 < tot; i0 += 256 * 8) {  // 8 loads in flight per thread
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = i0 + 256 * k < tot ? x[i0 + 256 * k] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + 256 * k < tot) xl[i0 + 256 * k] = v[k];
        }
        __syncthreads();
    }
    const float* src = in_lds ? xl : x;
    for (int c0 = 0; c0 < dim; c0 += 256) {  // one trip unless dim > 256
        const int c = c0 + (dim <= 256 ? tid % dim : tid), ph = dim <= 256 ? tid / dim : 0;
        const bool act = c < dim && ph < R;
        // sklearn.preprocessing.scale takes the statistics over the entries that are not NaN (nanmean / nanstd) and leaves the NaN
        // entries as they are: a digitally silent frame of a dialect without a log floor is such a row (GMM_UBM.py:89-93)
        const int rs = dim <= 256 ? dim : 256;
        float s = 0.f, cn = 0.f;
        if (act)
            for (int t = ph; t < T; t += R) {
                const float x = src[(size_t)t * dim + c];
                const bool ok = x == x;
                s += ok ? x : 0.f;
                cn += ok ? 1.f : 0.f;
            }
        if (act) red[ph * rs + (c - c0)] = cn;
        __syncthreads();
        float n_ok = 0.f;
        if (act)
            for (int k = 0; k < R; ++k) n_ok += red[k * rs + (c - c0)];
        __syncthreads();
        if (act) red[ph * rs + (c - c0)] = s;
        __syncthreads();
        if (act && ph == 0) {
            float tot_s = 0.f;
            for (int k = 0; k < R; ++k) tot_s += red[k * rs + (c - c0)];
            mean[c] = tot_s / n_ok;  // (no entry at all: NaN, as nanmean has it)
        }
        __syncthreads();
        float v = 0.f;
        if (act) {
            const float m = mean[c];
            for (int t = ph; t < T; t += R) {
                const float x = src[(size_t)t * dim + c];
                const float e = x == x ? x - m : 0.f;
                v = fmaf(e, e, v);
            }
            red[ph * rs + (c - c0)] = v;
        }
        __syncthreads();
        if (act && ph == 0) {
            float tot_v = 0.f;
            for (int k = 0; k < R; ++k) tot_v += red[k * rs + (c - c0)];
            float sd = sqrtf(tot_v / n_ok);
            if (sd < 10.0f * 1.1920929e-07f) sd = 1.0f;  // sk: _handle_zeros_in_scale
            istd[c] = 1.0f / sd;
        }
        __syncthreads();
    }
    float* y = out + a * dim;
    for (int i = tid; i < tot; i += 256) {
        const int d = i % dim;
        y[i] = (src[i] - mean[d]) * istd[d];
    }
}

static size_t cmvn_lds(int dim, int* lds_rows, int64_t max_T) {
    const size_t fixed = ((size_t)256 + 2 * (size_t)dim) * sizeof(float);  // partials (<= 256 floats), mean, inv std
    const size_t budget = 48 * 1024;                                       // three workgroups per CU
    int64_t rows = fixed < budget ? (int64_t)((budget - fixed) / ((size_t)dim * sizeof(float))) : 0;
    if (rows > max_T) rows = max_T;
    *lds_rows = (int)rows;
    return fixed + (size_t)rows * dim * sizeof(float);
}

int launch_cmvn(const float* in, float* out, const int64_t* frame_off_dev, int64_t n_utt, int dim, int64_t max_T, hipStream_t stream) {
    if (n_utt <= 0) return SSP_OK;
    if (n_utt > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "cmvn: too many utterances");
    int lds_rows = 0;
    const size_t lds = cmvn_lds(dim, &lds_rows, max_T);
    hipLaunchKernelGGL(cmvn_kernel, dim3((unsigned)n_utt), dim3(256), lds, stream, in, out, frame_off_dev, dim, lds_rows);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

// utils/processing.py:19-38 — framing + window; out is row-major (frame_size, n_frames) like the reference ndarray
__global__ __launch_bounds__(256) void enframe_kernel(const float* __restrict__ x, int64_t n, int frame_size, int step,
                                                      int64_t n_frames, const float* __restrict__ window,
                                                      float* __restrict__ out) {
    const int64_t total = (int64_t)frame_size * n_frames;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / n_frames, fr = i - k * n_frames;  // element (k, fr)
        const int64_t src = fr * step + k;
        out[i] = src < n ? x[src] * window[k] : 0.0f;
    }
}

// utils/processing.py:91-107 — one workgroup per spectrum row: filterbank dot products, log, DCT rows
__global__ __launch_bounds__(256) void cepstrum_kernel(const float* __restrict__ X, int n_bins, const float* __restrict__ fbank,
                                                       int n_filt, const float* __restrict__ dct, int n_ceps, int log_mode,
                                                       int floor_mode, float eps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lm = reinterpret_cast<float*>(smem);
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ xr = X + row * n_bins;
    for (int jf = wave; jf < n_filt; jf += 4) {
        const float* __restrict__ w = fbank + (size_t)jf * n_bins;
        float acc = 0.f;
        for (int k = lane; k < n_bins; k += 64) acc = fmaf(xr[k], w[k], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) {
            float v = acc;
            if (floor_mode == 1) v += eps;
            else if (floor_mode == 2) v = fmaxf(v, eps);
            lm[jf] = log_mode == 0 ? logf(v) : (log_mode == 1 ? log10f(v) : 10.0f * log10f(v));
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n_ceps; q += 256) {
        const float* __restrict__ d = dct + (size_t)q * n_filt;
        float acc = 0.f;
        for (int jf = 0; jf < n_filt; ++jf) acc = fmaf(lm[jf], d[jf], acc);
        out[row * n_ceps + q] = acc;
    }
}

}  // namespace ssp

using namespace ssp;

extern "C" {

int ssp_enframe(ssp_ctx* ctx, const float* samples, int64_t n, int32_t frame_size, int32_t step, const float* window,
                float* frames_out, int where, float* kernel_ms) {
    SSP_TRY(use_ctx(ctx));
    if (n < 0 || frame_size < 1 || step < 1 || !window) SSP_FAIL(SSP_ERR_INVALID, "ssp_enframe: bad argument");
    if (kernel_ms) *kernel_ms = 0.f;
    const int64_t n_frames = (n + step - 1) / step;  // math.ceil(wlen / step), utils/processing.py:27
    if (n_frames == 0) return SSP_OK;
    if (!samples || !frames_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_enframe: null data pointer");
    const size_t out_bytes = (size_t)frame_size * n_frames * sizeof(float);
    Staged sin, sout, sw;
    int rc;
    const float* d_in = (const float*)sin.in(ctx, samples, (size_t)n * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const float* d_w = (const float*)sw.in(ctx, window, (size_t)frame_size * sizeof(float), SSP_HOST, &rc);
    SSP_TRY(rc);
    float* d_out = (float*)sout.out(ctx, frames_out, out_bytes, where, &rc);
    SSP_TRY(rc);
    const int grid = (int)std::min<int64_t>(ceil_div<int64_t>((int64_t)frame_size * n_frames, 256), (int64_t)ctx->num_cu * 8);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, ctx->stream));
    hipLaunchKernelGGL(enframe_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_in, n, frame_size, step, n_frames, d_w, d_out);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(ctx->stream, kernel_ms));
    SSP_TRY(sout.back(ctx, frames_out, out_bytes, where));
    SSP_HIP(hipStreamSynchronize(ctx->stream));  // the staged window copy dies at return
    return SSP_OK;
}

int ssp_cepstrum(ssp_ctx* ctx, const float* X, int64_t n_rows, int32_t n_bins, const float* fbank, int32_t n_filt,
                 const float* dct, int32_t n_ceps, int32_t log_mode, int32_t floor_mode, float eps, float* out, int where,
                 float* kernel_ms) {
    SSP_TRY(use_ctx(ctx));
    if (n_rows < 0 || n_bins < 1 || n_filt < 1 || n_filt > 4096 || n_ceps < 1 || !fbank || !dct || log_mode < 0 || log_mode > 2 ||
        floor_mode < 0 || floor_mode > 2)
        SSP_FAIL(SSP_ERR_INVALID, "ssp_cepstrum: bad argument");
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_rows == 0) return SSP_OK;
    if (!X || !out) SSP_FAIL(SSP_ERR_INVALID, "ssp_cepstrum: null data pointer");
    if (n_rows > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_cepstrum: too many rows");
    Staged sx, so, sf, sd;
    int rc;
    const float* dX = (const float*)sx.in(ctx, X, (size_t)n_rows * n_bins * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const float* dF = (const float*)sf.in(ctx, fbank, (size_t)n_filt * n_bins * sizeof(float), SSP_HOST, &rc);
    SSP_TRY(rc);
    const float* dD = (const float*)sd.in(ctx, dct, (size_t)n_ceps * n_filt * sizeof(float), SSP_HOST, &rc);
    SSP_TRY(rc);
    float* dO = (float*)so.out(ctx, out, (size_t)n_rows * n_ceps * sizeof(float), where, &rc);
    SSP_TRY(rc);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, ctx->stream));
    hipLaunchKernelGGL(cepstrum_kernel, dim3((unsigned)n_rows), dim3(256), (size_t)n_filt * sizeof(float), ctx->stream, dX, n_bins, dF,
                       n_filt, dD, n_ceps, log_mode, floor_mode, eps, dO);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(ctx->stream, kernel_ms));
    SSP_TRY(so.back(ctx, out, (size_t)n_rows * n_ceps * sizeof(float), where));
    SSP_HIP(hipStreamSynchronize(ctx->stream));  // staged tables die at return
    return SSP_OK;
}

int ssp_delta(ssp_ctx* ctx, const float* feats, const ssp_segments* frame_seg, int32_t dim, int32_t N, float* out,
              int where, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_delta");
    SSP_TRY(use_ctx(ctx));
    if (!frame_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_delta: null segments");
    if (N < 1) SSP_FAIL(SSP_ERR_INVALID, "N must be an integer >= 1");  // GMM_UBM.py:59-60
    if (N > 64 || dim < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_delta: bad N or dim");
    if (kernel_ms) *kernel_ms = 0.f;
    const int64_t row0 = frame_seg->host.front(), rows = frame_seg->total();
    if (rows == 0) return SSP_OK;
    if (!feats || !out) SSP_FAIL(SSP_ERR_INVALID, "ssp_delta: null data pointer");
    const size_t bytes = (size_t)frame_seg->host.back() * dim * sizeof(float);
    Staged sin, sout;
    int rc;
    const float* d_in = (const float*)sin.in(ctx, feats, bytes, where, &rc);
    SSP_TRY(rc);
    float* d_out = (float*)sout.out(ctx, out, bytes, where, &rc);
    SSP_TRY(rc);
    int den = 0;
    for (int i = 1; i <= N; ++i) den += 2 * i * i;
    const int64_t n_blocks = ceil_div<int64_t>(rows, DELTA_RB);
    if (n_blocks > INT32_MAX || (int64_t)DELTA_RB * dim > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_delta: matrix too large for one launch");
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, ctx->stream));
    if (N == 2)
        hipLaunchKernelGGL(delta_kernel<2>, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, d_in, d_out, frame_seg->dev.as<int64_t>(),
                           (int)frame_seg->n, dim, N, 1.0f / (float)den, row0, rows);
    else
        hipLaunchKernelGGL(delta_kernel<0>, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, d_in, d_out, frame_seg->dev.as<int64_t>(),
                           (int)frame_seg->n, dim, N, 1.0f / (float)den, row0, rows);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(ctx->stream, kernel_ms));
    SSP_TRY(sout.back(ctx, out, bytes, where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(ctx->stream));
    return SSP_OK;
}

int ssp_cmvn(ssp_ctx* ctx, const float* feats, const ssp_segments* frame_seg, int32_t dim, float* out, int where,
             float* kernel_ms) {
    ssp::TraceRange trace_("ssp_cmvn");
    SSP_TRY(use_ctx(ctx));
    if (!frame_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_cmvn: null segments");
    if (dim < 1 || dim > 4096) SSP_FAIL(SSP_ERR_INVALID, "ssp_cmvn: dim out of range");
    if (kernel_ms) *kernel_ms = 0.f;
    if (frame_seg->total() == 0 || frame_seg->n == 0) return SSP_OK;
    if (!feats || !out) SSP_FAIL(SSP_ERR_INVALID, "ssp_cmvn: null data pointer");
    if (frame_seg->max_len() * dim > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_cmvn: utterance too long");
    const size_t bytes = (size_t)frame_seg->host.back() * dim * sizeof(float);
    Staged sin, sout;
    int rc;
    const float* d_in = (const float*)sin.in(ctx, feats, bytes, where, &rc);
    SSP_TRY(rc);
    float* d_out = (float*)sout.out(ctx, out, bytes, where, &rc);
    SSP_TRY(rc);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, ctx->stream));
    SSP_TRY(launch_cmvn(d_in, d_out, frame_seg->dev.as<int64_t>(), frame_seg->n, dim, frame_seg->max_len(), ctx->stream));
    SSP_TRY(tm.stop(ctx->stream, kernel_ms));
    SSP_TRY(sout.back(ctx, out, bytes, where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(ctx->stream));
    return SSP_OK;
}

}  // extern "C"

namespace ssp {
// |re + i im| (power 1) or re^2 + im^2 (power 2), scaled: rows of [re(0..nb) | im(0..nb)] -> rows of nb bins
__global__ void spectrum_abs_kernel(const float* __restrict__ reim, int64_t n, int nb, float scale, int power, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / nb;
    const int k = (int)(i - r * nb);
    const float re = reim[r * 2 * nb + k], im = reim[r * 2 * nb + nb + k];
    const float p = re * re + im * im;
    out[i] = scale * (power == 2 ? p : sqrtf(p));
}
}  // namespace ssp

extern "C" int ssp_spectrum_abs(ssp_ctx* ctx, const float* reim, int64_t n_rows, int32_t n_bins, float scale, int32_t power,
                                float* out, int where, float* kernel_ms) {
    using namespace ssp;
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_rows < 0 || n_bins < 1 || (power != 1 && power != 2)) SSP_FAIL(SSP_ERR_INVALID, "ssp_spectrum_abs: bad argument");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_spectrum_abs: where");
    if (n_rows == 0) return SSP_OK;
    if (!reim || !out) SSP_FAIL(SSP_ERR_INVALID, "ssp_spectrum_abs: null data pointer");
    const int64_t n = n_rows * n_bins;
    if ((n + 255) / 256 > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_spectrum_abs: too many elements");
    Staged si, so;
    int rc;
    const float* dI = (const float*)si.in(ctx, reim, (size_t)n * 2 * sizeof(float), where, &rc);
    SSP_TRY(rc);
    float* dO = (float*)so.out(ctx, out, (size_t)n * sizeof(float), where, &rc);
    SSP_TRY(rc);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, ctx->stream));
    hipLaunchKernelGGL(spectrum_abs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dI, n, n_bins, scale, power, dO);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(ctx->stream, kernel_ms));
    SSP_TRY(so.back(ctx, out, (size_t)n * sizeof(float), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(ctx->stream));
    return SSP_OK;
}

// EM sufficient statistics of a diagonal-covariance GMM on gfx950 — the O(frames x K x D) part of one iteration of
// sklearn's GaussianMixture.fit as the reference runs it (GMM_UBM.py:158-170: GaussianMixture(n_components, 'diag').fit):
//   E step   log_resp[t,k] = log w_k + log N(x_t; mu_k, diag cov_k) - logsumexp_k(...)      sk:mixture/_base.py:_e_step
//   M sums   nk[k] = sum_t resp[t,k],  sx[k,d] = sum_t resp[t,k] x[t,d],  sxx[k,d] = sum_t resp[t,k] x[t,d]^2
//            (the resp.T @ X and resp.T @ X*X products of _estimate_gaussian_parameters / _estimate_gaussian_covariances_diag)
// The O(K x D) closing arithmetic of the M step (means, covariances + reg_covar, weights, convergence test) stays on the
// host in float64 exactly as sklearn writes it (speech_signal_processing_amd/gmm_train.py).
//
// Kernels per call (D > 47; the MFMA path for D <= 47 is described at gmm_em_acc_mfma_kernel):
//   gmm_em_lse_kernel    workgroup = 64 frames: lp[t,k] for every mixture in 64-mixture chunks staged in LDS, online
//                        log-sum-exp -> lse[t] and one partial of sum_t lse[t] per workgroup
//   gmm_em_acc_kernel    grid (G, K/64): the workgroup keeps ONE 64-mixture chunk of parameters in LDS and walks its frame
//                        tiles; per tile  resp = exp(lp - lse)  (64 x 64, LDS)  then  thread (k, dim group) accumulates
//                        nk / sx / sxx in registers over all its tiles; one fp32 partial per (workgroup, mixture, column)
//   gmm_em_reduce_kernel partials -> float64 sums in a fixed order (bit-reproducible)
// Parameters enter as A = mu P, B = -P/2, c = ln w + 1/2 sum ln P - 1/2 sum mu^2 P - D/2 ln 2pi (float64 on the host, fp32
// on the device): lp = c + sum_d x_d (A_d + x_d B_d).
#include <cmath>

#include "common.hpp"

namespace ssp {

constexpr int EM_TF = 64;   // frames per tile
constexpr int EM_KC = 64;   // mixtures per chunk
constexpr int EM_DG = 16;   // max dims per accumulation group (4 groups: D <= 64)

struct EmArgs {
    const float* x;       // [n x D]
    const float* par;     // [Kp][2D+1]  (A[0..D), B[0..D), c), Kp = K rounded up to a multiple of 64, padded c = -1e30
    float* lse;           // [n]
    float* lse_part;      // [ceil(n / 64)]
    float* part;          // [G][Kp][2D+1]
    int64_t n;
    int32_t D, K, Kp, G, n_tiles;
};

// lp of frame `xs` (LDS row, D floats) under mixture row `w` (LDS, 2D+1 floats; wave-uniform address: broadcast reads)
__device__ __forceinline__ float em_lp(const float* __restrict__ xs, const float* __restrict__ w, int D) {
    float acc = w[2 * D];
    for (int d = 0; d < D; ++d) {
        const float xv = xs[d];
        acc = fmaf(xv, fmaf(xv, w[D + d], w[d]), acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void gmm_em_lse_kernel(EmArgs a) {
    extern __shared__ float sm[];
    const int D = a.D, W = 2 * D + 1, XS = D | 1;  // odd row stride: the 64 frames of a wave read distinct banks
    float* xs = sm;                   // [64][XS]
    float* ws = xs + EM_TF * XS;      // [64][W]
    float* red = ws + EM_KC * W;      // [4][64] x 2
    const int tid = threadIdx.x, t = tid & 63, kg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t base = (int64_t)blockIdx.x * EM_TF;
    const int nt = (int)min<int64_t>(EM_TF, a.n - base);
    for (int i = tid; i < EM_TF * D; i += 256) {
        const int r = i / D, c = i - r * D;
        xs[r * XS + c] = r < nt ? a.x[(base + r) * D + c] : 0.f;
    }
    float m = -INFINITY, s = 0.f;
    for (int kc = 0; kc < a.Kp; kc += EM_KC) {
        __syncthreads();
        for (int i = tid; i < EM_KC * W; i += 256) ws[i] = a.par[(size_t)kc * W + i];
        __syncthreads();
        for (int kk = 0; kk < 16; ++kk) {
            const float lp = em_lp(xs + t * XS, ws + (kg * 16 + kk) * W, D);
            const float mn = fmaxf(m, lp);
            s = s * __expf(m - mn) + __expf(lp - mn);
            m = mn;
        }
    }
    red[kg * 64 + t] = m;
    red[256 + kg * 64 + t] = s;
    __syncthreads();
    if (tid < 64) {
        float mm = red[t], ss = red[256 + t];
        for (int g = 1; g < 4; ++g) {
            const float m2 = red[g * 64 + t], s2 = red[256 + g * 64 + t];
            const float mn = fmaxf(mm, m2);
            ss = ss * __expf(mm - mn) + s2 * __expf(m2 - mn);
            mm = mn;
        }
        float l = t < nt ? mm + __logf(ss) : 0.f;
        if (t < nt) a.lse[base + t] = l;
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        if (t == 0) a.lse_part[blockIdx.x] = l;
    }
}

__global__ __launch_bounds__(256) void gmm_em_acc_kernel(EmArgs a) {
    extern __shared__ float sm[];
    const int D = a.D, W = 2 * D + 1, XS = D | 1;
    float* xs = sm;                   // [64][XS]
    float* ws = xs + EM_TF * XS;      // [64][W]    this workgroup's chunk of mixtures, loaded once
    float* rs = ws + EM_KC * W;       // [64 frames][65]  responsibilities of the tile
    const int tid = threadIdx.x, t = tid & 63, kg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kc = blockIdx.y * EM_KC;
    for (int i = tid; i < EM_KC * W; i += 256) ws[i] = a.par[(size_t)kc * W + i];
    // accumulation role: mixture k = tid & 63 of the chunk, dims [dg * dper, dg * dper + dper)
    const int dper = (D + 3) / 4, d0 = kg * dper;
    float ax[EM_DG], axx[EM_DG], an = 0.f;
#pragma unroll
    for (int i = 0; i < EM_DG; ++i) ax[i] = axx[i] = 0.f;
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += a.G) {
        const int64_t base = (int64_t)tile * EM_TF;
        const int nt = (int)min<int64_t>(EM_TF, a.n - base);
        __syncthreads();
        for (int i = tid; i < EM_TF * D; i += 256) {
            const int r = i / D, c = i - r * D;
            xs[r * XS + c] = r < nt ? a.x[(base + r) * D + c] : 0.f;
        }
        __syncthreads();
        const float l = t < nt ? a.lse[base + t] : INFINITY;  // frames beyond the end get resp = exp(-inf) = 0
        for (int kk = 0; kk < 16; ++kk) {
            const int k = kg * 16 + kk;
            rs[t * 65 + k] = __expf(em_lp(xs + t * XS, ws + k * W, D) - l);
        }
        __syncthreads();
        for (int r = 0; r < EM_TF; ++r) {
            const float rv = rs[r * 65 + t];  // mixture t of the chunk, frame r
            if (kg == 0) an += rv;
            const float* xr = xs + r * XS + d0;  // wave-uniform: broadcast
#pragma unroll
            for (int i = 0; i < EM_DG; ++i)
                if (i < dper && d0 + i < D) {
                    const float xv = xr[i], rx = rv * xv;
                    ax[i] += rx;
                    axx[i] = fmaf(rx, xv, axx[i]);
                }
        }
    }
    float* out = a.part + ((size_t)blockIdx.x * a.Kp + kc + t) * W;
#pragma unroll
    for (int i = 0; i < EM_DG; ++i)
        if (i < dper && d0 + i < D) {
            out[d0 + i] = ax[i];
            out[D + d0 + i] = axx[i];
        }
    if (kg == 0) out[2 * D] = an;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

// sum of lse over 64-frame tiles (one partial per tile, like gmm_em_lse_kernel writes)
__global__ __launch_bounds__(64) void gmm_em_lsesum_kernel(const float* __restrict__ lse, int64_t n, float* __restrict__ part) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    float v = i < n ? lse[i] : 0.f;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (threadIdx.x == 0) part[blockIdx.x] = v;
}

// MFMA form of gmm_em_acc_kernel (D <= 47: at most 3 column tiles of [x, x^2, 1]).  Grid (G, K/64), 4 waves; per 64-frame tile
//   GEMM1  lp[64 mix x 64 frames] = W[64 x (2D+1)] . aug^T        wave (r, c) owns one 32 x 32 tile; mixtures = MFMA rows, frames =
//                                                                  columns, so a lane's 16 accumulators belong to ONE frame
//   resp   = exp(lp - lse[frame])  -> LDS tile rs[frame][mix]
//   GEMM2  S[64 mix x (2D+1)] += resp[64 mix x 64 frames] . aug    split over the waves by frames (wave w: frames 16w..16w+15 of
//                                                                  the tile); accumulators stay in registers over all tiles
// Each wave leaves its own partial [64 mix][2D+1]; gmm_em_reduce_kernel adds the 4 G partials in float64.
// FUSE (K <= 64: the workgroup sees every mixture of a frame): the per-frame log-sum-exp is taken right here from the GEMM1
// accumulators (in-lane over 16 mixtures, across the two half-waves, across the two waves that share a frame through LDS) instead of a
// separate scoring pass; sum_t lse[t] leaves as one partial per workgroup.
template <int NCT, bool FUSE>
__global__ __launch_bounds__(256) void gmm_em_acc_mfma_kernel(EmArgs a) {
    extern __shared__ float sm[];
    const int D = a.D, W = 2 * D + 1, KS = (W + 1) / 2;  // KS k-steps of 2 over [x, x^2, 1] (+ a zero pad column)
    const int XS = (2 * KS) | 1;       // odd row stride of the augmented frame tile
    float* xs = sm;                    // [64 frames][XS]  aug = [x, x^2, 1, 0]: the B operand of both GEMMs, read as it stands (forming
                                       // x^2 / the constants per MFMA put branches and a dependent LDS read into the inner loops)
    float* wsT = xs + EM_TF * XS;      // [2 KS][64 mix]   parameter chunk, k-major: the A operand of GEMM1
    float* rs = wsT + 2 * KS * EM_KC;  // [64 frames][65]  responsibilities: the A operand of GEMM2
    float* ex = rs + EM_TF * 65;       // [2 mixture halves][64 frames][2]  (max, sum) exchange of the fused log-sum-exp
    float lsum = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, h = lane >> 5;
    const int kc = blockIdx.y * EM_KC;
    for (int i = tid; i < 2 * KS * EM_KC; i += 256) {
        const int k = i / EM_KC, m = i - k * EM_KC;
        wsT[i] = k < W ? a.par[(size_t)(kc + m) * W + k] : 0.f;
    }
    for (int i = tid; i < EM_TF * (2 * KS - 2 * D); i += 256) {  // the constant columns [1, 0...] of every row, written once
        const int r = i / (2 * KS - 2 * D), c = i - r * (2 * KS - 2 * D);
        xs[r * XS + 2 * D + c] = c == 0 ? 1.f : 0.f;
    }
    const int r1 = wave >> 1, c1 = wave & 1;  // GEMM1 tile of this wave
    f32x16 acc2[2][NCT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[r][c][i] = 0.f;
    // the frame tile of the NEXT iteration is fetched into registers while this one's MFMAs run (a synchronous load would expose a
    // full memory latency per 64-frame tile): EM_XP floats per thread cover 64 x D <= 64 x 47
    constexpr int EM_XP = 12;
    float xpre[EM_XP];
    float lpre = INFINITY;
    auto fetch = [&](int tile) {
        const int64_t base = (int64_t)tile * EM_TF;
        const int nt = tile < a.n_tiles ? (int)min<int64_t>(EM_TF, a.n - base) : 0;
#pragma unroll
        for (int u = 0; u < EM_XP; ++u) {
            const int i = tid + u * 256;
            const int r = i / D;
            xpre[u] = (i < EM_TF * D && r < nt) ? a.x[base * D + i] : 0.f;
        }
        const int fr = 32 * c1 + fl;
        if (FUSE) lpre = fr < nt ? 0.f : INFINITY;
        else lpre = fr < nt ? a.lse[base + fr] : INFINITY;  // frames beyond the end: resp = exp(-inf) = 0
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += a.G) {
        __syncthreads();  // the previous tile's GEMM2 is done with xs / rs
#pragma unroll
        for (int u = 0; u < EM_XP; ++u) {
            const int i = tid + u * 256;
            if (i < EM_TF * D) {
                const int r = i / D, c = i - r * D;
                const float v = xpre[u];
                xs[r * XS + c] = v;
                xs[r * XS + D + c] = v * v;
            }
        }
        const float l = lpre;
        fetch(tile + a.G);
        __syncthreads();
        // ---- GEMM1: A[row = mix 32 r1 + fl][k = 2 s + h], B[k = 2 s + h][col = frame 32 c1 + fl] = aug[frame][k]
        f32x16 acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
        const float* xrow = xs + (32 * c1 + fl) * XS;
        const float* wcol = wsT + h * EM_KC + 32 * r1 + fl;
        int s2 = 0;
        for (; s2 + 3 < KS; s2 += 4) {  // four k-steps' operands in flight
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = wcol[(2 * (s2 + u)) * EM_KC];
                bv[u] = xrow[2 * (s2 + u) + h];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc1, 0, 0, 0);
        }
        for (; s2 < KS; ++s2)
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[(2 * s2) * EM_KC], xrow[2 * s2 + h], acc1, 0, 0, 0);
        // ---- responsibilities of this lane's frame; accumulator i = mixture 32 r1 + (i & 3) + 8 (i >> 2) + 4 h
        const int fr = 32 * c1 + fl;
        if (FUSE) {
            float m = acc1[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, acc1[i]);
            m = fmaxf(m, __shfl_xor(m, 32));
            float e[16], ssum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                e[i] = __expf(acc1[i] - m);
                ssum += e[i];
            }
            ssum += __shfl_xor(ssum, 32);
            if (h == 0) *reinterpret_cast<float2*>(ex + (r1 * 64 + fr) * 2) = make_float2(m, ssum);
            __syncthreads();
            const float2 o = *reinterpret_cast<const float2*>(ex + ((1 - r1) * 64 + fr) * 2);
            const float mm = fmaxf(m, o.x);
            const float tot = ssum * __expf(m - mm) + o.y * __expf(o.x - mm);
            const bool valid = l == 0.f;
            const float scale = valid ? __expf(m - mm) / tot : 0.f;
            if (valid && r1 == 0 && h == 0) lsum += mm + __logf(tot);
#pragma unroll
            for (int i = 0; i < 16; ++i) rs[fr * 65 + 32 * r1 + (i & 3) + 8 * (i >> 2) + 4 * h] = e[i] * scale;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) rs[fr * 65 + 32 * r1 + (i & 3) + 8 * (i >> 2) + 4 * h] = __expf(acc1[i] - l);
        }
        __syncthreads();
        // ---- GEMM2 over this wave's 16 frames: A[row = mix 32 r + fl][k = frame], B[k = frame][col = 32 c + fl] = aug[frame][col]
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            const int f2 = 16 * wave + 2 * s2 + h;
            float av[2], bv[NCT];
#pragma unroll
            for (int r = 0; r < 2; ++r) av[r] = rs[f2 * 65 + 32 * r + fl];
            const float* xr = xs + f2 * XS;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int col = 32 * c + fl;
                bv[c] = col < 2 * KS ? xr[col] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc2[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bv[c], acc2[r][c], 0, 0, 0);
        }
    }
    if (FUSE) {  // sum of this workgroup's lse values: lanes (r1 == 0, h == 0) of waves 0 and 1 hold them
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
        __syncthreads();
        if (lane == 0 && r1 == 0) ex[c1] = lsum;
        __syncthreads();
        if (tid == 0) a.lse_part[blockIdx.x] = ex[0] + ex[1];
    }
    float* out = a.part + ((size_t)(blockIdx.x * 4 + wave) * a.Kp + kc) * W;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int col = 32 * c + fl;
            if (col < W)
#pragma unroll
                for (int i = 0; i < 16; ++i) out[(size_t)(32 * r + (i & 3) + 8 * (i >> 2) + 4 * h) * W + col] = acc2[r][c][i];
        }
}

// out[j] = sum_g part[g][j] (float64, fixed order): block = 32 columns x 8 slices of the partial index (coalesced 128-byte
// reads), the 8 slice sums are added in slice order;  block 0 also reduces the lse partials
__global__ __launch_bounds__(256) void gmm_em_reduce_kernel(const float* part, int G, int64_t cols, const float* lse_part, int64_t n_lse,
                                                            double* out) {
    __shared__ double sh[256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t j = (int64_t)blockIdx.x * 32 + tx;
    double s = 0.0;
    if (j < cols)
        for (int g = ty; g < G; g += 8) s += (double)part[(size_t)g * cols + j];
    sh[ty * 32 + tx] = s;
    __syncthreads();
    if (ty == 0 && j < cols) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += sh[k * 32 + tx];
        out[j] = t;
    }
    if (blockIdx.x == 0) {  // sum_t lse[t]: fixed strided order + tree
        __syncthreads();
        double v = 0.0;
        for (int64_t i = threadIdx.x; i < n_lse; i += 256) v += (double)lse_part[i];
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[cols] = sh[0];
    }
}

}  // namespace ssp

using namespace ssp;

extern "C" int ssp_gmm_em_stats(ssp_ctx* ctx, int32_t K, int32_t D, const double* weights, const double* means,
                                const double* covars, const float* feats, int64_t n_frames, double* nk_out, double* sx_out,
                                double* sxx_out, double* loglik_sum_out, int where, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_gmm_em_stats");
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (K < 1 || D < 1 || !weights || !means || !covars || !nk_out || !sx_out || !sxx_out || !loglik_sum_out)
        SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_em_stats: bad shape or null array");
    if (D > 4 * EM_DG) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_gmm_em_stats: D=%d exceeds the supported feature dimension (%d)", D, 4 * EM_DG);
    if (n_frames < 1 || !feats) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_em_stats: no frames");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_em_stats: where");
    const int W = 2 * D + 1, Kp = (K + EM_KC - 1) / EM_KC * EM_KC;
    std::vector<float> par((size_t)Kp * W, 0.f);
    const double ln2pi = std::log(2.0 * M_PI);
    for (int k = 0; k < Kp; ++k) {
        float* w = par.data() + (size_t)k * W;
        if (k >= K) {
            w[2 * D] = -1.0e30f;  // padded mixture: resp = exp(-1e30 - lse) = 0
            continue;
        }
        if (!(weights[k] > 0.0)) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_em_stats: non-positive weight (mix %d)", k);
        double c = std::log(weights[k]) - 0.5 * D * ln2pi;
        for (int d = 0; d < D; ++d) {
            const double cv = covars[(size_t)k * D + d], mu = means[(size_t)k * D + d];
            if (!(cv > 0.0)) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_em_stats: non-positive covariance (mix %d)", k);
            const double P = 1.0 / cv;
            w[d] = (float)(mu * P);
            w[D + d] = (float)(-0.5 * P);
            c += 0.5 * std::log(P) - 0.5 * mu * mu * P;
        }
        w[2 * D] = (float)c;
    }
    hipStream_t s = ctx->stream;
    const int64_t n_tiles = (n_frames + EM_TF - 1) / EM_TF;
    if (n_tiles > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_gmm_em_stats: too many frames");
    const int G = (int)std::min<int64_t>(n_tiles, 2 * (int64_t)ctx->num_cu);
    const int64_t cols = (int64_t)Kp * W;
    const int nct = (W + 31) / 32;
    const bool mfma = nct <= 3 && !getenv("SSP_EM_NO_MFMA");  // D <= 47: both GEMM-shaped products on the matrix cores
    const int GP = mfma ? 4 * G : G;                           // partials: one per wave on the MFMA path
    // device scratch lives in the ctx (grow-only): an EM loop calls this once per iteration
    DevBuf &d_par = ctx->scratch[0], &d_lse = ctx->scratch[1], &d_lsep = ctx->scratch[2], &d_part = ctx->scratch[3],
           &d_out = ctx->scratch[4];
    Staged sx;
    int rc;
    const float* d_x = (const float*)sx.in(ctx, feats, (size_t)n_frames * D * sizeof(float), where, &rc);
    SSP_TRY(rc);
    SSP_TRY(d_par.reserve(par.size() * sizeof(float)));
    SSP_TRY(d_lse.reserve((size_t)n_frames * sizeof(float)));
    SSP_TRY(d_lsep.reserve((size_t)std::max<int64_t>(n_tiles, G) * sizeof(float)));
    SSP_TRY(d_part.reserve((size_t)GP * cols * sizeof(float)));
    SSP_TRY(d_out.reserve((size_t)(cols + 1) * sizeof(double)));
    SSP_HIP(hipMemcpyAsync(d_par.p, par.data(), par.size() * sizeof(float), hipMemcpyHostToDevice, s));
    SSP_HIP(hipMemsetAsync(d_part.p, 0, (size_t)GP * cols * sizeof(float), s));
    EmArgs a{d_x, d_par.as<float>(), d_lse.as<float>(), d_lsep.as<float>(), d_part.as<float>(), n_frames, D, K, Kp, G, (int32_t)n_tiles};
    const int XS = D | 1;
    const size_t lds1 = ((size_t)EM_TF * XS + (size_t)EM_KC * W + 512) * sizeof(float);
    const size_t lds2 = ((size_t)EM_TF * XS + (size_t)EM_KC * W + (size_t)EM_TF * 65) * sizeof(float);
    const size_t lds3 = ((size_t)EM_TF * ((W + 1) | 1) + (size_t)(W + 1) * EM_KC + (size_t)EM_TF * 65 + 256) * sizeof(float);
    const bool fuse = mfma && Kp == EM_KC && !getenv("SSP_EM_NO_FUSE");  // K <= 64: log-sum-exp inside the accumulation kernel
    if (lds1 > 64 * 1024)
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_em_lse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    if (lds2 > 64 * 1024)
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_em_acc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    if (mfma && lds3 > 64 * 1024) {
        const void* ks[6] = {reinterpret_cast<const void*>(gmm_em_acc_mfma_kernel<1, true>), reinterpret_cast<const void*>(gmm_em_acc_mfma_kernel<2, true>),
                             reinterpret_cast<const void*>(gmm_em_acc_mfma_kernel<3, true>), reinterpret_cast<const void*>(gmm_em_acc_mfma_kernel<1, false>),
                             reinterpret_cast<const void*>(gmm_em_acc_mfma_kernel<2, false>), reinterpret_cast<const void*>(gmm_em_acc_mfma_kernel<3, false>)};
        for (const void* k : ks) SSP_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    }
    // the per-frame log-sum-exp of the MFMA path comes from the scoring kernel (csrc/gmm.hip: fp32 MFMA, score_samples of ONE model)
    ssp_gmm* scorer = nullptr;
    ssp_segments* seg = nullptr;
    if (mfma && !fuse) {
        const int64_t off[2] = {0, n_frames};
        rc = ssp_gmm_pack(ctx, 1, K, D, weights, means, covars, 0, &scorer);
        if (rc == SSP_OK) rc = segments_make(ctx, off, 1, &seg);
        if (rc != SSP_OK) {
            ssp_gmm_destroy(scorer);
            return rc;
        }
    }
    Timer tm;
    rc = tm.start(kernel_ms != nullptr, s);
    if (rc == SSP_OK && fuse) {
        switch (nct) {
            case 1: hipLaunchKernelGGL((gmm_em_acc_mfma_kernel<1, true>), dim3(G, 1), dim3(256), lds3, s, a); break;
            case 2: hipLaunchKernelGGL((gmm_em_acc_mfma_kernel<2, true>), dim3(G, 1), dim3(256), lds3, s, a); break;
            default: hipLaunchKernelGGL((gmm_em_acc_mfma_kernel<3, true>), dim3(G, 1), dim3(256), lds3, s, a); break;
        }
    } else if (rc == SSP_OK && mfma) {
        rc = ssp_gmm_score(scorer, d_x, seg, d_lse.as<float>(), nullptr, nullptr, SSP_DEVICE, 0, nullptr);
        if (rc == SSP_OK) {
            hipLaunchKernelGGL(gmm_em_lsesum_kernel, dim3((unsigned)n_tiles), dim3(64), 0, s, d_lse.as<float>(), n_frames, d_lsep.as<float>());
            switch (nct) {
                case 1: hipLaunchKernelGGL((gmm_em_acc_mfma_kernel<1, false>), dim3(G, Kp / EM_KC), dim3(256), lds3, s, a); break;
                case 2: hipLaunchKernelGGL((gmm_em_acc_mfma_kernel<2, false>), dim3(G, Kp / EM_KC), dim3(256), lds3, s, a); break;
                default: hipLaunchKernelGGL((gmm_em_acc_mfma_kernel<3, false>), dim3(G, Kp / EM_KC), dim3(256), lds3, s, a); break;
            }
        }
    } else if (rc == SSP_OK) {
        hipLaunchKernelGGL(gmm_em_lse_kernel, dim3((unsigned)n_tiles), dim3(256), lds1, s, a);
        hipLaunchKernelGGL(gmm_em_acc_kernel, dim3(G, Kp / EM_KC), dim3(256), lds2, s, a);
    }
    if (rc == SSP_OK) {
        hipLaunchKernelGGL(gmm_em_reduce_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, s, d_part.as<float>(), GP, cols,
                           d_lsep.as<float>(), fuse ? (int64_t)G : n_tiles, d_out.as<double>());
        if (hipGetLastError() != hipSuccess) {
            set_error("ssp_gmm_em_stats: kernel launch failed");
            rc = SSP_ERR_HIP;
        }
    }
    if (rc == SSP_OK) rc = tm.stop(s, kernel_ms);
    if (rc != SSP_OK) {
        (void)hipStreamSynchronize(s);
        ssp_gmm_destroy(scorer);
        ssp_segments_destroy(seg);
        return rc;
    }
    std::vector<double> host((size_t)cols + 1);
    hipError_t he = hipMemcpyAsync(host.data(), d_out.p, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    ssp_gmm_destroy(scorer);
    ssp_segments_destroy(seg);
    if (he != hipSuccess) SSP_FAIL(SSP_ERR_HIP, "ssp_gmm_em_stats: result copy failed: %s", hipGetErrorString(he));
    for (int k = 0; k < K; ++k) {
        const double* r = host.data() + (size_t)k * W;
        for (int d = 0; d < D; ++d) {
            sx_out[(size_t)k * D + d] = r[d];
            sxx_out[(size_t)k * D + d] = r[D + d];
        }
        nk_out[k] = r[2 * D];
    }
    *loglik_sum_out = host[(size_t)cols];
    return SSP_OK;
}

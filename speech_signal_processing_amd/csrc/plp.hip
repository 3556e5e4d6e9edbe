// PLP back end for gfx950: RASTA filtering of the log critical-band energies, equal-loudness weighting + cube-root compression,
// autocorrelation, Levinson-Durbin, LPC -> cepstrum, lifter.  The front end (framing, pre-emphasis, window, FFT, power spectrum,
// Bark filterbank, ln) is the fused MFCC pass with a Bark table and an identity DCT (frontend.preset_sidekit_plp).
//
// Replaces sidekit.frontend.features.plp at the reference's call sites GMM_UBM.py:95, d_vector.py:93, UI/GMM_UBM_GUI.py:93,
// UI/tmp.py:315-318 (sidekit's source is absent: the algorithm is the published rastamat one that sidekit ports, see
// oracle/ref_cpu.py "PLP" — parity unpinned).
//
//   plp_rasta_kernel   thread = (utterance, band): the recursion y[t] = sum_i b_i x[t-i] + 0.94 y[t-1] along time, transposed
//                      direct form II as scipy / Matlab run it; first four outputs zero, the FIR part alone primes the state
//   plp_cep_kernel     thread = frame: everything after the filter stays in registers (NB bands, P + 1 autocorrelation lags,
//                      P LPC coefficients, P + 1 cepstra); the cosine table of the real IDFT and the equal-loudness curve come
//                      from LDS
#include <cmath>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace ssp {

constexpr int PLP_NB_MAX = 40;  // bands: ceil(hz2bark(fs / 2)) + 1 = 21 at 16 kHz, 27 at 44.1 kHz
constexpr int PLP_P_MAX = 24;   // LPC order (sidekit: plp_order - 1 = 12)

__global__ __launch_bounds__(256) void plp_rasta_kernel(const float* __restrict__ x, float* __restrict__ y, const int64_t* __restrict__ off,
                                                        int64_t n_utt, int nb) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_utt * nb) return;
    const int64_t u = idx / nb;
    const int b = (int)(idx - u * nb);
    const int64_t f0 = off[u], T = off[u + 1] - f0;
    const float* __restrict__ xp = x + f0 * nb + b;
    float* __restrict__ yp = y + f0 * nb + b;
    float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
    // RB inputs per trip, the next trip's already in flight: the recursion is serial along time, and with one load ahead every step
    // waited out most of a memory latency (1.64 ms for 100k x 298 frames x 21 bands; the traffic alone is 0.6 ms)
    constexpr int RB = 8;
    float cur[RB], nxt[RB];
    if (T <= 0) return;
    // (loads past the utterance's end re-read its last frame, unconditionally: a load under a per-lane condition sits in its own
    //  branch with a wait behind it; the values are never stored)
#pragma unroll
    for (int i = 0; i < RB; ++i) cur[i] = xp[min((int64_t)i, T - 1) * nb];
    for (int64_t t = 0; t < T; t += RB) {
#pragma unroll
        for (int i = 0; i < RB; ++i) nxt[i] = xp[min(t + RB + i, T - 1) * nb];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const float xv = cur[i];
            const float a1 = t + i < 4 ? 0.f : -0.94f;
            const float out = fmaf(0.2f, xv, z0);
            z0 = fmaf(0.1f, xv, z1) - a1 * out;
            z1 = z2;  // b2 = 0
            z2 = fmaf(-0.1f, xv, z3);
            z3 = -0.2f * xv;
            if (t + i < T) yp[(t + i) * nb] = t + i < 4 ? 0.f : out;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) cur[i] = nxt[i];
    }
}

struct PlpArgs {
    const float* y;    // [F x nb]  ln critical-band energies (RASTA filtered or not)
    float* out;        // [F x (P + 1)]
    const float* tab;  // [nb] equal-loudness, [(P + 1) x nb] autocorrelation weights, [P + 1] lifter
    int64_t n_frames;
    int32_t nb, P;
};

// Tables of the fixed-size instances, passed BY VALUE in the kernel argument segment: every weight is then a scalar load into an SGPR
// and rides as the scalar operand of its FMA (from LDS the 273 weights of the 21-band / 12th-order autocorrelation were 273 LDS
// reads per frame, every thread fetching the same address).
template <int NB, int PP>
struct PlpTab {
    float lq[NB];              // 0.33 log2(equal loudness) (-inf where the curve is 0)
    float cw[(PP + 1) * NB];   // cw[k * NB + n]: weight of band n in lag k
    float lw[PP + 1];          // lifter
};

// a / b with a hardware reciprocal and one Newton step (the correctly rounded expansion is ~10 instructions and a branchy scale fix-up)
__device__ __forceinline__ float div_nr(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    return a * r;
}

// everything after the RASTA filter for one frame, in registers.  z = (exp(y) eql)^0.33 = exp2(0.33 log2(e) y + 0.33 log2(eql)): one
// FMA and one v_exp_f32 per band (powf(expf()) is ~60 instructions)
template <int NBM, int PM, class EQ, class CW, class LW>
__device__ __forceinline__ void plp_frame(const float* __restrict__ yp, float* __restrict__ o, int nb, int P, EQ lq, CW cw, LW lw) {
    float z[NBM];
#pragma unroll
    for (int n = 0; n < NBM; ++n)
        if (n < nb) z[n] = __builtin_amdgcn_exp2f(fmaf(yp[n], 0.33f * 1.4426950408889634f, lq(n)));
    // first and last band are replaced by their neighbours
    float zz[NBM];
#pragma unroll
    for (int n = 0; n < NBM; ++n)
        if (n < nb) zz[n] = n == 0 ? z[1] : (n == nb - 1 ? z[n - 1] : z[n]);
    // autocorrelation lags 0..P = real IDFT of the symmetric extension
    float r[PM + 1];
#pragma unroll
    for (int k = 0; k <= PM; ++k)
        if (k <= P) {
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < NBM; ++n)
                if (n < nb) acc = fmaf(zz[n], cw(k, n), acc);
            r[k] = acc;
        }
    // Levinson-Durbin
    float lp[PM];
    float e = r[0];
#pragma unroll
    for (int k = 0; k < PM; ++k)
        if (k < P) {
            float acc = r[k + 1];
#pragma unroll
            for (int j = 0; j < PM; ++j)
                if (j < k) acc = fmaf(lp[j], r[k - j], acc);
            const float refl = -div_nr(acc, e);
            e *= 1.0f - refl * refl;
#pragma unroll
            for (int j = 0; j < PM / 2 + 1; ++j)
                if (j < (k + 1) / 2) {
                    const int kj = k - 1 - j;
                    const float s = lp[j], t = lp[kj];
                    lp[j] = fmaf(refl, t, s);
                    if (j != kj) lp[kj] = fmaf(refl, s, t);
                }
            lp[k] = refl;
        }
    // cepstra of the gain-normalised polynomial [1, lp] / (e + 1e-8): c0 = ln(e + 1e-8), then the LPC recursion
    float c[PM + 1];
    c[0] = __builtin_amdgcn_logf(e + 1e-8f) * 0.6931471805599453f;
#pragma unroll
    for (int n = 1; n <= PM; ++n)
        if (n <= P) {
            float acc = 0.f;
#pragma unroll
            for (int m = 1; m < PM + 1; ++m)
                if (m < n) acc = fmaf((float)(n - m) * lp[m - 1], c[n - m], acc);
            c[n] = -fmaf(acc, 1.0f / (float)n, lp[n - 1]);
        }
#pragma unroll
    for (int n = 0; n <= PM; ++n)
        if (n <= P) o[n] = c[n] * lw(n);
}

// fixed sizes: thread = frame, tables in the kernel arguments
template <int NB, int PP>
__global__ __launch_bounds__(128) void plp_cep_fixed_kernel(PlpArgs a, PlpTab<NB, PP> t) {
    const int64_t f = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (f >= a.n_frames) return;
    plp_frame<NB, PP>(a.y + f * NB, a.out + f * (PP + 1), NB, PP, [&](int n) { return t.lq[n]; }, [&](int k, int n) { return t.cw[k * NB + n]; },
                      [&](int n) { return t.lw[n]; });
}

// runtime sizes up to the maxima: tables from LDS
__global__ __launch_bounds__(128) void plp_cep_kernel(PlpArgs a) {
    extern __shared__ float sh[];
    const int nb = a.nb, P = a.P;
    const int n_tab = nb + (P + 1) * nb + (P + 1);
    for (int i = threadIdx.x; i < n_tab; i += 128) sh[i] = a.tab[i];
    __syncthreads();
    const float* lq = sh;
    const float* cw = sh + nb;
    const float* lw = sh + nb + (P + 1) * nb;
    const int64_t f = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (f >= a.n_frames) return;
    plp_frame<PLP_NB_MAX, PLP_P_MAX>(a.y + f * nb, a.out + f * (P + 1), nb, P, [&](int n) { return lq[n]; }, [&](int k, int n) { return cw[k * nb + n]; },
                                     [&](int n) { return lw[n]; });
}

}  // namespace ssp

using namespace ssp;

extern "C" int ssp_plp_post(ssp_ctx* ctx, const float* logspec, const ssp_segments* frame_seg, int32_t n_bands, float fmax_hz,
                            int32_t plp_order, int32_t rasta, float lift, float* ceps_out, int where, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_plp_post");
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (!frame_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_plp_post: null segments");
    const int nb = n_bands, P = plp_order - 1;  // plp_order counts c0, as sidekit's argument does
    if (nb < 3 || nb > PLP_NB_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_plp_post: %d bands (3..%d supported)", nb, PLP_NB_MAX);
    if (P < 1 || P > PLP_P_MAX || P > nb - 1)
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_plp_post: plp_order=%d needs 1 <= order - 1 <= min(%d, bands - 1)", plp_order, PLP_P_MAX);
    if (!(fmax_hz > 0.f)) SSP_FAIL(SSP_ERR_INVALID, "ssp_plp_post: fmax");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_plp_post: where");
    if (frame_seg->host.front() != 0) SSP_FAIL(SSP_ERR_INVALID, "ssp_plp_post: segments must start at frame 0");
    const int64_t F = frame_seg->total();
    if (F == 0) return SSP_OK;
    if (!logspec || !ceps_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_plp_post: null data pointer");
    // tables (float64 on the host): equal loudness at the band centres, the IDFT weights of the symmetric extension, lifter
    std::vector<float> tab((size_t)nb + (size_t)(P + 1) * nb + (P + 1));
    const double zmax = 6.0 * std::asinh((double)fmax_hz / 600.0);
    for (int n = 0; n < nb; ++n) {
        const double hz = 600.0 * std::sinh(zmax * n / (nb - 1) / 6.0), fsq = hz * hz, ft = fsq + 1.6e5;
        const double eql = (fsq / ft) * (fsq / ft) * ((fsq + 1.44e6) / (fsq + 9.61e6));
        tab[n] = eql > 0.0 ? (float)(0.33 * std::log2(eql)) : -INFINITY;  // (the kernel forms (exp(y) eql)^0.33 as exp2(0.33 log2(e) y + this))
    }
    const int N = 2 * (nb - 1);
    for (int k = 0; k <= P; ++k)
        for (int n = 0; n < nb; ++n) {
            double w = std::cos(M_PI * k * n / (nb - 1)) / N;
            if (n > 0 && n < nb - 1) w *= 2.0;
            tab[(size_t)nb + (size_t)k * nb + n] = (float)w;
        }
    for (int n = 0; n <= P; ++n) tab[(size_t)nb + (size_t)(P + 1) * nb + n] = n == 0 || lift == 0.f ? 1.f : (float)std::pow((double)n, (double)lift);
    hipStream_t s = ctx->stream;
    DevBuf &d_tab = ctx->scratch[0], &d_y = ctx->scratch[5];
    SSP_TRY(d_tab.reserve(tab.size() * sizeof(float)));
    SSP_HIP(hipMemcpyAsync(d_tab.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, s));
    SSP_HIP(hipStreamSynchronize(s));  // `tab` (host) dies at return
    const size_t in_bytes = (size_t)F * nb * sizeof(float), out_bytes = (size_t)F * (P + 1) * sizeof(float);
    Staged sin, sout;
    int rc;
    const float* d_x = (const float*)sin.in(ctx, logspec, in_bytes, where, &rc);
    SSP_TRY(rc);
    float* d_out = (float*)sout.out(ctx, ceps_out, out_bytes, where, &rc);
    SSP_TRY(rc);
    const float* d_in = d_x;
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    if (rasta) {
        SSP_TRY(d_y.reserve(in_bytes));
        const int64_t n_thr = frame_seg->n * nb;
        hipLaunchKernelGGL(plp_rasta_kernel, dim3((unsigned)ceil_div<int64_t>(n_thr, 256)), dim3(256), 0, s, d_x, d_y.as<float>(),
                           frame_seg->dev.as<int64_t>(), frame_seg->n, nb);
        d_in = d_y.as<float>();
    }
    PlpArgs a{d_in, d_out, d_tab.as<float>(), F, nb, P};
    const size_t lds = tab.size() * sizeof(float);
    const unsigned grid = (unsigned)ceil_div<int64_t>(F, 128);
    auto launch_fixed = [&](auto tag_nb, auto tag_p) {
        constexpr int NB = decltype(tag_nb)::value, PP = decltype(tag_p)::value;
        PlpTab<NB, PP> t;
        memcpy(t.lq, tab.data(), sizeof(t.lq));
        memcpy(t.cw, tab.data() + NB, sizeof(t.cw));
        memcpy(t.lw, tab.data() + NB + (PP + 1) * NB, sizeof(t.lw));
        hipLaunchKernelGGL((plp_cep_fixed_kernel<NB, PP>), dim3(grid), dim3(128), 0, s, a, t);
    };
    if (nb == 21 && P == 12) launch_fixed(std::integral_constant<int, 21>{}, std::integral_constant<int, 12>{});
    else if (nb == 17 && P == 12) launch_fixed(std::integral_constant<int, 17>{}, std::integral_constant<int, 12>{});
    else hipLaunchKernelGGL(plp_cep_kernel, dim3(grid), dim3(128), lds, s, a);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(sout.back(ctx, ceps_out, out_bytes, where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));
    return SSP_OK;
}

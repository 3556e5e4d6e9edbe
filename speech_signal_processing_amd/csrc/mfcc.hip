// Fused MFCC pass for gfx950: framing + pre-emphasis + window + real FFT + spectrum + filterbank + log + DCT-II
// + delta / delta-delta + per-utterance CMVN in ONE kernel launch.
//
// Replaces (see include/ssp.h): utils/processing.py:19-144 (in-repo MFCC), the sidekit mfcc call sites
// GMM_UBM.py:89 / d_vector.py:91, the librosa call site MFCC_DTW.py:29, GMM_UBM.delta (GMM_UBM.py:53-69) and
// sklearn.preprocessing.scale (GMM_UBM.py:93).
//
// Two kernels:
//   mfcc_generic_kernel  — table driven, any power-of-two n_fft <= 2048, every cfg knob. One wave per frame, Stockham FFT
//                          with radix-16 / 8 / 4 passes in registers over ONE wave-private (in-place, padded) LDS buffer.
//   mfcc_fused512_kernel — (mfcc_fast.hip) the throughput kernel for n_fft == 512.
#include <algorithm>
#include <cstdlib>

#include "mfcc.hpp"
#include "cplx.hpp"

namespace ssp {

__device__ __forceinline__ void wave_lds_sync() {
    // LDS ops of one wave execute in issue order; this only stops the compiler from reordering across it.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct __attribute__((packed, aligned(4))) f4u {
    float x, y, z, w;
};

// One wave keeps its frame's M complex points in ONE wave-private LDS buffer, element i at i + (i >> 4) (one pad slot per 16: the
// stride-R scatters of the Stockham passes stay spread over the banks).  A pass reads everything it needs into registers, then
// writes in place.
__host__ __device__ constexpr int zpad(int i) { return i + (i >> 4); }

// points per lane and FFT pass: 16 for n_fft 2048 (passes 16, 16, 4), 8 for 1024 (8, 8, 8), 4 below (4, ..., then 2 if needed)
__host__ __device__ constexpr int fft_points_per_lane(int n_fft) { return n_fft >= 2048 ? 16 : (n_fft >= 1024 ? 8 : 4); }

// Stockham autosort pass of radix R: butterfly j < M/R takes in[j + m M/R] W_{R Ns}^{m k} (k = j mod Ns) and leaves its DFT_R at
// (j - k) R + k + r Ns.  A lane owns E/R butterflies (E points).  FIRST: the input is the unpadded windowed frame, no twiddles.
template <int NFFT, int R, int NS, bool FIRST>
__device__ __forceinline__ void fft_pass(v2f* buf, const v2f* tw, int lane) {
    constexpr int M = NFFT / 2, E = fft_points_per_lane(NFFT), NB = E / R, nb = M / R, active = M / E;
    constexpr int tstride = NFFT / (R * NS);
    v2f v[NB][R];
    if (active >= 64 || lane < active) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = lane + b * active;
#pragma unroll
            for (int r = 0; r < R; ++r) v[b][r] = FIRST ? buf[j + r * nb] : buf[zpad(j + r * nb)];
            if (!FIRST) {
                const int kt = (j & (NS - 1)) * tstride;
#pragma unroll
                for (int r = 1; r < R; ++r) v[b][r] = cmul(v[b][r], tw[r * kt]);
            }
            fft_small<R>(v[b]);
        }
    }
    wave_lds_sync();
    if (active >= 64 || lane < active) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = lane + b * active;
            const int k = j & (NS - 1);
            const int d = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[zpad(d + r * NS)] = v[b][r];
        }
    }
    wave_lds_sync();
}

template <int NFFT, int NS>
__device__ __forceinline__ void fft_rest(v2f* buf, const v2f* tw, int lane) {
    constexpr int M = NFFT / 2, E = fft_points_per_lane(NFFT);
    if constexpr (NS < M) {
        constexpr int R = M / NS >= E ? E : M / NS;
        fft_pass<NFFT, R, NS, false>(buf, tw, lane);
        fft_rest<NFFT, NS * R>(buf, tw, lane);
    }
}

// sample i of frame t (frame-local index), before pre-emphasis / window
__device__ __forceinline__ float frame_sample(const MfccArgs& a, const float* __restrict__ x, int64_t N, int64_t t,
                                              int i) {
    int64_t g = t * a.hop + i;
    if (a.frame_mode == 2) {
        g -= a.n_fft >> 1;
        if (g < 0) g = -g;
        if (g >= N) g = 2 * (N - 1) - g;
        g = g < 0 ? 0 : (g >= N ? N - 1 : g);
        return x[g];
    }
    return g < N ? x[g] : 0.0f;
}

__device__ __forceinline__ float apply_log(const MfccArgs& a, float v) {
    if (a.floor_mode == 1) v += a.eps;
    else if (a.floor_mode == 2) v = nanmax(v, a.eps);  // (numpy.maximum: a NaN stays a NaN)
    if (a.log_mode == 0) return logf(v);
    if (a.log_mode == 1) return log10f(v);
    return 10.0f * log10f(v);
}

#ifdef SSP_GSTAMP
// Diagnostic build only (never shipped): per-phase cycle accounting with s_memtime stamps.
__device__ unsigned long long g_gstamps[16];
#define GSTAMP(ph)                                                                                  \
    {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        st_acc[ph] += t_ - st_last;                                                                 \
        st_last = t_;                                                                               \
    }
#else
#define GSTAMP(ph)
#endif

template <int NFFT>
__global__ __launch_bounds__(512) void mfcc_generic_kernel(MfccArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = NFFT / 2, E = fft_points_per_lane(NFFT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockDim.x, nw = nt >> 6;
    const int nc = a.n_ceps;
#ifdef SSP_GSTAMP
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

    v2f* buf = reinterpret_cast<v2f*>(smem) + (size_t)wave * (M + (M >> 4));  // wave-private, M padded complex points
    // the pad slots (one per 16 points) are never written by the FFT, and the filterbank's 16-byte reads run past bin M over them
    // under zero weights: whatever a previous kernel left in LDS (a NaN, an infinity) must not be there
    for (int i = lane; i < M + (M >> 4); i += 64) buf[i] = v2f{0.f, 0.f};
    float* logmel = reinterpret_cast<float*>(smem + a.lds_logmel_off) + wave * a.n_filt;
    float* ceps = reinterpret_cast<float*>(smem + a.lds_ceps_off);
    float* dlt = reinterpret_cast<float*>(smem + a.lds_dlt_off);
    float* ddl = reinterpret_cast<float*>(smem + a.lds_ddl_off);
    float* lmrows = reinterpret_cast<float*>(smem + a.lds_lmrows_off);
    float* stats = reinterpret_cast<float*>(smem + a.lds_stats_off);  // [2][d_out] mean, inv std; [8] reduce scratch
    // tables staged in LDS: twiddles W_nfft^k (k < n_fft) and, when they fit, the transposed DCT matrix and the filter taps
    v2f* tw = reinterpret_cast<v2f*>(smem + a.lds_tw_off);
    for (int i = tid; i < NFFT; i += nt) tw[i] = reinterpret_cast<const v2f*>(a.twiddle)[i];
    // DCT matrix: transposed [n_filt][n_ceps] for the per-frame product; [n_ceps][n_filt4] (rows zero padded to a multiple of 4) for
    // the clamped product over a whole utterance's log-mel rows (top_db)
    float* dct_lds = reinterpret_cast<float*>(smem + (a.lds_dct_off >= 0 ? a.lds_dct_off : 0));
    const int n_filt4 = (a.n_filt + 3) & ~3;
    if (a.lds_dct_off >= 0) {
        if (a.top_db >= 0.f) {
            for (int i = tid; i < nc * n_filt4; i += nt) {
                const int q = i / n_filt4, j = i - q * n_filt4;
                dct_lds[i] = j < a.n_filt ? a.dct[q * a.n_filt + j] : 0.f;
            }
        } else {
            for (int i = tid; i < a.n_filt * nc; i += nt) dct_lds[i] = a.dctT[i];
        }
    }
    v4f* wt_lds = reinterpret_cast<v4f*>(smem + (a.lds_wt_off >= 0 ? a.lds_wt_off : 0));
    if (a.lds_wt_off >= 0)
        for (int i = tid; i < a.filt_w4_total; i += nt) wt_lds[i] = reinterpret_cast<const v4f*>(a.filt_wT)[i];
    __syncthreads();

    // this lane's filters in the first two groups of 64 (later groups reload theirs per frame): first bin rounded down to 4
    const int lo_0 = lane < a.n_filt ? a.filt_lo4[lane] : 0, lo_1 = lane + 64 < a.n_filt ? a.filt_lo4[lane + 64] : 0;
    const int lo_p = a.n_filt <= 32 ? a.filt_lo4[lane % a.n_filt] : 0;  // this lane's filter when every filter has 64 / n_filt lanes
    const int gs_0 = a.filt_grp[0], gs_1 = a.filt_grp[1], go_0 = a.filt_grp[8], go_1 = a.filt_grp[9];
    // DCT: lanes = (coefficient, part of the filter range); parts are summed by xor shuffles
    const int ncp = a.dct_ncp, dparts = 64 / ncp, dpart = lane / ncp, dq = lane & (ncp - 1);

    const bool pre = a.preemph_mode == 1;
    // one chunk per workgroup (persistent workgroups looping over chunks measured 20 % slower: the hardware dispatcher balances better
    // than a static stride, and the loop cost registers)
    const MfccChunk ch = a.chunks[blockIdx.x];
    const int64_t s0 = a.sample_off[ch.utt];
    const int64_t N = a.sample_off[ch.utt + 1] - s0;
    const int64_t f0 = a.frame_off[ch.utt];
    const int T = (int)(a.frame_off[ch.utt + 1] - f0);
    const float* __restrict__ x = a.samples + s0;
    const int t0 = ch.t0, n = ch.n;
    const int H = a.delta_order * a.delta_N;
    const int ta = max(t0 - H, 0), tb = min(t0 + n + H, T);  // cepstra rows kept in LDS
    float wave_max = -INFINITY;  // two-pass top_db: this wave's largest log-mel value
    GSTAMP(5)
    for (int t = ta + wave; t < tb; t += nw) {
        // ---- frame -> y[i] = (x[i] - p x[i-1]) w[i] (zero beyond win_len), unpadded floats over the wave buffer.  x[i-1] of a
        // lane's first sample comes from the neighbouring lane (lane 0: lane 63 of the previous sweep; i = 0: itself)
        float* yb = reinterpret_cast<float*>(buf);
        float carry = 0.f;
        const int64_t g0 = (int64_t)t * a.hop - (a.frame_mode == 2 ? M : 0);
        if (NFFT >= 256 && g0 >= 0 && g0 + NFFT <= N) {
            // interior frame: 4 consecutive samples per lane and sweep, every load of the frame in flight together
            constexpr int NSW = NFFT >= 256 ? NFFT / 256 : 1;
            const float* __restrict__ xp = x + g0 + 4 * lane;
            const float* __restrict__ wp = a.window + 4 * lane;
            f4u xv[NSW];
            v4f wv[NSW];
#pragma unroll
            for (int s = 0; s < NSW; ++s) {
                xv[s] = *reinterpret_cast<const f4u*>(xp + 256 * s);
                wv[s] = *reinterpret_cast<const v4f*>(wp + 256 * s);
            }
            GSTAMP(8)
#pragma unroll
            for (int s = 0; s < NSW; ++s) {
                const int i = 4 * lane + 256 * s;
                v4f xs = {xv[s].x, xv[s].y, xv[s].z, xv[s].w};
                if (i + 3 >= a.win_len) {
                    xs.x = i < a.win_len ? xs.x : 0.f;
                    xs.y = i + 1 < a.win_len ? xs.y : 0.f;
                    xs.z = i + 2 < a.win_len ? xs.z : 0.f;
                    xs.w = 0.f;
                }
                float prev = __shfl_up(xs.w, 1);
                if (lane == 0) prev = s == 0 ? xs.x : carry;
                carry = __shfl(xs.w, 63);
                v4f y = xs;
                if (pre) y = xs - a.preemph * v4f{prev, xs.x, xs.y, xs.z};
                *reinterpret_cast<v4f*>(yb + i) = y * wv[s];
            }
        } else {
            for (int i0 = lane; i0 < NFFT; i0 += 512) {  // 8 sweeps at a time: their loads are all in flight together
                float xv[8], wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 64 * u;
                    xv[u] = i < a.win_len ? frame_sample(a, x, N, t, i) : 0.f;
                    wv[u] = i < NFFT ? a.window[i] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 64 * u;
                    float prev = __shfl_up(xv[u], 1);
                    if (lane == 0) prev = i == 0 ? xv[u] : carry;
                    carry = __shfl(xv[u], 63);
                    if (i < NFFT) yb[i] = (pre ? xv[u] - a.preemph * prev : xv[u]) * wv[u];
                }
            }
        }
        wave_lds_sync();
        GSTAMP(0)
        // touch the cache lines of this wave's NEXT frame (one dword per 128-byte line and lane): its staging loads then hit the
        // L2 instead of waiting microseconds on HBM with nothing else to run
        float touch0 = 0.f, touch1 = 0.f;
        if (a.prefetch && t + nw < tb) {
            const int64_t gn = (int64_t)(t + nw) * a.hop - (a.frame_mode == 2 ? M : 0);
            const int64_t gl = gn + 32 * min(lane, NFFT / 32), ge = gn + NFFT - 1;
            touch0 = x[max((int64_t)0, min(gl, N - 1))];
            touch1 = x[max((int64_t)0, min(ge, N - 1))];
        }
        // ---- M-point complex FFT of z[m] = (y[2m], y[2m+1])
        fft_pass<NFFT, (E < M ? E : M), 1, true>(buf, tw, lane);
        fft_rest<NFFT, (E < M ? E : M)>(buf, tw, lane);
        GSTAMP(1)
        // ---- real-FFT split + magnitude / power.  Bins k and M - k come from the same pair: with E = (Z[k] + conj Z[M-k]) / 2,
        // D = (Z[k] - conj Z[M-k]) / 2 and o = (-i D) W^k:  X[k] = E + o,  X[M-k] = conj(E - o).  Into registers, then P[0..M] as
        // unpadded floats over the same buffer.
        constexpr int NSP = (M / 2 + 1 + 63) / 64;
        float pa[NSP], pb[NSP];
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            const int k = lane + 64 * i;
            pa[i] = pb[i] = 0.f;
            if (k <= M / 2) {
                const v2f zk = buf[zpad(k)];
                const v2f zm = buf[zpad((M - k) & (M - 1))];
                const v2f hz = zk * 0.5f;
                const v2f e = __builtin_elementwise_fma(zm, v2f{0.5f, -0.5f}, hz);
                const v2f d = __builtin_elementwise_fma(zm, v2f{-0.5f, 0.5f}, hz);
                const v2f o = cmul_negi(d, tw[k]);
                const v2f xa = e + o, xb = e - o;
                float p0 = xa.x * xa.x + xa.y * xa.y, p1 = xb.x * xb.x + xb.y * xb.y;
                pa[i] = p0;
                pb[i] = p1;
            }
        }
        // magnitude dialects: v_sqrt_f32 (1 ulp) under ONE wave-uniform branch behind the loop.  (Inside the unrolled loop the
        // compiler if-converted the test: sqrtf()'s correctly rounded expansion, ~18 instructions per bin, ran for the power
        // dialects too and a select dropped the result.)
        if (a.spec_power == 1) {
#pragma unroll
            for (int i = 0; i < NSP; ++i) {
                pa[i] = __builtin_amdgcn_sqrtf(pa[i]);
                pb[i] = __builtin_amdgcn_sqrtf(pb[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            pa[i] *= a.spec_scale;
            pb[i] *= a.spec_scale;
        }
        wave_lds_sync();
        float* P = reinterpret_cast<float*>(buf);
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            const int k = lane + 64 * i;
            if (k <= M / 2) {
                P[k] = pa[i];
                P[M - k] = pb[i];
            }
        }
        wave_lds_sync();
        GSTAMP(2)
        // ---- banded filterbank + log: a lane per filter, 64 filters at a time, four taps per 16-byte read; the taps of a group
        // come from a transposed, zero-padded table (a filter's reads start at its first bin rounded down to 4 and may run past
        // bin M into stale, finite buffer contents under zero weights)
        float* lm = a.top_db >= 0.f ? lmrows + (size_t)(t - ta) * a.lm_stride : logmel;
        auto filterbank = [&](const v4f* wtab) {
            for (int g = 0; g * 64 < a.n_filt; ++g) {
                const int j = g * 64 + lane;
                const int nl = min(64, a.n_filt - g * 64);  // filters (= table columns) of this group
                const bool valid = lane < nl;
                const int lo = g == 0 ? lo_0 : (g == 1 ? lo_1 : (valid ? a.filt_lo4[j] : 0));
                const int nstep = g == 0 ? gs_0 : (g == 1 ? gs_1 : a.filt_grp[g]);  // 16-byte steps, even
                const v4f* wt = wtab + (g == 0 ? go_0 : (g == 1 ? go_1 : a.filt_grp[8 + g])) + min(lane, nl - 1);
                const float* pp = P + lo;
                v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                for (int st = 0; st < nstep; st += 2) {  // (4 steps per iteration measured no faster and cost registers)
                    const v4f w0 = wt[st * nl], w1 = wt[(st + 1) * nl];
                    const v4f p0 = *reinterpret_cast<const v4f*>(pp + 4 * st), p1 = *reinterpret_cast<const v4f*>(pp + 4 * st + 4);
                    acc0 = __builtin_elementwise_fma(p0, w0, acc0);
                    acc1 = __builtin_elementwise_fma(p1, w1, acc1);
                }
                if (valid) {
                    const v4f acc = acc0 + acc1;
                    const float v = apply_log(a, (acc.x + acc.y) + (acc.z + acc.w));
                    lm[j] = v;
                    if (a.lm_out && t >= t0 && t < t0 + n) {
                        a.lm_out[(size_t)(f0 + t) * a.n_filt + j] = v;
                        wave_max = nanmax(wave_max, v);
                    }
                }
            }
        };
        // <= 32 filters (sidekit's 24, the 21 Bark bands of PLP): every filter is spread over np = 64 / n_filt lanes (lane = part *
        // n_filt + filter), each part takes a contiguous share of the 16-byte steps and the partial sums meet in the part-0 lane —
        // the dependent load -> FMA chain per lane is np times shorter (PLP front end: 51 -> 39 ms)
        auto filterbank_parts = [&](const v4f* wtab) {
            const int nl = a.n_filt, np = 64 / nl;
            const int part = lane / nl, fi = lane - part * nl;
            const int spp = ((gs_0 / 2 + np - 1) / np) * 2;  // steps per part, even
            const int st0 = part * spp, st1 = part < np ? min(gs_0, st0 + spp) : st0;
            const v4f* wt = wtab + go_0 + fi;
            const float* pp = P + lo_p;
            v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int st = st0; st < st1; st += 2) {
                const v4f w0 = wt[st * nl], w1 = wt[(st + 1) * nl];
                const v4f p0 = *reinterpret_cast<const v4f*>(pp + 4 * st), p1 = *reinterpret_cast<const v4f*>(pp + 4 * st + 4);
                acc0 = __builtin_elementwise_fma(p0, w0, acc0);
                acc1 = __builtin_elementwise_fma(p1, w1, acc1);
            }
            const v4f acc = acc0 + acc1;
            const float partial = (acc.x + acc.y) + (acc.z + acc.w);
            float tot = partial;
            for (int q = 1; q < np; ++q) tot += __shfl(partial, (lane + q * nl) & 63);
            if (part == 0) {
                const float v = apply_log(a, tot);
                lm[fi] = v;
                if (a.lm_out && t >= t0 && t < t0 + n) {
                    a.lm_out[(size_t)(f0 + t) * a.n_filt + fi] = v;
                    wave_max = nanmax(wave_max, v);
                }
            }
        };
        if (a.n_filt <= 32) {
            if (a.lds_wt_off >= 0) filterbank_parts(wt_lds);
            else filterbank_parts(reinterpret_cast<const v4f*>(a.filt_wT));
        } else if (a.lds_wt_off >= 0) filterbank(wt_lds);
        else filterbank(reinterpret_cast<const v4f*>(a.filt_wT));
        if (a.top_db >= 0.f && a.n_filt + lane < n_filt4) lm[a.n_filt + lane] = 0.f;  // rows are read 16 bytes at a time below
        wave_lds_sync();
        GSTAMP(3)
        if (a.lm_out) {  // two-pass top_db: the clamp needs the utterance maximum, the DCT follows in a second kernel
            asm volatile("" ::"v"(touch0), "v"(touch1));
            continue;
        }
        // ---- DCT-II rows (skipped here when the utterance-level top_db clamp must come first)
        if (a.top_db < 0.f && a.dct_identity) {  // (the PLP front end: Bark log spectrum out)
            for (int q = lane; q < nc; q += 64) ceps[(size_t)(t - ta) * nc + q] = lm[q];
        } else if (a.top_db < 0.f) {
            auto dct_rows = [&](const float* tbl) {
                for (int q0 = 0; q0 < nc; q0 += ncp) {
                    const int q = q0 + dq;
                    const float* tq = tbl + (q < nc ? q : 0);
                    float acc = 0.f;
                    int j = dpart;
                    for (; j + 3 * dparts < a.n_filt; j += 4 * dparts) {
                        const float l0 = lm[j], l1 = lm[j + dparts], l2 = lm[j + 2 * dparts], l3 = lm[j + 3 * dparts];
                        const float c0 = tq[j * nc], c1 = tq[(j + dparts) * nc], c2 = tq[(j + 2 * dparts) * nc], c3 = tq[(j + 3 * dparts) * nc];
                        acc = fmaf(l0, c0, acc);
                        acc = fmaf(l1, c1, acc);
                        acc = fmaf(l2, c2, acc);
                        acc = fmaf(l3, c3, acc);
                    }
                    for (; j < a.n_filt; j += dparts) acc = fmaf(lm[j], tq[j * nc], acc);
                    for (int o = ncp; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
                    if (q < nc && dpart == 0) ceps[(size_t)(t - ta) * nc + q] = acc;
                }
            };
            if (a.lds_dct_off >= 0) dct_rows(dct_lds);
            else dct_rows(a.dctT);
        }
        wave_lds_sync();
        asm volatile("" ::"v"(touch0), "v"(touch1));
        GSTAMP(4)
    }
    __syncthreads();
    GSTAMP(6)
    if (a.lm_out) {  // utterance maximum for the second pass: one atomic per wave (float order through the integer trick)
        for (int o = 32; o > 0; o >>= 1) wave_max = nanmax(wave_max, __shfl_xor(wave_max, o));
        if (lane == 0 && (wave_max > -INFINITY || wave_max != wave_max)) {
            // (a NaN maximum sticks: its canonical bits 0x7fc00000 are below every negative float's as unsigned and above every positive
            //  float's as int, so whichever atomic a later chunk uses leaves it in place)
            if (wave_max != wave_max) atomicExch(reinterpret_cast<unsigned*>(a.utt_max + ch.utt), 0x7fc00000u);
            else if (wave_max >= 0.f) atomicMax(reinterpret_cast<int*>(a.utt_max + ch.utt), __float_as_int(wave_max));
            else atomicMin(reinterpret_cast<unsigned*>(a.utt_max + ch.utt), __float_as_uint(wave_max));
        }
        return;
    }

    if (a.top_db >= 0.f) {  // whole utterance is in this chunk (host guarantees): max over all log-mel values
        float mx = -INFINITY;
        const int rows = tb - ta;
        for (int i = tid; i < rows * a.n_filt; i += nt) {
            const int r = i / a.n_filt;
            mx = nanmax(mx, lmrows[(size_t)r * a.lm_stride + (i - r * a.n_filt)]);
        }
        for (int o = 32; o > 0; o >>= 1) mx = nanmax(mx, __shfl_xor(mx, o));
        float* red = stats + 2 * a.d_out;
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        float thr = red[0];
        for (int w = 1; w < nw; ++w) thr = nanmax(thr, red[w]);
        thr -= a.top_db;
        // consecutive lanes take consecutive rows (stride 4 x odd floats: conflict-free 16-byte reads) of two coefficients, whose
        // DCT rows are broadcast reads
        auto dct_clamped4 = [&](const float* tbl) {
            const int nq2 = (nc + 1) >> 1;
            for (int i = tid; i < rows * nq2; i += nt) {
                const int qb = i / rows, r = i - qb * rows;
                const int q0 = 2 * qb, q1 = min(q0 + 1, nc - 1);
                const float* lm = lmrows + (size_t)r * a.lm_stride;
                const float *d0 = tbl + q0 * n_filt4, *d1 = tbl + q1 * n_filt4;
                v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < n_filt4; j += 4) {
                    v4f l = *reinterpret_cast<const v4f*>(lm + j);
                    l = v4f{nanmax(l.x, thr), nanmax(l.y, thr), nanmax(l.z, thr), nanmax(l.w, thr)};
                    acc0 = __builtin_elementwise_fma(l, *reinterpret_cast<const v4f*>(d0 + j), acc0);
                    acc1 = __builtin_elementwise_fma(l, *reinterpret_cast<const v4f*>(d1 + j), acc1);
                }
                ceps[(size_t)r * nc + q0] = (acc0.x + acc0.y) + (acc0.z + acc0.w);
                if (q0 + 1 < nc) ceps[(size_t)r * nc + q1] = (acc1.x + acc1.y) + (acc1.z + acc1.w);
            }
        };
        auto dct_clamped = [&](const float* tbl) {  // table in global memory, transposed [n_filt][n_ceps]
            for (int i = tid; i < rows * nc; i += nt) {
                const int q = i / rows, r = i - q * rows;
                const float* lm = lmrows + (size_t)r * a.lm_stride;
                float acc = 0.f;
                for (int j = 0; j < a.n_filt; ++j) acc = fmaf(nanmax(lm[j], thr), tbl[j * nc + q], acc);
                ceps[(size_t)r * nc + q] = acc;
            }
        };
        if (a.lds_dct_off >= 0) dct_clamped4(dct_lds);
        else dct_clamped(a.dctT);
        __syncthreads();
    }

    // ---- delta (rows da..db) and delta-delta (rows t0..t0+n), edge padding at utterance ends (GMM_UBM.py:64)
    const int Nd = a.delta_N;
    const float inv_den = a.delta_inv_denom;
    int da = t0, db = t0 + n;
    if (a.delta_order >= 1) {
        const int ext = (a.delta_order - 1) * Nd;
        da = max(t0 - ext, 0);
        db = min(t0 + n + ext, T);
        for (int i = tid; i < (db - da) * nc; i += nt) {
            const int r = i / nc, q = i - r * nc;
            const int u = da + r;
            float acc = 0.f * ceps[(size_t)(u - ta) * nc + q];  // (the n = 0 term of numpy.dot, GMM_UBM.py:68: 0 . NaN = NaN)
            for (int m = 1; m <= Nd; ++m) {
                const int up = min(u + m, T - 1), um = max(u - m, 0);
                acc += (float)m * (ceps[(size_t)(up - ta) * nc + q] - ceps[(size_t)(um - ta) * nc + q]);
            }
            dlt[i] = acc * inv_den;
        }
        __syncthreads();
    }
    if (a.delta_order >= 2) {
        for (int i = tid; i < n * nc; i += nt) {
            const int r = i / nc, q = i - r * nc;
            const int u = t0 + r;
            float acc = 0.f * dlt[(size_t)(u - da) * nc + q];
            for (int m = 1; m <= Nd; ++m) {
                const int up = min(u + m, T - 1), um = max(u - m, 0);
                acc += (float)m * (dlt[(size_t)(up - da) * nc + q] - dlt[(size_t)(um - da) * nc + q]);
            }
            ddl[i] = acc * inv_den;
        }
        __syncthreads();
    }

    const int D = a.d_out;
    auto value = [&](int r /*row within chunk*/, int d) -> float {
        const int blk = d / nc, q = d - blk * nc;
        if (blk == 0) return ceps[(size_t)(t0 + r - ta) * nc + q];
        if (blk == 1) return dlt[(size_t)(t0 + r - da) * nc + q];
        return ddl[(size_t)r * nc + q];
    };

    if (a.cmvn) {  // per-utterance, per-dimension (x - mean) / std, ddof = 0, std < 10 eps -> 1 (sklearn scale)
        for (int d = wave; d < D; d += nw) {
            // (statistics over the entries that are not NaN, which stay NaN: sklearn's nanmean / nanstd — see cmvn_kernel)
            float s = 0.f, cn = 0.f;
            for (int r = lane; r < n; r += 64) {
                const float x = value(r, d);
                const bool ok = x == x;
                s += ok ? x : 0.f;
                cn += ok ? 1.f : 0.f;
            }
            for (int o = 32; o > 0; o >>= 1) {
                s += __shfl_xor(s, o);
                cn += __shfl_xor(cn, o);
            }
            const float mean = s / cn;
            float v = 0.f;
            for (int r = lane; r < n; r += 64) {
                const float x = value(r, d);
                const float e = x == x ? x - mean : 0.f;
                v = fmaf(e, e, v);
            }
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            float sd = sqrtf(v / cn);
            if (sd < 10.0f * 1.1920929e-07f) sd = 1.0f;
            if (lane == 0) {
                stats[d] = mean;
                stats[D + d] = 1.0f / sd;
            }
        }
        __syncthreads();
    }

    float* __restrict__ out = a.out + (size_t)(f0 + t0) * D;
    for (int i = tid; i < n * D; i += nt) {
        const int r = i / D, d = i - r * D;
        float v = value(r, d);
        if (a.cmvn) v = (v - stats[d]) * stats[D + d];
        out[i] = v;
    }
    GSTAMP(7)
#ifdef SSP_GSTAMP
    if (lane == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&g_gstamps[i], st_acc[i]);
    if (tid == 0) atomicAdd(&g_gstamps[15], (unsigned long long)nw);
#endif
}

// one workgroup per utterance: max over its log-mel rows, clamp at max - top_db (librosa power_to_db over the WHOLE utterance),
// DCT-II rows.  Any utterance length (global memory, no LDS rows).
__global__ __launch_bounds__(256) void topdb_dct_kernel(const float* __restrict__ lm, const int64_t* __restrict__ off, int n_filt, int nc,
                                                        const float* __restrict__ dct, float top_db, float* __restrict__ out) {
    __shared__ float red[4];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t f0 = off[u];
    const int64_t T = off[u + 1] - f0;
    if (T == 0) return;
    const float* __restrict__ x = lm + f0 * n_filt;
    float mx = -INFINITY;
    for (int64_t i = tid; i < T * n_filt; i += 256) mx = nanmax(mx, x[i]);
    for (int o = 32; o > 0; o >>= 1) mx = nanmax(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    const float thr = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3])) - top_db;
    for (int64_t i = tid; i < T * nc; i += 256) {
        const int64_t r = i / nc;
        const int q = (int)(i - r * nc);
        const float* __restrict__ row = x + r * n_filt;
        const float* __restrict__ drow = dct + (size_t)q * n_filt;
        float acc = 0.f;
        for (int j = 0; j < n_filt; ++j) acc = fmaf(nanmax(row[j], thr), drow[j], acc);
        out[(f0 + r) * nc + q] = acc;
    }
}

// second pass, one workgroup per chunk of the first pass: rows staged in LDS 64 at a time (stride 4 x odd floats), DCT rows in LDS
// [n_ceps][n_filt4], clamp at the utterance maximum - top_db, two coefficients per thread
__global__ __launch_bounds__(256) void topdb_dct_chunk_kernel(const float* __restrict__ lm, const MfccChunk* __restrict__ chunks,
                                                              const int64_t* __restrict__ off, const float* __restrict__ utt_max,
                                                              int n_filt, int nc, const float* __restrict__ dct, float top_db,
                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int n_filt4 = (n_filt + 3) & ~3, stride = 4 * ((n_filt4 / 4) | 1);
    float* dl = reinterpret_cast<float*>(smem);  // [nc][n_filt4]
    float* rows = dl + nc * n_filt4;             // [64][stride]
    for (int i = tid; i < nc * n_filt4; i += 256) {
        const int q = i / n_filt4, j = i - q * n_filt4;
        dl[i] = j < n_filt ? dct[q * n_filt + j] : 0.f;
    }
    const MfccChunk ch = chunks[blockIdx.x];
    const int64_t fbase = off[ch.utt] + ch.t0;
    const float thr = utt_max[ch.utt] - top_db;
    const int nq2 = (nc + 1) >> 1;
    for (int r0 = 0; r0 < ch.n; r0 += 64) {
        const int nr = min(64, ch.n - r0);
        __syncthreads();
        const float* __restrict__ src = lm + (fbase + r0) * n_filt;
        for (int i = tid; i < nr * n_filt4; i += 256) {
            const int r = i / n_filt4, j = i - r * n_filt4;
            rows[r * stride + j] = j < n_filt ? src[(size_t)r * n_filt + j] : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < nr * nq2; i += 256) {
            const int qb = i / nr, r = i - qb * nr;
            const int q0 = 2 * qb, q1 = min(q0 + 1, nc - 1);
            const float* l_ = rows + r * stride;
            const float *d0 = dl + q0 * n_filt4, *d1 = dl + q1 * n_filt4;
            v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < n_filt4; j += 4) {
                v4f l = *reinterpret_cast<const v4f*>(l_ + j);
                l = v4f{nanmax(l.x, thr), nanmax(l.y, thr), nanmax(l.z, thr), nanmax(l.w, thr)};
                acc0 = __builtin_elementwise_fma(l, *reinterpret_cast<const v4f*>(d0 + j), acc0);
                acc1 = __builtin_elementwise_fma(l, *reinterpret_cast<const v4f*>(d1 + j), acc1);
            }
            float* __restrict__ o = out + (fbase + r0 + r) * nc;
            o[q0] = (acc0.x + acc0.y) + (acc0.z + acc0.w);
            if (q0 + 1 < nc) o[q1] = (acc1.x + acc1.y) + (acc1.z + acc1.w);
        }
    }
}

int launch_topdb_dct(const float* logmel, const int64_t* frame_off_dev, int64_t n_utt, const MfccChunk* chunks, int n_chunks,
                     const float* utt_max, int n_filt, int n_ceps, const float* dct, float top_db, float* out, hipStream_t stream) {
    if (n_utt <= 0) return SSP_OK;
    if (n_utt > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "top_db: too many utterances");
    const int n_filt4 = (n_filt + 3) & ~3, stride = 4 * ((n_filt4 / 4) | 1);
    const size_t lds = ((size_t)n_ceps * n_filt4 + (size_t)64 * stride) * sizeof(float);
    if (lds <= 64 * 1024 && n_chunks > 0) {
        hipLaunchKernelGGL(topdb_dct_chunk_kernel, dim3(n_chunks), dim3(256), lds, stream, logmel, chunks, frame_off_dev, utt_max, n_filt,
                           n_ceps, dct, top_db, out);
    } else {  // a DCT matrix too large for the LDS: one workgroup per utterance straight from global memory
        hipLaunchKernelGGL(topdb_dct_kernel, dim3((unsigned)n_utt), dim3(256), 0, stream, logmel, frame_off_dev, n_filt, n_ceps, dct, top_db, out);
    }
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

template <int NFFT>
static int launch_generic_n(MfccArgs args, int n_chunks, size_t lds_bytes, int n_waves, int num_cu, hipStream_t stream) {
    const void* kfn = reinterpret_cast<const void*>(mfcc_generic_kernel<NFFT>);
    if (lds_bytes > 64 * 1024) SSP_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    (void)num_cu;
    args.n_chunks = n_chunks;
    args.prefetch = 1;
    hipLaunchKernelGGL((mfcc_generic_kernel<NFFT>), dim3(n_chunks), dim3(64 * n_waves), lds_bytes, stream, args);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

int launch_mfcc_generic(const MfccArgs& args, int n_chunks, size_t lds_bytes, int n_waves, int num_cu, hipStream_t stream) {
    if (n_chunks <= 0) return SSP_OK;
    if (n_waves != 4 && n_waves != 8) SSP_FAIL(SSP_ERR_INVALID, "mfcc: %d waves per workgroup", n_waves);
    switch (args.n_fft) {
        case 64: return launch_generic_n<64>(args, n_chunks, lds_bytes, n_waves, num_cu, stream);
        case 128: return launch_generic_n<128>(args, n_chunks, lds_bytes, n_waves, num_cu, stream);
        case 256: return launch_generic_n<256>(args, n_chunks, lds_bytes, n_waves, num_cu, stream);
        case 512: return launch_generic_n<512>(args, n_chunks, lds_bytes, n_waves, num_cu, stream);
        case 1024: return launch_generic_n<1024>(args, n_chunks, lds_bytes, n_waves, num_cu, stream);
        case 2048: return launch_generic_n<2048>(args, n_chunks, lds_bytes, n_waves, num_cu, stream);
    }
    SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: n_fft=%d", args.n_fft);
}

}  // namespace ssp

#ifdef SSP_GSTAMP
extern "C" int ssp_debug_gstamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ssp::g_gstamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ssp::g_gstamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// Fused MFCC pass for gfx950: framing + pre-emphasis + window + real FFT + spectrum + filterbank + log + DCT-II
// + delta / delta-delta + per-utterance CMVN in ONE kernel launch.
//
// Replaces (see include/ssp.h): utils/processing.py:19-144 (in-repo MFCC), the sidekit mfcc call sites
// GMM_UBM.py:89 / d_vector.py:91, the librosa call site MFCC_DTW.py:29, GMM_UBM.delta (GMM_UBM.py:53-69) and
// sklearn.preprocessing.scale (GMM_UBM.py:93).
//
// Two kernels:
//   mfcc_generic_kernel  — table driven, any power-of-two n_fft <= 2048, every cfg knob. One wave per frame,
//                          radix-4 Stockham FFT through wave-private LDS.
//   mfcc_fused512_kernel — (mfcc_fast.hip) the throughput kernel for n_fft == 512.
#include "mfcc.hpp"

namespace ssp {

__device__ __forceinline__ void wave_lds_sync() {
    // LDS ops of one wave execute in issue order; this only stops the compiler from reordering across it.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
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
    else if (a.floor_mode == 2) v = fmaxf(v, a.eps);
    if (a.log_mode == 0) return logf(v);
    if (a.log_mode == 1) return log10f(v);
    return 10.0f * log10f(v);
}

__global__ __launch_bounds__(256) void mfcc_generic_kernel(MfccArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = a.n_fft >> 1;
    const int nc = a.n_ceps;

    float2* fftbuf = reinterpret_cast<float2*>(smem) + (size_t)wave * 2 * M;  // [2][M] per wave
    float* logmel = reinterpret_cast<float*>(smem + a.lds_logmel_off) + wave * a.n_filt;
    float* ceps = reinterpret_cast<float*>(smem + a.lds_ceps_off);
    float* dlt = reinterpret_cast<float*>(smem + a.lds_dlt_off);
    float* ddl = reinterpret_cast<float*>(smem + a.lds_ddl_off);
    float* lmrows = reinterpret_cast<float*>(smem + a.lds_lmrows_off);
    float* stats = reinterpret_cast<float*>(smem + a.lds_stats_off);  // [2][d_out] mean, inv std; [256] reduce scratch
    // twiddles W_nfft^k, k < n_fft, in LDS: the FFT passes would otherwise wait on 3 dependent global loads per butterfly
    float2* tw = reinterpret_cast<float2*>(smem + a.lds_tw_off);
    for (int i = tid; i < a.n_fft; i += 256) tw[i] = a.twiddle[i];
    __syncthreads();

    const MfccChunk ch = a.chunks[blockIdx.x];
    const int64_t s0 = a.sample_off[ch.utt];
    const int64_t N = a.sample_off[ch.utt + 1] - s0;
    const int64_t f0 = a.frame_off[ch.utt];
    const int T = (int)(a.frame_off[ch.utt + 1] - f0);
    const float* __restrict__ x = a.samples + s0;
    const int t0 = ch.t0, n = ch.n;
    const int H = a.delta_order * a.delta_N;
    const int ta = max(t0 - H, 0), tb = min(t0 + n + H, T);  // cepstra rows kept in LDS

    for (int t = ta + wave; t < tb; t += 4) {
        // ---- frame -> complex sequence z[m] = (y[2m], y[2m+1]), windowed, zero padded to n_fft
        float2* in = fftbuf;
        float2* outb = fftbuf + M;
        for (int m = lane; m < M; m += 64) {
            const int i0 = 2 * m;
            float v0 = 0.f, v1 = 0.f;
            if (i0 < a.win_len) {
                const float xm1 = frame_sample(a, x, N, t, i0 > 0 ? i0 - 1 : 0);
                const float x0 = frame_sample(a, x, N, t, i0);
                const float x1 = (i0 + 1 < a.win_len) ? frame_sample(a, x, N, t, i0 + 1) : 0.f;
                if (a.preemph_mode == 1) {
                    v0 = x0 - a.preemph * xm1;
                    v1 = x1 - a.preemph * x0;
                } else {
                    v0 = x0;
                    v1 = x1;
                }
                v0 *= a.window[i0];
                v1 = (i0 + 1 < a.win_len) ? v1 * a.window[i0 + 1] : 0.f;
            }
            in[m] = make_float2(v0, v1);
        }
        wave_lds_sync();
        // ---- M-point complex FFT, Stockham autosort, radix 4 (+ one radix-2 pass if log2(M) is odd)
        int Ns = 1;
        for (; Ns * 4 <= M; Ns *= 4) {
            const int q = M >> 2;
            const int tstride = M / (2 * Ns);  // W_{4Ns}^{k} = W_nfft^{k * nfft/(4Ns)}
            for (int j = lane; j < q; j += 64) {
                const int k = j & (Ns - 1);
                float2 v0 = in[j], v1 = in[j + q], v2 = in[j + 2 * q], v3 = in[j + 3 * q];
                if (Ns > 1) {
                    v1 = cmul(v1, tw[k * tstride]);
                    v2 = cmul(v2, tw[2 * k * tstride]);
                    v3 = cmul(v3, tw[3 * k * tstride]);
                }
                const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
                const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
                const float2 a3 = make_float2(v1.y - v3.y, -(v1.x - v3.x));  // -i * (v1 - v3)
                const int d = ((j - k) << 2) + k;
                outb[d] = make_float2(a0.x + a2.x, a0.y + a2.y);
                outb[d + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
                outb[d + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
                outb[d + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
            }
            wave_lds_sync();
            float2* tmp = in;
            in = outb;
            outb = tmp;
        }
        if (Ns < M) {  // one radix-2 pass, Ns == M/2
            const int h = M >> 1;
            const int tstride = M / Ns;  // W_{2Ns}^k = W_nfft^{k * nfft/(2Ns)}
            for (int j = lane; j < h; j += 64) {
                const int k = j & (Ns - 1);
                const float2 v0 = in[j];
                const float2 v1 = cmul(in[j + h], tw[k * tstride]);
                const int d = ((j - k) << 1) + k;
                outb[d] = make_float2(v0.x + v1.x, v0.y + v1.y);
                outb[d + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
            }
            wave_lds_sync();
            float2* tmp = in;
            in = outb;
            outb = tmp;
        }
        // ---- real-FFT split + magnitude / power: bins 0..M from Z = in[]
        float* P = reinterpret_cast<float*>(outb);
        for (int k = lane; k <= M; k += 64) {
            const float2 zk = in[k & (M - 1)];
            const float2 zm = in[(M - k) & (M - 1)];
            const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);   // E = (Z[k] + conj Z[M-k]) / 2
            const float dr = 0.5f * (zk.x - zm.x), di = 0.5f * (zk.y + zm.y);   // D = (Z[k] - conj Z[M-k]) / 2
            const float2 w = tw[k];                                              // W_nfft^k
            const float2 o = cmul(make_float2(di, -dr), w);                     // (-i D) W^k
            const float re = er + o.x, im = ei + o.y;
            float p = re * re + im * im;
            if (a.spec_power == 1) p = sqrtf(p);
            P[k] = p * a.spec_scale;
        }
        wave_lds_sync();
        // ---- banded filterbank + log
        float* lm = a.top_db >= 0.f ? lmrows + (size_t)(t - ta) * a.n_filt : logmel;
        for (int j = lane; j < a.n_filt; j += 64) {
            const int lo = a.filt_lo[j], len = a.filt_len[j];
            const float* __restrict__ w = a.filt_w + a.filt_ofs[j];
            float acc = 0.f;
            for (int i = 0; i < len; ++i) acc = fmaf(P[lo + i], w[i], acc);
            lm[j] = apply_log(a, acc);
            if (a.lm_out && t >= t0 && t < t0 + n) a.lm_out[(size_t)(f0 + t) * a.n_filt + j] = lm[j];
        }
        wave_lds_sync();
        if (a.lm_out) continue;  // two-pass top_db: the clamp needs the utterance maximum, the DCT follows in a second kernel
        // ---- DCT-II rows (skipped here when the utterance-level top_db clamp must come first)
        if (a.top_db < 0.f) {
            for (int q = lane; q < nc; q += 64) {
                const float* __restrict__ drow = a.dct + (size_t)q * a.n_filt;
                float acc = 0.f;
                for (int j = 0; j < a.n_filt; ++j) acc = fmaf(lm[j], drow[j], acc);
                ceps[(size_t)(t - ta) * nc + q] = acc;
            }
        }
        wave_lds_sync();
    }
    __syncthreads();
    if (a.lm_out) return;

    if (a.top_db >= 0.f) {  // whole utterance is in this chunk (host guarantees): max over all log-mel values
        float mx = -INFINITY;
        const int tot = (tb - ta) * a.n_filt;
        for (int i = tid; i < tot; i += 256) mx = fmaxf(mx, lmrows[i]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float* red = stats + 2 * a.d_out;
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        const float thr = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - a.top_db;
        for (int i = tid; i < (tb - ta) * nc; i += 256) {
            const int r = i / nc, q = i - r * nc;
            const float* __restrict__ drow = a.dct + (size_t)q * a.n_filt;
            const float* lm = lmrows + (size_t)r * a.n_filt;
            float acc = 0.f;
            for (int j = 0; j < a.n_filt; ++j) acc = fmaf(fmaxf(lm[j], thr), drow[j], acc);
            ceps[i] = acc;
        }
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
        for (int i = tid; i < (db - da) * nc; i += 256) {
            const int r = i / nc, q = i - r * nc;
            const int u = da + r;
            float acc = 0.f;
            for (int m = 1; m <= Nd; ++m) {
                const int up = min(u + m, T - 1), um = max(u - m, 0);
                acc += (float)m * (ceps[(size_t)(up - ta) * nc + q] - ceps[(size_t)(um - ta) * nc + q]);
            }
            dlt[i] = acc * inv_den;
        }
        __syncthreads();
    }
    if (a.delta_order >= 2) {
        for (int i = tid; i < n * nc; i += 256) {
            const int r = i / nc, q = i - r * nc;
            const int u = t0 + r;
            float acc = 0.f;
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
        for (int d = wave; d < D; d += 4) {
            float s = 0.f;
            for (int r = lane; r < n; r += 64) s += value(r, d);
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s / (float)n;
            float v = 0.f;
            for (int r = lane; r < n; r += 64) {
                const float e = value(r, d) - mean;
                v = fmaf(e, e, v);
            }
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            float sd = sqrtf(v / (float)n);
            if (sd < 10.0f * 1.1920929e-07f) sd = 1.0f;
            if (lane == 0) {
                stats[d] = mean;
                stats[D + d] = 1.0f / sd;
            }
        }
        __syncthreads();
    }

    float* __restrict__ out = a.out + (size_t)(f0 + t0) * D;
    for (int i = tid; i < n * D; i += 256) {
        const int r = i / D, d = i - r * D;
        float v = value(r, d);
        if (a.cmvn) v = (v - stats[d]) * stats[D + d];
        out[i] = v;
    }
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
    for (int64_t i = tid; i < T * n_filt; i += 256) mx = fmaxf(mx, x[i]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    const float thr = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - top_db;
    for (int64_t i = tid; i < T * nc; i += 256) {
        const int64_t r = i / nc;
        const int q = (int)(i - r * nc);
        const float* __restrict__ row = x + r * n_filt;
        const float* __restrict__ drow = dct + (size_t)q * n_filt;
        float acc = 0.f;
        for (int j = 0; j < n_filt; ++j) acc = fmaf(fmaxf(row[j], thr), drow[j], acc);
        out[(f0 + r) * nc + q] = acc;
    }
}

int launch_topdb_dct(const float* logmel, const int64_t* frame_off_dev, int64_t n_utt, int n_filt, int n_ceps, const float* dct,
                     float top_db, float* out, hipStream_t stream) {
    if (n_utt <= 0) return SSP_OK;
    if (n_utt > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "top_db: too many utterances");
    hipLaunchKernelGGL(topdb_dct_kernel, dim3((unsigned)n_utt), dim3(256), 0, stream, logmel, frame_off_dev, n_filt, n_ceps, dct, top_db, out);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

int launch_mfcc_generic(const MfccArgs& args, int n_chunks, size_t lds_bytes, hipStream_t stream) {
    if (n_chunks <= 0) return SSP_OK;
    if (lds_bytes > 64 * 1024) {
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_generic_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    hipLaunchKernelGGL(mfcc_generic_kernel, dim3(n_chunks), dim3(256), lds_bytes, stream, args);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

}  // namespace ssp

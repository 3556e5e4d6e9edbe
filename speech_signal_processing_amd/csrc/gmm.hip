// Diagonal-covariance GMM log-likelihood scoring on gfx950 MFMA.
//
// Replaces the S x U Python loop  pred[j,i] = GMM[i].score(x_j) - UBM.score(x_j)  (GMM_UBM.py:181-197) and sklearn's
// GaussianMixture.score_samples / score for covariance_type='diag' (sk:mixture/_gaussian_mixture.py:453-512,
// sk:mixture/_base.py:337-373): every frame is scored against every mixture of every model in one launch.
//
// Formulation: lp[t,k] = sum_j aug[t][j] * W[k][j] with aug = [x, x^2, 1] and
//   W[k] = [mu*P, -P/2, ln w_k + 1/2 sum ln P - 1/2 (D ln 2pi + sum mu^2 P)],  P = 1/sigma^2,
// i.e. one (mixtures x 2D+1) . (2D+1 x frames) contraction on v_mfma_f32_32x32x2_f32 (exact fp32), followed by a
// log-sum-exp over the mixtures of each model.  Mixtures are the MFMA ROWS and frames the COLUMNS, so the 16 accumulator
// registers of a lane all belong to ONE frame: the LSE is in-lane plus one exchange with lane^32.
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace ssp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// async global -> LDS copy of one packed row tile (NQ pieces of 1 KiB; each wave moves whole pieces, the image is
// linear so the LDS destination is wave-uniform base + lane * 16)
template <int NQ>
__device__ __forceinline__ void stage_tile(const float* __restrict__ tile, float* dst, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < (NQ + 3) / 4; ++p) {
        const int piece = wave + 4 * p;
        if (piece < NQ)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(tile + piece * 256 + lane * 4), (lds_ptr_t)(dst + piece * 256), 16, 0, 0);
    }
}

struct GmmArgs {
    const float* feats;   // [F x D]
    const float* wimg;    // [n_tiles][NQ][2][32][4]  packed row tiles (32 mixtures each)
    const char* wimg16;   // bf16x3 path: [n_tiles][NK][hi|lo][2][32][8 bf16] + [2][16] fp32 constants + pad (see pack)
    float* llT;           // [n_models x F] model-major per-frame log-likelihood (FUSED == false)
    int64_t F;            // total frames (rows of feats)
    int32_t D, n_models, tiles_per_model, n_tiles;
    // fused per-utterance epilogue (FUSED == true): a wave's 64 frames are cut at utterance boundaries into pieces; the sum of a
    // model's log-likelihood over a piece goes to partial[piece][model] (fixed in-wave order: bit-reproducible) and a small second
    // kernel adds an utterance's pieces in order — the [n_models x F] matrix never exists
    const int64_t* frame_off;   // [n_utt + 1] absolute frame offsets of this batch's utterances (device)
    const int32_t* piece_base;  // [n_utt + 1] first piece id of every utterance of the batch
    double* partial;            // [n_pieces x n_models], float64: the order in which an utterance's frames meet (which depends on where
                                // the utterance sits in the batch) then has no visible effect on the fp32 mean
    int64_t frame_base;         // absolute index of the batch's first frame
    int32_t n_utt;
    // candidate re-scoring (fp32 kernel only): workgroup b scores only the models block_models[b * bl_stride + 1 ..] (count in entry 0;
    // -1 or a null table: every model) — their tiles in list order, partial sums under their real model index
    const int32_t* block_models;
    int32_t bl_stride;
};

// sum over lanes 0..31 of a value that is zero on lanes 32..63, in float64 and in a fixed order: inclusive DPP row scans, then the two
// row totals
template <int CTRL>
__device__ __forceinline__ double row_shr_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double sum_lower_half(double v) {
    v += row_shr_f64<0x111>(v);
    v += row_shr_f64<0x112>(v);
    v += row_shr_f64<0x114>(v);
    v += row_shr_f64<0x118>(v);
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    const long long r0 = ((long long)__builtin_amdgcn_readlane(hi, 15) << 32) | (unsigned int)__builtin_amdgcn_readlane(lo, 15);
    const long long r1 = ((long long)__builtin_amdgcn_readlane(hi, 31) << 32) | (unsigned int)__builtin_amdgcn_readlane(lo, 31);
    return __builtin_bit_cast(double, r0) + __builtin_bit_cast(double, r1);
}

// Per-wave bookkeeping of the fused epilogue: which utterances the wave's 64 frames belong to.
template <int CT>
struct PieceMap {
    int utt[CT];    // utterance (batch-relative) of this lane's frame in column tile ct; -1 on the upper lane half / past the batch
    int u_lo, np;   // first utterance overlapping the wave's span, utterances overlapping it (wave-uniform)
    int pid0;       // piece id of u_lo's piece in this span
    __device__ __forceinline__ void init(const GmmArgs& a, int64_t span0 /*batch frame index of the span's first frame*/, int fl, int h) {
        u_lo = 0;
        np = 0;
        pid0 = -1;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) utt[ct] = -1;
        if (span0 >= a.F) return;
        const int64_t g0 = a.frame_base + span0;
        // last utterance that starts at or before g0 and is not empty-before-g0: upper_bound(frame_off, g0) - 1
        int lo = 0, hi = a.n_utt;  // invariant: frame_off[lo] <= g0 < frame_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.frame_off[mid] <= g0) lo = mid; else hi = mid;
        }
        u_lo = lo;
        const int64_t span_end = min(span0 + 32 * CT, a.F) + a.frame_base;
        int u_hi = u_lo;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int64_t g = g0 + ct * 32 + fl;
            int u = u_lo;
            if (g < span_end) {
                while (a.frame_off[u + 1] <= g) ++u;
                if (h == 0) utt[ct] = u;
                u_hi = max(u_hi, u);
            }
        }
        // (lanes hold increasing utterance ids: the largest sits on the span's last valid frame; a max over the wave finds it)
        for (int o = 32; o > 0; o >>= 1) u_hi = max(u_hi, __shfl_xor(u_hi, o));
        u_hi = __builtin_amdgcn_readfirstlane(u_hi);
        np = u_hi - u_lo + 1;
        pid0 = a.piece_base[u_lo] + (int)(span0 / (32 * CT) - (a.frame_off[u_lo] - a.frame_base) / (32 * CT));
    }
    // piece id of the p-th utterance of the span (p > 0: the utterance starts inside the span, its first piece); -1: no frames
    __device__ __forceinline__ int pid(const GmmArgs& a, int p) const {
        if (p == 0) return pid0;
        const int u = u_lo + p;
        return a.frame_off[u + 1] > a.frame_off[u] ? a.piece_base[u] : -1;
    }
    template <class LL>
    __device__ __forceinline__ void emit(const GmmArgs& a, int model, const LL& ll, int lane) const {
        for (int p = 0; p < np; ++p) {
            const int u = u_lo + p;
            double v = 0.0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) v += utt[ct] == u ? (double)ll[ct] : 0.0;
            const double tot = sum_lower_half(v);
            const int id = pid(a, p);
            if (lane == 0 && id >= 0) a.partial[(size_t)id * a.n_models + model] = tot;
        }
    }
};

// Online log-sum-exp over one 32x32 accumulator tile (16 mixtures per lane).  The packed weights carry a factor log2(e),
// so the accumulator is log2 p: exponentials are bare v_exp_f32, the result is converted back with one multiply by ln 2.
__device__ __forceinline__ void lse2_update(const f32x16& acc, float& run_m, float& run_s) {
#ifdef SSP_GMM_ABL_NOLSE  // ablation (wrong results): the MFMAs alone
    {
        run_s += acc[0] + acc[15];
        run_m = 0.f;
        return;
    }
#endif
#ifdef SSP_GMM_ABL_NOMAX  // ablation (wrong results): what the running maximum and the subtraction cost
    {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            s0 += __builtin_amdgcn_exp2f(acc[i]);
            s1 += __builtin_amdgcn_exp2f(acc[i + 1]);
        }
        run_s += s0 + s1;
        run_m = 0.f;
        return;
    }
#endif
    float tm = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) tm = fmaxf(fmaxf(tm, acc[i]), acc[i + 1]);
    tm = fmaxf(tm, acc[15]);
    const float nm = fmaxf(run_m, tm);
    // (scalar subtract / add: beside MFMAs a packed v_pk_add_f32 costs the SIMD's vector issue more than the two instructions it
    //  replaces — 42.8 -> 40.4 ms on configs[2]; the file is built with -fno-slp-vectorize so the compiler does not re-pack them)
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        s0 += __builtin_amdgcn_exp2f(acc[i] - nm);
        s1 += __builtin_amdgcn_exp2f(acc[i + 1] - nm);
    }
    run_s = run_s * __builtin_amdgcn_exp2f(run_m - nm) + (s0 + s1);
    run_m = nm;
}

// final value of a model for one frame: combine the two lane halves, back to natural log
__device__ __forceinline__ float lse2_finish(float run_m, float run_s) {
    const float m2 = __shfl_xor(run_m, 32), s2 = __shfl_xor(run_s, 32);
    const float mm = fmaxf(run_m, m2);
    const float ss = run_s * __builtin_amdgcn_exp2f(run_m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
    return (mm + __builtin_amdgcn_logf(ss)) * 0.6931471805599453f;
}

// this workgroup's frames (contiguous rows, n_floats = frames x D, a multiple of 4) into LDS: up to K float4 per thread, all in
// flight together (a dword per loop trip with a wait behind each exposed ~D memory latencies per workgroup); zero beyond `tot`
template <int K>
__device__ __forceinline__ void stage_frames(float* __restrict__ xs, const float* __restrict__ src, int tot, int n_floats, int tid) {
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int n4 = n_floats >> 2;
        float4 v[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int i4 = tid + 256 * u, i = 4 * i4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i4 < n4) {
                if (i + 3 < tot) {
                    v[u] = *reinterpret_cast<const float4*>(src + i);
                } else {
                    if (i < tot) v[u].x = src[i];
                    if (i + 1 < tot) v[u].y = src[i + 1];
                    if (i + 2 < tot) v[u].z = src[i + 2];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int i4 = tid + 256 * u;
            if (i4 < n4) *reinterpret_cast<float4*>(xs + 4 * i4) = v[u];
        }
        for (int i4 = tid + 256 * K; i4 < n4; i4 += 256)  // (never taken while D <= 4 K)
            for (int c = 0; c < 4; ++c) xs[4 * i4 + c] = 4 * i4 + c < tot ? src[4 * i4 + c] : 0.f;
    } else {
        for (int i = tid; i < n_floats; i += 256) xs[i] = i < tot ? src[i] : 0.f;
    }
}

// NQ = k-depth / 8 of the packed image (k-depth >= 2D+1); CT = 32-frame column tiles per wave
template <int NQ, int CT, bool FUSED>
__global__ __launch_bounds__(256) void gmm_loglik_kernel(GmmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_FLOATS = NQ * 2 * 32 * 4;
    constexpr int FRAMES_WG = 4 * CT * 32;
    float* wbuf = reinterpret_cast<float*>(smem);                   // [2][TILE_FLOATS]
    float* xs = reinterpret_cast<float*>(smem);                     // [FRAMES_WG * D], dead once the B operand sits in registers: the
                                                                    // tile ring reuses its bytes (40 instead of 60 KiB at D = 39: three
                                                                    // workgroups per CU, which the 161 registers allow)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 31, h = lane >> 5;
    const int D = a.D;
    const int64_t f0 = (int64_t)blockIdx.x * FRAMES_WG;
    const int n_valid = (int)min((int64_t)FRAMES_WG, a.F - f0);

    // stage this workgroup's frames (contiguous rows) into LDS, coalesced
    {
        const float* __restrict__ src = a.feats + f0 * D;
        const int tot = n_valid * D;
        stage_frames<CT * NQ / 2 + 1>(xs, src, tot, FRAMES_WG * D, tid);
    }
    __syncthreads();

    // B operand (this wave's frames) in registers: b[ct][q][e] = aug[frame][8q + 2e + h]
    float b[CT][NQ][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float* xr = xs + (size_t)((wave * CT + ct) * 32 + fl) * D;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 8 * q + 2 * e + h;
                const int jj = j < D ? j : (j < 2 * D ? j - D : 0);
                const float v = xr[jj];
                b[ct][q][e] = j < D ? v : (j < 2 * D ? v * v : (j == 2 * D ? 1.0f : 0.0f));
            }
    }

    PieceMap<CT> pm;
    if (FUSED) pm.init(a, f0 + (int64_t)wave * CT * 32, fl, h);
    __syncthreads();  // every wave has its frames in registers: the staging bytes become the tile ring
    // which row tiles this workgroup walks: all of them, or (candidate re-scoring) the tiles of its listed models only
    const int32_t* bl = a.block_models ? a.block_models + (size_t)blockIdx.x * a.bl_stride : nullptr;
    const int n_list = bl ? __builtin_amdgcn_readfirstlane(bl[0]) : -1;
    const int tpm = a.tiles_per_model;
    const int n_it = n_list >= 0 ? n_list * tpm : a.n_tiles;
    auto tile_at = [&](int r) -> int {
        if (n_list < 0) return r;
        const int k = r / tpm;
        return __builtin_amdgcn_readfirstlane(bl[1 + k]) * tpm + (r - k * tpm);
    };
    if (n_it == 0) return;  // (wave-uniform over the whole workgroup: no barrier is left waiting)
    stage_tile<NQ>(a.wimg + (size_t)tile_at(0) * TILE_FLOATS, wbuf, wave, lane);  // the first row tile
    __syncthreads();

    float run_m[CT], run_s[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        run_m[ct] = -INFINITY;
        run_s[ct] = 0.f;
    }

    int rt = 0, model = 0;
    for (int r = 0; r < n_it; ++r) {
        const float* wcur = wbuf + (r & 1) * TILE_FLOATS;
        // tile r+1 streams into the other buffer while this one feeds the MFMAs (the barrier at the end of the
        // iteration drains the LDS-DMA: __syncthreads() waits vmcnt(0))
#ifndef SSP_GMM_ABL_NODMA  // (ablation, wrong results: every tile computes on the first one's bytes — what issuing the tile stream costs)
        if (r + 1 < n_it)
            stage_tile<NQ>(a.wimg + (size_t)tile_at(r + 1) * TILE_FLOATS, wbuf + ((r + 1) & 1) * TILE_FLOATS, wave, lane);
#endif
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(wcur + ((q * 2 + h) * 32 + fl) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b[ct][q][e], acc[ct], 0, 0, 0);
        }
        // online log-sum-exp over this tile's 16 mixtures per lane
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) lse2_update(acc[ct], run_m[ct], run_s[ct]);
        if (++rt == a.tiles_per_model) {
            const int mid = n_list >= 0 ? __builtin_amdgcn_readfirstlane(bl[1 + model]) : model;  // the model these tiles belong to
            float llv[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                llv[ct] = lse2_finish(run_m[ct], run_s[ct]);
                const int fidx = (wave * CT + ct) * 32 + fl;
                if (!FUSED && h == 0 && fidx < n_valid) a.llT[(size_t)mid * a.F + f0 + fidx] = llv[ct];
                run_m[ct] = -INFINITY;
                run_s[ct] = 0.f;
            }
            if (FUSED) pm.emit(a, mid, llv, lane);
            rt = 0;
            ++model;
        }
#ifdef SSP_GMM_ABL_NOBARRIER  // ablation (racy, wrong results): what the workgroup barrier per row tile costs (the wave waits for its own DMA only)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        __syncthreads();
#endif
    }
}

// ---- bf16 x 3 split-precision path (precision = 1) ---------------------------------------------------------------
// Same contraction on v_mfma_f32_32x32x16_bf16 (16x the per-clock rate of the fp32-input MFMA): every operand is split
// into hi = bf16(v) and lo = bf16(v - hi) and the product is accumulated as  Wh.ah + Wh.al + Wl.ah  in fp32 (the lo.lo
// term, 2^-16 relative, is dropped) — 3 bf16 MFMAs per k-step instead of 8 fp32 ones.  The additive constant of every
// mixture stays exact: it is the fp32 INITIAL VALUE of the accumulator, not a matrix column.  aug = [x, x^2].
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NK>
__device__ __forceinline__ void stage_tile16(const char* __restrict__ tile, char* dst, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < (2 * NK + 3) / 4; ++p) {
        const int piece = wave + 4 * p;
        if (piece < 2 * NK)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(tile + piece * 1024 + lane * 16), (lds_ptr_t)(dst + piece * 1024), 16, 0, 0);
    }
    if (wave == 3)  // the 64 constants (2 x 16 used + pad) ride along as one 256-byte piece
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(tile + 2 * NK * 1024 + lane * 4), (lds_ptr_t)(dst + 2 * NK * 1024), 4, 0, 0);
}

template <int NK, int CT, bool FUSED>
__global__ __launch_bounds__(256) void gmm_loglik_bf16x3_kernel(GmmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_BYTES = 2 * NK * 1024 + 256;
    constexpr int GROUP = 2;                       // row tiles staged (and consumed) per workgroup barrier
    constexpr int GROUP_BYTES = GROUP * TILE_BYTES;
    constexpr int FRAMES_WG = 4 * CT * 32;
    // LDS: a ring of [2][GROUP_BYTES] tile slots; its first bytes stage the frames once
    float* xs = reinterpret_cast<float*>(smem);          // [FRAMES_WG * D] (dead before the first tile lands)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, h = lane >> 5;
    const int D = a.D;
    const int64_t f0 = (int64_t)blockIdx.x * FRAMES_WG;
    const int n_valid = (int)min((int64_t)FRAMES_WG, a.F - f0);
    {
        const float* __restrict__ src = a.feats + f0 * D;
        const int tot = n_valid * D;
        stage_frames<CT * NK + 1>(xs, src, tot, FRAMES_WG * D, tid);
    }
    __syncthreads();

    // B operand: lane (frame fl, half h) holds aug[frame][16 ks + 8 h + j], j = 0..7, as hi and lo bf16 fragments
    bf16x8 bh[CT][NK], bl[CT][NK];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float* xr = xs + (size_t)((wave * CT + ct) * 32 + fl) * D;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int ai = 16 * ks + 8 * h + jj;
                const float xv = xr[ai < D ? ai : (ai < 2 * D ? ai - D : 0)];
                const float v = ai < D ? xv : (ai < 2 * D ? xv * xv : 0.0f);
                const __bf16 hi = (__bf16)v;
                bh[ct][ks][jj] = hi;
                bl[ct][ks][jj] = (__bf16)(v - (float)hi);
            }
    }
    PieceMap<CT> pm;
    if (FUSED) pm.init(a, f0 + (int64_t)wave * CT * 32, fl, h);
    __syncthreads();  // every wave has its fragments: the staging area becomes the tile ring
    // stage the (up to GROUP) tiles of group g into ring slot `slot` (offsets are formed from the LDS base directly so
    // the address stays in the LDS address space)
#define SSP_STAGE_GROUP(g_, slot_)                                                                                        \
    {                                                                                                                     \
        _Pragma("unroll") for (int t_ = 0; t_ < GROUP; ++t_) {                                                            \
            const int tile_ = (g_) * GROUP + t_;                                                                          \
            if (tile_ < a.n_tiles)                                                                                        \
                stage_tile16<NK>(a.wimg16 + (size_t)tile_ * TILE_BYTES, smem + (slot_) * GROUP_BYTES + t_ * TILE_BYTES, wave, lane); \
        }                                                                                                                 \
    }
    SSP_STAGE_GROUP(0, 0)
    __syncthreads();

    float run_m[CT], run_s[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        run_m[ct] = -INFINITY;
        run_s[ct] = 0.f;
    }
    int rt = 0, model = 0;
    const int n_groups = (a.n_tiles + GROUP - 1) / GROUP;
    for (int g = 0; g < n_groups; ++g) {
        if (g + 1 < n_groups) SSP_STAGE_GROUP(g + 1, (g + 1) & 1)
#pragma unroll
        for (int t = 0; t < GROUP; ++t) {
            if (g * GROUP + t >= a.n_tiles) break;
            const char* wcur = smem + (g & 1) * GROUP_BYTES + t * TILE_BYTES;
            f32x16 acc[CT];
            {
                const f32x4* ci = reinterpret_cast<const f32x4*>(wcur + 2 * NK * 1024 + h * 64);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x4 cv = ci[c4];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        acc[ct][4 * c4 + 0] = cv[0];
                        acc[ct][4 * c4 + 1] = cv[1];
                        acc[ct][4 * c4 + 2] = cv[2];
                        acc[ct][4 * c4 + 3] = cv[3];
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wcur + (((ks * 2 + 0) * 2 + h) * 32 + fl) * 16);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wcur + (((ks * 2 + 1) * 2 + h) * 32 + fl) * 16);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#ifdef SSP_GMM_ABL_NOMFMA  // ablation (wrong results): the epilogue alone
                    acc[ct][ks] += (float)al[0] * (float)bh[ct][ks][0] + (float)ah[1] * (float)bl[ct][ks][1];
#else
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ct][ks], acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ct][ks], acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ct][ks], acc[ct], 0, 0, 0);
#endif
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) lse2_update(acc[ct], run_m[ct], run_s[ct]);
            if (++rt == a.tiles_per_model) {
                float llv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    llv[ct] = lse2_finish(run_m[ct], run_s[ct]);
                    const int fidx = (wave * CT + ct) * 32 + fl;
                    if (!FUSED && h == 0 && fidx < n_valid) a.llT[(size_t)model * a.F + f0 + fidx] = llv[ct];
                    run_m[ct] = -INFINITY;
                    run_s[ct] = 0.f;
                }
                if (FUSED) pm.emit(a, model, llv, lane);
                rt = 0;
                ++model;
            }
        }
        __syncthreads();
    }
}

// per-utterance mean over frames (GaussianMixture.score), score differences against the UBM and arg-max
// (GMM_UBM.py:185-187).  One workgroup per utterance, fixed summation order => bitwise reproducible.
__global__ __launch_bounds__(256) void gmm_utt_reduce_kernel(const float* __restrict__ llT, int64_t F,
                                                             const int64_t* __restrict__ frame_off, int64_t frame_base,
                                                             int n_models, int has_ubm, float* __restrict__ scores,
                                                             int32_t* __restrict__ argmax_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t a0 = frame_off[u] - frame_base;  // column of the utterance's first frame in this batch's llT
    const int T = (int)(frame_off[u + 1] - frame_off[u]);
    for (int m = wave; m < n_models; m += 4) {
        const float* __restrict__ p = llT + (size_t)m * F + a0;
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += p[t];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float v = s / (float)T;  // T == 0 -> NaN (numpy mean of empty)
        if (lane == 0) {
            sc[m] = v;
            if (scores) scores[(size_t)u * n_models + m] = v;
        }
    }
    __syncthreads();
    if (wave == 0 && argmax_out) {
        const float base = has_ubm ? sc[0] : 0.f;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int m = has_ubm + lane; m < n_models; m += 64) {
            const float v = sc[m] - base;
            if (v > best || bi == 0x7fffffff) {
                best = v;
                bi = m - has_ubm;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) {
                best = ob;
                bi = oi;
            }
        }
        if (lane == 0) argmax_out[u] = bi == 0x7fffffff ? 0 : bi;
    }
}

// fused path, second kernel: an utterance's score under model m = (sum of its pieces' partial sums, in piece order) / T, then the score
// differences against the UBM, their arg-max (first index on ties, numpy) and the margin between the best and the second best
// difference.  One workgroup per utterance; partial is [piece][model], so a piece's row is read coalesced.
__global__ __launch_bounds__(256) void gmm_piece_reduce_kernel(const double* __restrict__ partial, const int64_t* __restrict__ frame_off,
                                                               const int32_t* __restrict__ piece_base, int n_models, int has_ubm,
                                                               float* __restrict__ scores, int32_t* __restrict__ argmax_out,
                                                               float* __restrict__ margin_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = (int)(frame_off[u + 1] - frame_off[u]);
    const int p0 = piece_base[u], p1 = piece_base[u + 1];
    for (int m = tid; m < n_models; m += 256) {
        double s = 0.0;
        for (int p = p0; p < p1; ++p) s += partial[(size_t)p * n_models + m];
        const float v = (float)(s / (double)T);  // T == 0 -> NaN (numpy mean of empty)
        sc[m] = v;
        if (scores) scores[(size_t)u * n_models + m] = v;
    }
    __syncthreads();
    if (wave == 0 && (argmax_out || margin_out)) {
        const float base = has_ubm ? sc[0] : 0.f;
        float best = -INFINITY, second = -INFINITY;
        int bi = 0x7fffffff;
        for (int m = has_ubm + lane; m < n_models; m += 64) {
            const float v = sc[m] - base;
            if (v > best || bi == 0x7fffffff) {
                second = bi == 0x7fffffff ? second : best;
                best = v;
                bi = m - has_ubm;
            } else if (v > second) {
                second = v;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o), os = __shfl_xor(second, o);
            const int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) {
                second = bi == 0x7fffffff ? os : fmaxf(best, os);
                best = ob;
                bi = oi;
            } else if (oi != 0x7fffffff) {
                second = fmaxf(second, ob);
            }
        }
        if (lane == 0) {
            if (argmax_out) argmax_out[u] = bi == 0x7fffffff ? 0 : bi;
            if (margin_out) margin_out[u] = best - second;  // +inf with one speaker model, NaN when scores are NaN
        }
    }
}

// ---- bf16x3 close calls: utterances whose top-2 margin is within the split-precision error band are listed IN ORDER (one workgroup,
// ballot prefix: deterministic) for re-scoring on the exact fp32 path
// The band: what the split-precision scores of TWO models can differ by, relative to each other, from the fp32 path's.  A mixture's
// exponent is c_k + sum_j W_kj aug_j over aug = [x, x^2] (2 D terms; c_k is the accumulator's exact initial value on both paths).  With
// W and aug split hi + lo (two round-to-nearest bf16 steps: |v - hi - lo| <= 2^-18 |v|) the products left out sum to at most
// 3.01 2^-18 sum |W_kj| |aug_j|; bf16 products are exact in fp32, and the accumulation of the 6 D terms rounds (or truncates: the matrix
// core's internal order is undocumented, so 2^-23 per term) at most 6 D 2^-23 sum |W| |aug| (1 + 2^-7); the fp32 path's own 2 D terms
// another 2 D 2^-23.  sum_j |W_kj| |aug_j| <= S(x) = sum_d |x_d| A_d + x_d^2 B_d with A_d = max |mu P|, B_d = max P / 2 over every mixture
// of every model (packed once).  The log-sum-exp moves by at most the largest exponent error (it is 1-Lipschitz in the max norm) plus
// its own evaluation noise (a few ulp of its magnitude on either path: 2^-20 (max_m |score_m| + 1) covers both), the utterance mean by the mean:
//   |score_bf16x3 - score_fp32| <= eps mean_t S(x_t) + 2^-20 (|score| + 1),  eps = 3.01 2^-18 + 8 D 2^-23 1.01
// and a margin between two models by twice that.  One workgroup per utterance.
__global__ __launch_bounds__(256) void gmm_band_kernel(const float* __restrict__ feats, const int64_t* __restrict__ frame_off, int D,
                                                       const float* __restrict__ tab, float eps, const float* __restrict__ scores,
                                                       int n_models, float* __restrict__ band) {
    __shared__ float red[4], redm[4];
    const int u = blockIdx.x, tid = threadIdx.x;
    if (eps < 0.f) {  // the calibrated (heuristic) band
        if (tid == 0) band[u] = 8.0e-5f * (fabsf(scores[(size_t)u * n_models]) + 1.0f);
        return;
    }
    const int64_t a0 = frame_off[u], T = frame_off[u + 1] - a0;
    const float* __restrict__ x = feats + a0 * D;
    // thread = (row phase, column): its two table entries stay in registers and a sweep of the workgroup reads whole consecutive rows
    const int R = 256 / D;  // rows per sweep (D <= 127)
    const int col = tid % D, ph = tid / D;
    float s = 0.f;
    if (ph < R) {
        const float ta = tab[col], tb = tab[D + col];
        int64_t r = ph;
        for (; r + 3 * R < T; r += 4 * R) {  // four loads in flight
            const float v0 = x[r * D + col], v1 = x[(r + R) * D + col], v2 = x[(r + 2 * R) * D + col], v3 = x[(r + 3 * R) * D + col];
            s = fmaf(fabsf(v0), ta, fmaf(v0 * v0, tb, s));
            s = fmaf(fabsf(v1), ta, fmaf(v1 * v1, tb, s));
            s = fmaf(fabsf(v2), ta, fmaf(v2 * v2, tb, s));
            s = fmaf(fabsf(v3), ta, fmaf(v3 * v3, tb, s));
        }
        for (; r < T; r += R) {
            const float v = x[r * D + col];
            s = fmaf(fabsf(v), ta, fmaf(v * v, tb, s));
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    // the log-sum-exp noise term takes the LARGEST |score| of the utterance's row (a speaker model's score may be larger than model 0's)
    float mg = 0.f;
    for (int m = tid; m < n_models; m += 256) mg = fmaxf(mg, fabsf(scores[(size_t)u * n_models + m]));
    for (int o = 32; o > 0; o >>= 1) mg = fmaxf(mg, __shfl_xor(mg, o));
    if ((tid & 63) == 0) {
        red[tid >> 6] = s;
        redm[tid >> 6] = mg;
    }
    __syncthreads();
    if (tid == 0) {
        const float S = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(T > 0 ? T : 1);
        const float mag = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3])) + 1.0f;
        band[u] = 2.0f * (eps * S * 1.001f + 0x1p-20f * mag);
    }
}

// ---- bf16x3 close calls: utterances whose top-2 margin is within the split-precision error band are listed IN ORDER (one workgroup,
// ballot prefix: deterministic) for re-scoring on the exact fp32 path
__global__ __launch_bounds__(256) void gmm_flag_kernel(const float* __restrict__ margin, const float* __restrict__ band, int n_utt,
                                                       int32_t* __restrict__ list, int32_t* __restrict__ count) {
    __shared__ int s_base, s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int u0 = 0; u0 < n_utt; u0 += 256) {
        const int u = u0 + tid;
        bool f = false;
        if (u < n_utt) f = !(margin[u] >= band[u]);  // close call, or not comparable (NaN)
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (f) list[off + __popcll(b & ((1ull << lane) - 1ull))] = u;
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) *count = s_base;
}

// ---- which models of a listed (close-call) utterance can still win on the fp32 path: every speaker model whose split-precision score
// difference is within the band of the best one (the band is twice the bound on a score's error: a model further behind cannot catch
// up), plus the UBM (the differences are formed against its fp32 score).  cand[i] = (count, models ...); count -1: every model (more
// than CM - 1 candidates, or scores that are not numbers).  One workgroup per listed utterance.
constexpr int GMM_CAND = 16;
__global__ __launch_bounds__(256) void gmm_candidates_kernel(const int32_t* __restrict__ list, const float* __restrict__ scores,
                                                             const float* __restrict__ band, int n_models, int has_ubm,
                                                             int32_t* __restrict__ cand) {
    __shared__ float redf[4];
    __shared__ int redi[4], s_cnt;
    const int i = blockIdx.x, u = list[i], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ row = scores + (size_t)u * n_models;
    const float base = has_ubm ? row[0] : 0.f;
    float best = -INFINITY;
    int bad = 0;
    for (int m = has_ubm + tid; m < n_models; m += 256) {
        const float v = row[m] - base;
        bad |= !(v == v);
        best = fmaxf(best, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        best = fmaxf(best, __shfl_xor(best, o));
        bad |= __shfl_xor(bad, o);
    }
    if (lane == 0) {
        redf[wave] = best;
        redi[wave] = bad;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    best = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    bad = redi[0] | redi[1] | redi[2] | redi[3];
    const float thr = best - band[u];
    int32_t* out = cand + (size_t)i * GMM_CAND;
    if (!bad)
        for (int m = has_ubm + tid; m < n_models; m += 256)
            if (row[m] - base >= thr) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < GMM_CAND - 1 - has_ubm) out[1 + has_ubm + pos] = m;
            }
    __syncthreads();
    if (tid == 0) {
        const int c = s_cnt;
        if (has_ubm) out[1] = 0;
        out[0] = (bad || !(thr == thr) || c > GMM_CAND - 1 - has_ubm) ? -1 : c + has_ubm;
    }
}

// results of the candidate re-scoring back into the batch's outputs: the fp32 scores of the listed utterance's candidates (sub_scores
// holds garbage for models its workgroups skipped), and the arg-max by gmm_piece_reduce_kernel's rule — fp32 difference against the
// UBM, larger wins, lower index on ties — over the candidates; count -1: every model was scored, the reduce kernel's own arg-max stands
__global__ __launch_bounds__(64) void gmm_scatter_cand_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ cand, int n_models,
                                                              int has_ubm, const float* __restrict__ sub_scores,
                                                              const int32_t* __restrict__ sub_argmax, float* __restrict__ scores,
                                                              int32_t* __restrict__ argmax_out, int all_scored) {
    const int i = blockIdx.x, u = list[i], lane = threadIdx.x;
    const int32_t* c = cand + (size_t)i * GMM_CAND;
    const float* __restrict__ src = sub_scores + (size_t)i * n_models;
    const int n = all_scored ? -1 : c[0];  // (a re-scoring call that had to be cut scored every model: the whole fp32 row goes back)
    if (n < 0) {
        if (scores)
            for (int m = lane; m < n_models; m += 64) scores[(size_t)u * n_models + m] = src[m];
        if (argmax_out && lane == 0) argmax_out[u] = sub_argmax[i];
        return;
    }
    if (scores && lane < n) scores[(size_t)u * n_models + c[1 + lane]] = src[c[1 + lane]];
    if (argmax_out && lane == 0) {
        const float base = has_ubm ? src[0] : 0.f;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int k = has_ubm; k < n; ++k) {
            const int m = c[1 + k];
            const float v = src[m] - base;
            if (bi == 0x7fffffff || v > best || (v == best && m - has_ubm < bi)) {
                best = v;
                bi = m - has_ubm;
            }
        }
        argmax_out[u] = bi == 0x7fffffff ? 0 : bi;
    }
}

// rows of the listed utterances into a compact matrix (sub_off: their frame offsets in the compact matrix)
__global__ __launch_bounds__(256) void gmm_gather_frames_kernel(const float* __restrict__ feats, const int64_t* __restrict__ frame_off,
                                                                const int32_t* __restrict__ list, const int64_t* __restrict__ sub_off,
                                                                int D, float* __restrict__ out) {
    const int i = blockIdx.x, u = list[i];
    const int64_t a0 = frame_off[u], n = (frame_off[u + 1] - a0) * D;
    const float* __restrict__ src = feats + a0 * D;
    float* __restrict__ dst = out + sub_off[i] * D;
    for (int64_t k = threadIdx.x; k < n; k += 256) dst[k] = src[k];
}

// results of the re-scored utterances back into the batch's outputs
__global__ __launch_bounds__(256) void gmm_scatter_kernel(const int32_t* __restrict__ list, int n_models, const float* __restrict__ sub_scores,
                                                          const int32_t* __restrict__ sub_argmax, float* __restrict__ scores,
                                                          int32_t* __restrict__ argmax_out) {
    const int i = blockIdx.x, u = list[i];
    if (scores)
        for (int m = threadIdx.x; m < n_models; m += 256) scores[(size_t)u * n_models + m] = sub_scores[(size_t)i * n_models + m];
    if (argmax_out && threadIdx.x == 0) argmax_out[u] = sub_argmax[i];
}

template <int NQ, int CT, bool FUSED>
static int launch_loglik(const GmmArgs& a, hipStream_t s) {
    constexpr int FRAMES_WG = 4 * CT * 32;
    const size_t lds = std::max<size_t>((size_t)2 * NQ * 2 * 32 * 4, (size_t)FRAMES_WG * a.D) * sizeof(float);
    const int64_t grid = ceil_div<int64_t>(a.F, FRAMES_WG);
    if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "gmm: too many frames for one launch");
    if (lds > 64 * 1024)
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_loglik_kernel<NQ, CT, FUSED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gmm_loglik_kernel<NQ, CT, FUSED>), dim3((unsigned)grid), dim3(256), lds, s, a);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

template <int NK, int CT, bool FUSED>
static int launch_loglik16(const GmmArgs& a, hipStream_t s) {
    constexpr int FRAMES_WG = 4 * CT * 32;
    const size_t lds = std::max<size_t>((size_t)2 * 2 * (2 * NK * 1024 + 256), (size_t)FRAMES_WG * a.D * sizeof(float));
    const int64_t grid = ceil_div<int64_t>(a.F, FRAMES_WG);
    if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "gmm: too many frames for one launch");
    if (lds > 64 * 1024)
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_loglik_bf16x3_kernel<NK, CT, FUSED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gmm_loglik_bf16x3_kernel<NK, CT, FUSED>), dim3((unsigned)grid), dim3(256), lds, s, a);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

static const int kNQ[] = {4, 7, 10, 16, 24, 32};

static int pick_nq(int D) {
    const int need = (2 * D + 1 + 7) / 8;
    for (int v : kNQ)
        if (v >= need) return v;
    return -1;
}

}  // namespace ssp

struct ssp_gmm {
    ssp_ctx* ctx = nullptr;
    int32_t n_models = 0, K = 0, D = 0, has_ubm = 0;
    int32_t nq = 0, tiles_per_model = 0;
    ssp::DevBuf wimg;
    ssp::DevBuf wimg16;   // bf16 hi/lo image of the same models (precision = 1)
    int32_t nk16 = 0;     // k-depth / 16 of the bf16 image (0: D too large for the bf16 kernels)
    ssp::DevBuf scratch;  // llT when the caller does not ask for it (grow-only)
    // fused per-utterance epilogue: piece tables of the last single-batch call (cached per segments handle) + grow-only buffers
    uint64_t pb_serial = 0;
    int32_t pb_gran = 0, pb_pieces = 0;
    std::vector<int32_t> pb_host;
    ssp::DevBuf cand, blk;        // candidate re-scoring: [n_flag][GMM_CAND] per listed utterance, [n_blocks][1 + BL] per workgroup of the re-scoring launch
    std::vector<int32_t> cand_host, blk_host;
    ssp::DevBuf bound_tab, band;  // precision = 1: [2 D] largest |mu P| and P / 2 per dimension over every mixture; [n_utt] error band of the margins
    ssp::DevBuf pb_dev, partial, margin, flag_list, flag_count, sub_feats, sub_off, sub_scores, sub_argmax, sub_pb, sub_partial;
    std::vector<int32_t> sub_list_host, sub_pb_host;
    std::vector<int64_t> sub_off_host;
    int32_t last_rescored = 0;
    // precision = 4 (auto): what the pilot of the last such call saw and chose
    int32_t auto_choice = -1, auto_pilot_utts = 0, auto_pilot_listed = 0;
    float auto_predicted = 0.f;  // predicted cost of the proven-band split path, in units of the fp32 path's
    bool auto_late = false;      // precision 4 on a batch too small for a pilot: the split pass runs, the re-scoring is priced from the FULL lists
};

using namespace ssp;

extern "C" {

int ssp_gmm_pack(ssp_ctx* ctx, int32_t n_models, int32_t K, int32_t D, const double* weights, const double* means,
                 const double* covars, int32_t has_ubm, ssp_gmm** out) {
    ssp::TraceRange trace_("ssp_gmm_pack");
    if (!out) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_pack: null out");
    *out = nullptr;
    SSP_TRY(use_ctx(ctx));
    if (n_models < 1 || K < 1 || D < 1 || !weights || !means || !covars)
        SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_pack: bad shape or null parameter array");
    if (has_ubm && n_models < 2) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_pack: has_ubm needs at least one speaker model");
    const int nq = pick_nq(D);
    if (nq < 0) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_gmm_pack: D=%d exceeds the supported feature dimension (127)", D);
    const int tpm = (K + 31) / 32;
    const size_t tile_floats = (size_t)nq * 2 * 32 * 4;
    const size_t n_tiles = (size_t)n_models * tpm;
    std::vector<float> img(n_tiles * tile_floats, 0.f);
    const double ln2pi = std::log(2.0 * M_PI);
    const double LOG2E = 1.4426950408889634;
    std::vector<double> w((size_t)nq * 8);
    for (int m = 0; m < n_models; ++m)
        for (int k = 0; k < tpm * 32; ++k) {
            std::fill(w.begin(), w.end(), 0.0);
            if (k < K) {
                const double* mu = means + ((size_t)m * K + k) * D;
                const double* cv = covars + ((size_t)m * K + k) * D;
                const double wk = weights[(size_t)m * K + k];
                double c = std::log(wk) - 0.5 * D * ln2pi;
                for (int d = 0; d < D; ++d) {
                    if (!(cv[d] > 0.0)) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_pack: non-positive covariance (model %d, mix %d)", m, k);
                    const double P = 1.0 / cv[d];
                    w[d] = mu[d] * P * LOG2E;   // everything in log2 units: the kernels exponentiate with bare v_exp_f32
                    w[D + d] = -0.5 * P * LOG2E;
                    c += 0.5 * std::log(P) - 0.5 * mu[d] * mu[d] * P;
                }
                w[2 * D] = c * LOG2E;
            } else {
                w[2 * D] = -1.0e30;  // padded mixture: contributes exp(-1e30 - max) = 0 to the LSE
            }
            float* tile = img.data() + ((size_t)m * tpm + k / 32) * tile_floats;
            const int mix = k & 31;
            for (int j = 0; j < nq * 8; ++j) {
                const int q = j >> 3, e = (j & 7) >> 1, hh = j & 1;
                tile[((size_t)(q * 2 + hh) * 32 + mix) * 4 + e] = (float)w[j];
            }
        }
    // ---- bf16 hi/lo image: [tile][ks][hi|lo][h][mix][8] bf16, then [h][16] fp32 constants in accumulator order + pad
    auto to_bf16 = [](double v) -> uint16_t {  // round to nearest even on the fp32 bit pattern
        float f = (float)v;
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto from_bf16 = [](uint16_t b) -> double {
        uint32_t u = (uint32_t)b << 16;
        float f;
        memcpy(&f, &u, 4);
        return (double)f;
    };
    const int nk16_need = (2 * D + 15) / 16;
    int nk16 = 0;
    for (int v : {1, 2, 3, 4, 5, 6, 8})
        if (v >= nk16_need) {
            nk16 = v;
            break;
        }
    std::vector<unsigned char> img16;
    if (nk16 > 0) {
        const size_t tb = (size_t)2 * nk16 * 1024 + 256;
        img16.assign(n_tiles * tb, 0);
        for (int m = 0; m < n_models; ++m)
            for (int k = 0; k < tpm * 32; ++k) {
                unsigned char* tile = img16.data() + ((size_t)m * tpm + k / 32) * tb;
                const int mix = k & 31;
                double cst = -1.0e30;
                std::vector<double> wv((size_t)nk16 * 16, 0.0);
                if (k < K) {
                    const double* mu = means + ((size_t)m * K + k) * D;
                    const double* cv = covars + ((size_t)m * K + k) * D;
                    cst = std::log(weights[(size_t)m * K + k]) - 0.5 * D * ln2pi;
                    for (int d = 0; d < D; ++d) {
                        const double P = 1.0 / cv[d];
                        wv[d] = mu[d] * P * LOG2E;
                        wv[D + d] = -0.5 * P * LOG2E;
                        cst += 0.5 * std::log(P) - 0.5 * mu[d] * mu[d] * P;
                    }
                    cst *= LOG2E;
                }
                for (int j = 0; j < nk16 * 16; ++j) {
                    const int ks = j >> 4, hh = (j >> 3) & 1, e = j & 7;
                    const uint16_t hi = to_bf16(wv[j]);
                    const uint16_t lo = to_bf16(wv[j] - from_bf16(hi));
                    uint16_t* ph = reinterpret_cast<uint16_t*>(tile + ((((size_t)ks * 2 + 0) * 2 + hh) * 32 + mix) * 16) + e;
                    uint16_t* pl = reinterpret_cast<uint16_t*>(tile + ((((size_t)ks * 2 + 1) * 2 + hh) * 32 + mix) * 16) + e;
                    *ph = hi;
                    *pl = lo;
                }
                // accumulator order: lane half h, register i  <->  row (i & 3) + 8 (i >> 2) + 4 h
                const int hh = (mix >> 2) & 1, i = (mix & 3) + 4 * (mix >> 3);
                reinterpret_cast<float*>(tile + (size_t)2 * nk16 * 1024)[hh * 16 + i] = (float)cst;
            }
    }
    // precision = 1's error bound (gmm_band_kernel): per dimension the largest |mu P| and P / 2 over every mixture of every model
    std::vector<float> btab((size_t)2 * D, 0.f);
    for (size_t mk = 0; mk < (size_t)n_models * K; ++mk)
        for (int d = 0; d < D; ++d) {
            const double P = 1.0 / covars[mk * D + d];
            btab[d] = std::max(btab[d], (float)(std::fabs(means[mk * D + d]) * P * (1.0 + 1e-6)));
            btab[D + d] = std::max(btab[D + d], (float)(0.5 * P * (1.0 + 1e-6)));
        }
    ssp_gmm* g = new (std::nothrow) ssp_gmm;
    if (!g) SSP_FAIL(SSP_ERR_NOMEM, "gmm: host alloc");
    g->ctx = ctx;
    g->n_models = n_models;
    g->K = K;
    g->D = D;
    g->has_ubm = has_ubm ? 1 : 0;
    g->nq = nq;
    g->tiles_per_model = tpm;
    g->nk16 = nk16;
    int rc = g->wimg.alloc(img.size() * sizeof(float));
    if (rc == SSP_OK) rc = g->bound_tab.alloc(btab.size() * sizeof(float));
    if (rc == SSP_OK && hipMemcpyAsync(g->bound_tab.p, btab.data(), btab.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        set_error("gmm: bound table upload failed");
        rc = SSP_ERR_HIP;
    }
    if (rc == SSP_OK && nk16 > 0) {
        rc = g->wimg16.alloc(img16.size());
        if (rc == SSP_OK && hipMemcpyAsync(g->wimg16.p, img16.data(), img16.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            set_error("gmm: bf16 image upload failed");
            rc = SSP_ERR_HIP;
        }
    }
    if (rc == SSP_OK) {
        hipError_t e = hipMemcpyAsync(g->wimg.p, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("gmm: upload failed: %s", hipGetErrorString(e));
            rc = SSP_ERR_HIP;
        }
    }
    if (rc != SSP_OK) {
        delete g;
        return rc;
    }
    *out = g;
    return SSP_OK;
}

int ssp_gmm_destroy(ssp_gmm* gmm) {
    if (!gmm) return SSP_OK;
    ssp::quiesce_ctx(gmm->ctx);  // (the ctx may already be gone: common.hpp)
    delete gmm;
    return SSP_OK;
}

}  // extern "C"

namespace ssp {

static int launch_kernel_any(ssp_gmm* gmm, const GmmArgs& a, bool bf16, bool fused, hipStream_t s) {
    if (a.F <= 0) return SSP_OK;
    if (bf16) {
#define SSP_G16(NK_)                                                                          \
    case NK_:                                                                                  \
        return fused ? launch_loglik16<NK_, 2, true>(a, s) : launch_loglik16<NK_, 2, false>(a, s);
        switch (gmm->nk16) {
            SSP_G16(1) SSP_G16(2) SSP_G16(3) SSP_G16(4) SSP_G16(5) SSP_G16(6) SSP_G16(8)
            default: SSP_FAIL(SSP_ERR_UNSUPPORTED, "gmm: no bf16 kernel for nk=%d", gmm->nk16);
        }
#undef SSP_G16
    }
#define SSP_G32(NQ_, CT_)                                                                      \
    case NQ_:                                                                                  \
        return fused ? launch_loglik<NQ_, CT_, true>(a, s) : launch_loglik<NQ_, CT_, false>(a, s);
    switch (gmm->nq) {
        SSP_G32(4, 2) SSP_G32(7, 2) SSP_G32(10, 2) SSP_G32(16, 2) SSP_G32(24, 1) SSP_G32(32, 1)
        default: SSP_FAIL(SSP_ERR_UNSUPPORTED, "gmm: no kernel for nq=%d", gmm->nq);
    }
#undef SSP_G32
}

// frames per piece granule = frames of one wave of the scoring kernel
static int piece_granule(const ssp_gmm* gmm, bool bf16) { return (bf16 || gmm->nq <= 16) ? 64 : 32; }

// piece table of utterances [u0, u1) (absolute host offsets h_off): pb[i] = first piece of utterance u0 + i, pb[u1 - u0] = pieces
static void piece_table(const int64_t* h_off, int64_t u0, int64_t u1, int gran, std::vector<int32_t>& pb) {
    pb.resize((size_t)(u1 - u0) + 1);
    const int64_t base = h_off[u0];
    int32_t acc = 0;
    for (int64_t u = u0; u < u1; ++u) {
        pb[(size_t)(u - u0)] = acc;
        const int64_t a = h_off[u] - base, b = h_off[u + 1] - base;
        if (b > a) acc += (int32_t)((b - 1) / gran - a / gran + 1);
    }
    pb[(size_t)(u1 - u0)] = acc;
}

// Fused scoring of n_utt utterances: d_feats is indexed by ABSOLUTE frame (row h_off[u] is utterance u's first frame), h_off / d_off
// are the same n_utt + 1 offsets on the host / device.  bf16: the split-precision kernel.  d_margin (nullable): top-2 margins.
// `sub`: use the second buffer set (the re-scoring pass runs while the first set still holds the batch's tables).
static int score_fused(ssp_gmm* gmm, const float* d_feats, const int64_t* h_off, const int64_t* d_off, int64_t n_utt, uint64_t seg_serial,
                       bool bf16, bool sub, float* d_sc, int32_t* d_am, float* d_margin, hipStream_t s,
                       const int32_t* block_models = nullptr, int bl_stride = 0, bool* used_lists = nullptr) {
    const int M = gmm->n_models;
    const int gran = piece_granule(gmm, bf16);
    const size_t cap = [] {
        const char* e = getenv("SSP_GMM_SCRATCH_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)4 << 30;
    }();
    DevBuf& pb_dev = sub ? gmm->sub_pb : gmm->pb_dev;
    DevBuf& partial = sub ? gmm->sub_partial : gmm->partial;
    std::vector<int32_t>& pb_host = sub ? gmm->sub_pb_host : gmm->pb_host;
    // batches of whole utterances whose piece sums fit the cap (pieces <= frames / gran + utterances).  All batches' piece tables go to
    // the device in ONE upload before the first launch (pb_host stays untouched until the next call).
    const int64_t max_pieces = std::max<int64_t>((int64_t)(cap / ((size_t)M * sizeof(double))), 2);
    std::vector<int64_t> cuts{0};
    {
        int64_t u0 = 0;
        while (u0 < n_utt) {
            int64_t u1 = u0 + 1;
            auto bound = [&](int64_t ue) { return (h_off[ue] - h_off[u0]) / gran + (ue - u0); };
            if (bound(n_utt) <= max_pieces) {
                u1 = n_utt;
            } else {
                while (u1 < n_utt && bound(u1 + 1) <= max_pieces) ++u1;
            }
            cuts.push_back(u1);
            u0 = u1;
        }
    }
    const bool whole = cuts.size() == 2;
    const bool cached = !sub && whole && seg_serial != 0 && gmm->pb_serial == seg_serial && gmm->pb_gran == gran;
    int32_t max_batch_pieces = cached ? gmm->pb_pieces : 1;
    if (!cached) {
        pb_host.clear();
        std::vector<int32_t> one;
        for (size_t bi = 0; bi + 1 < cuts.size(); ++bi) {
            piece_table(h_off, cuts[bi], cuts[bi + 1], gran, one);
            max_batch_pieces = std::max(max_batch_pieces, one.back());
            pb_host.insert(pb_host.end(), one.begin(), one.end());
        }
        SSP_TRY(pb_dev.reserve(pb_host.size() * sizeof(int32_t)));
        SSP_HIP(hipMemcpyAsync(pb_dev.p, pb_host.data(), pb_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (!sub) {
            gmm->pb_serial = whole ? seg_serial : 0;
            gmm->pb_gran = gran;
            gmm->pb_pieces = max_batch_pieces;
        }
    }
    SSP_TRY(partial.reserve((size_t)max_batch_pieces * M * sizeof(double)));
    size_t pb_pos = 0;
    for (size_t bi = 0; bi + 1 < cuts.size(); ++bi) {
        const int64_t u0 = cuts[bi], u1 = cuts[bi + 1];
        const int32_t* pb_b = pb_dev.as<int32_t>() + pb_pos;
        pb_pos += (size_t)(u1 - u0) + 1;
        GmmArgs a{};
        a.feats = d_feats + (size_t)h_off[u0] * gmm->D;
        a.wimg = gmm->wimg.as<float>();
        a.wimg16 = gmm->wimg16.as<char>();
        a.F = h_off[u1] - h_off[u0];
        a.D = gmm->D;
        a.n_models = M;
        a.tiles_per_model = gmm->tiles_per_model;
        a.n_tiles = M * gmm->tiles_per_model;
        a.frame_off = d_off + u0;
        a.piece_base = pb_b;
        a.partial = partial.as<double>();
        a.frame_base = h_off[u0];
        a.n_utt = (int32_t)(u1 - u0);
        // (workgroup-indexed model lists belong to a batch that runs as ONE launch; a call that had to be cut scores every model)
        a.block_models = (whole && !bf16) ? block_models : nullptr;
        if (used_lists) *used_lists = a.block_models != nullptr;
        a.bl_stride = bl_stride;
        SSP_TRY(launch_kernel_any(gmm, a, bf16, true, s));
        hipLaunchKernelGGL(gmm_piece_reduce_kernel, dim3((unsigned)(u1 - u0)), dim3(256), (size_t)M * sizeof(float), s, partial.as<double>(),
                           d_off + u0, pb_b, M, gmm->has_ubm, d_sc ? d_sc + (size_t)u0 * M : nullptr,
                           d_am ? d_am + u0 : nullptr, d_margin ? d_margin + u0 : nullptr);
        SSP_HIP(hipGetLastError());
    }
    return SSP_OK;
}

// ---- precision = 4 (auto): the proven-band split path only when it is the faster one --------------------------------------------------
// The exact-arg-max guarantee of precision 1 scores every close call a second time in fp32; when most utterances are close calls that
// costs more than the fp32 path alone (bench `gmm_bf16x3_close_calls`: 164 ms against 121 with every utterance listed).  The pilot runs
// the split-precision pass, the band and the candidate selection on the first ~2 % of the utterances and prices the re-scoring from what
// it lists: work = sum over listed utterances of frames x candidate models (every model when the list overflows), relative to
// frames x models of the pilot.  Predicted cost of precision 1 in units of the fp32 path = SPLIT + RESCORE x work (SPLIT: the bf16x3
// sweep, ~0.32 of the fp32 sweep at any shape — same tiles, three MFMAs at 8 x the rate, the epilogue shared; RESCORE: a listed frame's
// workgroup scores the UNION of its utterances' candidate lists, measured ~2 x the sum on bench shapes, 1.05 x when every model is
// asked for).  fp32 is chosen when the prediction exceeds AUTO_CUTOVER.  Cost of asking: the pilot (2 % of a split pass) + one host wait.
static constexpr float AUTO_SPLIT = 0.33f, AUTO_RESCORE_LISTS = 2.0f, AUTO_RESCORE_ALL = 1.05f, AUTO_CUTOVER = 0.92f;
static constexpr int64_t AUTO_MIN_UTTS = 1024;   // below this a call is launch-latency bound either way: fp32 (fewest kernels, no host wait)

static int gmm_auto_choice(ssp_gmm* gmm, const float* d_feats, const ssp_segments* frame_seg, hipStream_t s, int* choice) {
    const int64_t n_utt = frame_seg->n;
    const int M = gmm->n_models;
    gmm->auto_pilot_utts = gmm->auto_pilot_listed = 0;
    gmm->auto_predicted = 0.f;
    gmm->auto_late = false;
    *choice = 0;
    if (gmm->nk16 == 0 || n_utt < AUTO_MIN_UTTS) return SSP_OK;       // no split kernels for this D / tiny batch
    if (gmm->has_ubm + 1 >= M) {                                         // one speaker model: nothing to confuse, the split path needs no re-scoring
        *choice = 1;
        return SSP_OK;
    }
    // A pilot costs at least one machine-filling round of the split kernel however few utterances it holds; on a batch of a few rounds
    // that is a large share of the pass (measured 17 % at 2 000 utterances, K = 512).  Such batches run the split pass on everything and
    // decide LATE, from the full close-call lists the precision-1 flow reads back anyway: when re-scoring them would cost more than a
    // whole fp32 pass, that pass runs instead (bounded at 1.33 x the fp32 path in the worst case, nothing extra in the usual one).
    if (frame_seg->total() < (int64_t)16 * gmm->ctx->num_cu * 3 * 256) {
        *choice = 1;
        gmm->auto_late = true;
        return SSP_OK;
    }
    int64_t n_p = std::max<int64_t>(256, n_utt / 50);
    if (const char* e = getenv("SSP_GMM_AUTO_PILOT")) n_p = std::max<int64_t>(1, atoll(e));
    n_p = std::min(n_p, n_utt);
    const int64_t* h_off = frame_seg->host.data();
    const int64_t* d_off = frame_seg->dev.as<int64_t>();
    SSP_TRY(gmm->margin.reserve((size_t)n_utt * sizeof(float)));
    SSP_TRY(gmm->sub_scores.reserve((size_t)n_p * M * sizeof(float)));
    SSP_TRY(gmm->band.reserve((size_t)n_utt * sizeof(float)));
    SSP_TRY(gmm->flag_list.reserve((size_t)n_utt * sizeof(int32_t)));
    SSP_TRY(gmm->flag_count.reserve(sizeof(int32_t)));
    float* sc = gmm->sub_scores.as<float>();
    // (second buffer set: the batch's cached piece table stays as it is)
    SSP_TRY(score_fused(gmm, d_feats, h_off, d_off, n_p, 0, /*bf16=*/true, /*sub=*/true, sc, nullptr, gmm->margin.as<float>(), s));
    const float eps = 3.01f * 0x1p-18f + 8.0f * (float)gmm->D * 0x1p-23f * 1.01f;
    hipLaunchKernelGGL(gmm_band_kernel, dim3((unsigned)n_p), dim3(256), 0, s, d_feats, d_off, gmm->D, gmm->bound_tab.as<float>(), eps, sc, M,
                       gmm->band.as<float>());
    hipLaunchKernelGGL(gmm_flag_kernel, dim3(1), dim3(256), 0, s, gmm->margin.as<float>(), gmm->band.as<float>(), (int)n_p,
                       gmm->flag_list.as<int32_t>(), gmm->flag_count.as<int32_t>());
    SSP_HIP(hipGetLastError());
    ssp_ctx* ctx = gmm->ctx;
    if (!ctx->pinned_words) SSP_HIP(hipHostMalloc((void**)&ctx->pinned_words, 64, hipHostMallocDefault));
    SSP_HIP(hipMemcpyAsync(ctx->pinned_words, gmm->flag_count.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipStreamSynchronize(s));
    const int32_t n_flag = ctx->pinned_words[0];
    double work = 0.0, work_all = 0.0;
    if (n_flag > 0) {
        SSP_TRY(gmm->cand.reserve((size_t)n_flag * GMM_CAND * sizeof(int32_t)));
        hipLaunchKernelGGL(gmm_candidates_kernel, dim3((unsigned)n_flag), dim3(256), 0, s, gmm->flag_list.as<int32_t>(), sc, gmm->band.as<float>(), M,
                           gmm->has_ubm, gmm->cand.as<int32_t>());
        SSP_HIP(hipGetLastError());
        gmm->sub_list_host.resize((size_t)n_flag);
        gmm->cand_host.resize((size_t)n_flag * GMM_CAND);
        SSP_HIP(hipMemcpyAsync(gmm->sub_list_host.data(), gmm->flag_list.p, (size_t)n_flag * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SSP_HIP(hipMemcpyAsync(gmm->cand_host.data(), gmm->cand.p, (size_t)n_flag * GMM_CAND * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SSP_HIP(hipStreamSynchronize(s));
        for (int32_t i = 0; i < n_flag; ++i) {
            const int32_t u = gmm->sub_list_host[(size_t)i];
            const double T = (double)(h_off[(size_t)u + 1] - h_off[(size_t)u]);
            const int32_t c0 = gmm->cand_host[(size_t)i * GMM_CAND];
            if (c0 < 0) work_all += T * M;
            else work += T * (double)c0;
        }
    }
    const double total = (double)std::max<int64_t>(h_off[(size_t)n_p] - h_off[0], 1) * M;
    const float pred = AUTO_SPLIT + (float)((AUTO_RESCORE_LISTS * work + AUTO_RESCORE_ALL * work_all) / total);
    gmm->auto_pilot_utts = (int32_t)n_p;
    gmm->auto_pilot_listed = n_flag;
    gmm->auto_predicted = pred;
    float cut = AUTO_CUTOVER;
    if (const char* e = getenv("SSP_GMM_AUTO_CUTOVER")) cut = (float)atof(e);
    *choice = pred < cut ? 1 : 0;
    return SSP_OK;
}

}  // namespace ssp

extern "C" {

int ssp_gmm_score(ssp_gmm* gmm, const float* feats, const ssp_segments* frame_seg, float* loglik_out,
                  float* scores_out, int32_t* argmax_out, int where, int precision, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_gmm_score");
    if (!gmm || !frame_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_score: null handle");
    ssp_ctx* ctx = gmm->ctx;
    SSP_TRY(use_ctx(ctx));
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_score: where");
    if (precision < 0 || precision > 4)
        SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_score: precision must be 0 (fp32 MFMA), 1 (bf16x3 MFMA + fp32 re-scoring inside the proven error bound), "
                                  "2 (bf16x3 MFMA alone), 3 (bf16x3 MFMA + fp32 re-scoring inside the calibrated, heuristic band) or 4 (auto: 1 or 0, "
                                  "whichever a pilot on the first utterances predicts to be faster)");
    if (precision != 0 && precision != 4 && gmm->nk16 == 0) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_gmm_score: bf16x3 path covers D <= 64");
    const bool want_auto = precision == 4;
    gmm->auto_late = false;
    if (want_auto) precision = 0;   // (until the pilot has spoken; score_samples requests — loglik_out — stay on the parity path)
    gmm->auto_choice = -1;
    if (kernel_ms) *kernel_ms = 0.f;
    const int64_t F = frame_seg->host.back();
    const int64_t n_utt = frame_seg->n;
    if (n_utt == 0) return SSP_OK;
    if (F > 0 && !feats) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_score: null feats");
    if (n_utt > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "gmm: too many utterances");
    hipStream_t s = ctx->stream;
    const int M = gmm->n_models;
    Staged sin, sll, ssc, sam;
    int rc;
    // Host-fed batches without re-scoring (precision 0 / 2) above two slices of features (GMM_UBM.py:181-197 hands host arrays): the
    // feature rows go through the ctx's ring in runs of whole utterances, copied in ahead of the kernels that score them (feed_rows,
    // staging.hpp) — the 81 ms a configs[2] batch spends on PCIe hide under its 120 ms of scoring.  Same kernels, same piece sums:
    // bits equal to the one-piece path.
    if (where == SSP_HOST && !loglik_out && (scores_out || argmax_out) && (precision == 0 || precision == 2) && !want_auto && n_utt >= 2 &&
        frame_seg->host.front() == 0 && (size_t)F * gmm->D * sizeof(float) >= 2 * host_slice_bytes()) {
        const std::vector<int64_t>& ho = frame_seg->host;
        const int64_t per = (int64_t)(host_slice_bytes() / ((size_t)gmm->D * sizeof(float)));
        std::vector<int64_t> ucut{0}, fcut{0};
        for (int64_t u = 0; u < n_utt;) {
            int64_t e = u + 1;
            while (e < n_utt && ho[(size_t)e + 1] - ho[(size_t)u] <= per) ++e;
            ucut.push_back(e);
            fcut.push_back(ho[(size_t)e]);
            u = e;
        }
        float* d_sc = (float*)ssc.out(ctx, scores_out, (size_t)n_utt * M * sizeof(float), where, &rc);
        SSP_TRY(rc);
        int32_t* d_am = (int32_t*)sam.out(ctx, argmax_out, (size_t)n_utt * sizeof(int32_t), where, &rc);
        SSP_TRY(rc);
        gmm->last_rescored = 0;
        Timer tms;
        SSP_TRY(tms.start(kernel_ms != nullptr, s));
        const int64_t* d_off = frame_seg->dev.as<int64_t>();
        SSP_TRY(feed_rows(ctx, feats, (size_t)gmm->D * sizeof(float), fcut, [&](int i, void* dev) -> int {
            const int64_t u0 = ucut[(size_t)i], u1 = ucut[(size_t)i + 1];
            if (ho[(size_t)u1] == ho[(size_t)u0] && !d_sc && !d_am) return SSP_OK;
            // (the slot holds frames [ho[u0], ho[u1]); score_fused addresses with the batch's absolute offsets: the pointer is biased)
            const float* biased = static_cast<const float*>(dev) - (size_t)ho[(size_t)u0] * gmm->D;
            return score_fused(gmm, biased, ho.data() + u0, d_off + u0, u1 - u0, 0, precision != 0, false, d_sc ? d_sc + (size_t)u0 * M : nullptr,
                               d_am ? d_am + u0 : nullptr, nullptr, s);
        }));
        SSP_TRY(tms.stop(s, kernel_ms));
        SSP_TRY(ssc.back(ctx, scores_out, (size_t)n_utt * M * sizeof(float), where));
        SSP_TRY(sam.back(ctx, argmax_out, (size_t)n_utt * sizeof(int32_t), where));
        SSP_HIP(hipStreamSynchronize(s));
        return SSP_OK;
    }
    const float* d_feats = (const float*)sin.in(ctx, feats, (size_t)F * gmm->D * sizeof(float), where, &rc);
    SSP_TRY(rc);
    Timer tm;
    bool timer_on = false;
    if (want_auto && !loglik_out && (scores_out || argmax_out)) {
        SSP_TRY(tm.start(kernel_ms != nullptr, s));   // (the pilot is part of what the call costs)
        timer_on = true;
        SSP_TRY(gmm_auto_choice(gmm, d_feats, frame_seg, s, &precision));
        gmm->auto_choice = precision;
    }
    const bool bf16 = precision != 0;
    const size_t ll_bytes = (size_t)M * (size_t)std::max<int64_t>(F, 1) * sizeof(float);
    float* d_sc = (float*)ssc.out(ctx, scores_out, (size_t)n_utt * M * sizeof(float), where, &rc);
    SSP_TRY(rc);
    int32_t* d_am = (int32_t*)sam.out(ctx, argmax_out, (size_t)n_utt * sizeof(int32_t), where, &rc);
    SSP_TRY(rc);
    if (loglik_out) {
        // the caller wants score_samples: the whole [M x F] matrix in one pass, per-utterance means from it
        float* d_ll = (float*)sll.out(ctx, loglik_out, ll_bytes, where, &rc);
        SSP_TRY(rc);
        SSP_TRY(tm.start(kernel_ms != nullptr, s));
        GmmArgs a{};
        a.feats = d_feats;
        a.wimg = gmm->wimg.as<float>();
        a.wimg16 = gmm->wimg16.as<char>();
        a.llT = d_ll;
        a.F = F;
        a.D = gmm->D;
        a.n_models = M;
        a.tiles_per_model = gmm->tiles_per_model;
        a.n_tiles = M * gmm->tiles_per_model;
        SSP_TRY(launch_kernel_any(gmm, a, bf16, false, s));
        if (scores_out || argmax_out) {
            hipLaunchKernelGGL(gmm_utt_reduce_kernel, dim3((unsigned)n_utt), dim3(256), (size_t)M * sizeof(float), s, d_ll, F,
                               frame_seg->dev.as<int64_t>(), (int64_t)0, M, gmm->has_ubm, d_sc, d_am);
            SSP_HIP(hipGetLastError());
        }
    } else if (scores_out || argmax_out) {
        // fused path: per-utterance sums leave the scoring kernel as piece partials, [M x F] never reaches HBM
        const bool rescore = (precision == 1 || precision == 3) && gmm->has_ubm + 1 < M;  // (one speaker model: nothing to confuse)
        if (!rescore) gmm->last_rescored = 0;
        float* d_margin = nullptr;
        float* d_sc_work = d_sc;
        if (rescore) {
            SSP_TRY(gmm->margin.reserve((size_t)n_utt * sizeof(float)));
            d_margin = gmm->margin.as<float>();
            if (!d_sc_work) {  // the close-call test looks at the score's magnitude: scores are needed even when only arg-max is asked for
                SSP_TRY(gmm->sub_scores.reserve((size_t)n_utt * M * sizeof(float)));
                d_sc_work = gmm->sub_scores.as<float>();
            }
        }
        if (!timer_on) SSP_TRY(tm.start(kernel_ms != nullptr, s));
        SSP_TRY(score_fused(gmm, d_feats, frame_seg->host.data(), frame_seg->dev.as<int64_t>(), n_utt, frame_seg->serial, bf16, false,
                            d_sc_work, d_am, d_margin, s));
        if (rescore) {
            // utterances whose top-2 margin lies within the split-precision error BOUND (gmm_band_kernel) are scored again on the exact
            // fp32 path, so the arg-max is the fp32 path's for every utterance
            const float eps = 3.01f * 0x1p-18f + 8.0f * (float)gmm->D * 0x1p-23f * 1.01f;
            SSP_TRY(gmm->band.reserve((size_t)n_utt * sizeof(float)));
            SSP_TRY(gmm->flag_list.reserve((size_t)n_utt * sizeof(int32_t)));
            SSP_TRY(gmm->flag_count.reserve(sizeof(int32_t)));
            // (precision 3: the calibrated band of rounds 2 - 3 instead — 8e-5 (|UBM score| + 1), eight times the error measured at K = 64 and
            //  K = 512, D = 39: a heuristic, 100 x narrower than the bound because rounding errors do not conspire and average over the frames)
            hipLaunchKernelGGL(gmm_band_kernel, dim3((unsigned)n_utt), dim3(256), 0, s, d_feats, frame_seg->dev.as<int64_t>(), gmm->D,
                               gmm->bound_tab.as<float>(), precision == 3 ? -1.0f : eps, d_sc_work, M, gmm->band.as<float>());
            hipLaunchKernelGGL(gmm_flag_kernel, dim3(1), dim3(256), 0, s, d_margin, gmm->band.as<float>(), (int)n_utt,
                               gmm->flag_list.as<int32_t>(), gmm->flag_count.as<int32_t>());
            SSP_HIP(hipGetLastError());
            int32_t n_flag = 0;
            SSP_HIP(hipMemcpyAsync(&n_flag, gmm->flag_count.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            SSP_HIP(hipStreamSynchronize(s));
            gmm->last_rescored = n_flag;
            if (n_flag > 0) {
                // only the models that can still win are scored again: per listed utterance the speaker models within the band of its best
                // split-precision difference (+ the UBM); the fp32 kernel's workgroups walk the union of their frames' lists
                SSP_TRY(gmm->cand.reserve((size_t)n_flag * GMM_CAND * sizeof(int32_t)));
                hipLaunchKernelGGL(gmm_candidates_kernel, dim3((unsigned)n_flag), dim3(256), 0, s, gmm->flag_list.as<int32_t>(), d_sc_work,
                                   gmm->band.as<float>(), M, gmm->has_ubm, gmm->cand.as<int32_t>());
                SSP_HIP(hipGetLastError());
                gmm->sub_list_host.resize((size_t)n_flag);
                gmm->cand_host.resize((size_t)n_flag * GMM_CAND);
                SSP_HIP(hipMemcpyAsync(gmm->sub_list_host.data(), gmm->flag_list.p, (size_t)n_flag * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                SSP_HIP(hipMemcpyAsync(gmm->cand_host.data(), gmm->cand.p, (size_t)n_flag * GMM_CAND * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                SSP_HIP(hipStreamSynchronize(s));
                if (gmm->auto_late) {   // precision 4, small batch: is re-scoring these lists dearer than the fp32 pass itself?
                    double work = 0.0, work_all = 0.0;
                    for (int32_t i = 0; i < n_flag; ++i) {
                        const int32_t u = gmm->sub_list_host[(size_t)i];
                        const double T = (double)(frame_seg->host[(size_t)u + 1] - frame_seg->host[(size_t)u]);
                        const int32_t c0 = gmm->cand_host[(size_t)i * GMM_CAND];
                        if (c0 < 0) work_all += T * M;
                        else work += T * (double)c0;
                    }
                    const double total = (double)std::max<int64_t>(F, 1) * M;
                    const float again = (float)((AUTO_RESCORE_LISTS * work + AUTO_RESCORE_ALL * work_all) / total);
                    gmm->auto_pilot_utts = (int32_t)n_utt;
                    gmm->auto_pilot_listed = n_flag;
                    gmm->auto_predicted = AUTO_SPLIT + again;
                    if (again >= 1.0f) {   // the whole batch on the fp32 path: every row and the arg-max are precision 0's
                        gmm->auto_choice = 0;
                        gmm->last_rescored = 0;
                        SSP_TRY(score_fused(gmm, d_feats, frame_seg->host.data(), frame_seg->dev.as<int64_t>(), n_utt, frame_seg->serial, false, false,
                                            d_sc, d_am, nullptr, s));
                        n_flag = 0;
                    }
                }
                if (n_flag > 0) {
                gmm->sub_off_host.resize((size_t)n_flag + 1);
                gmm->sub_off_host[0] = 0;
                for (int32_t i = 0; i < n_flag; ++i) {
                    const int32_t u = gmm->sub_list_host[(size_t)i];
                    gmm->sub_off_host[(size_t)i + 1] = gmm->sub_off_host[(size_t)i] + (frame_seg->host[(size_t)u + 1] - frame_seg->host[(size_t)u]);
                }
                const int64_t Fs = gmm->sub_off_host.back();
                // model list of every workgroup of the re-scoring launch (4 waves x the piece granule frames each): the sorted union of the
                // lists of the utterances its frames belong to; -1 (every model) when one of them asks for that or the union outgrows the row
                constexpr int BL = 48;
                const int64_t fw = 4 * (int64_t)piece_granule(gmm, false);
                const int64_t n_blk = (Fs + fw - 1) / fw;
                gmm->blk_host.assign((size_t)std::max<int64_t>(n_blk, 1) * (1 + BL), 0);
                {
                    int32_t i0 = 0;
                    std::vector<int32_t> un;
                    for (int64_t b = 0; b < n_blk; ++b) {
                        const int64_t fa = b * fw, fb = std::min(Fs, fa + fw);
                        while (i0 < n_flag && gmm->sub_off_host[(size_t)i0 + 1] <= fa) ++i0;
                        un.clear();
                        bool all = false;
                        for (int32_t i = i0; i < n_flag && gmm->sub_off_host[(size_t)i] < fb; ++i) {
                            if (gmm->sub_off_host[(size_t)i + 1] == gmm->sub_off_host[(size_t)i]) continue;  // (an empty utterance)
                            const int32_t* c = gmm->cand_host.data() + (size_t)i * GMM_CAND;
                            if (c[0] < 0) all = true;
                            else un.insert(un.end(), c + 1, c + 1 + c[0]);
                        }
                        std::sort(un.begin(), un.end());
                        un.erase(std::unique(un.begin(), un.end()), un.end());
                        int32_t* row = gmm->blk_host.data() + (size_t)b * (1 + BL);
                        if (all || (int)un.size() > BL) {
                            row[0] = -1;
                        } else {
                            row[0] = (int32_t)un.size();
                            std::copy(un.begin(), un.end(), row + 1);
                        }
                    }
                }
                SSP_TRY(gmm->blk.reserve(gmm->blk_host.size() * sizeof(int32_t)));
                SSP_HIP(hipMemcpyAsync(gmm->blk.p, gmm->blk_host.data(), gmm->blk_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
                SSP_TRY(gmm->sub_off.reserve(((size_t)n_flag + 1) * sizeof(int64_t)));
                SSP_HIP(hipMemcpyAsync(gmm->sub_off.p, gmm->sub_off_host.data(), ((size_t)n_flag + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
                SSP_TRY(gmm->sub_feats.reserve((size_t)std::max<int64_t>(Fs, 1) * gmm->D * sizeof(float)));
                hipLaunchKernelGGL(gmm_gather_frames_kernel, dim3((unsigned)n_flag), dim3(256), 0, s, d_feats, frame_seg->dev.as<int64_t>(),
                                   gmm->flag_list.as<int32_t>(), gmm->sub_off.as<int64_t>(), gmm->D, gmm->sub_feats.as<float>());
                SSP_HIP(hipGetLastError());
                // (when the caller asked for no scores, sub_scores already holds the work copy: the re-scored rows go to a second area)
                const size_t work_rows = d_sc ? 0 : (size_t)n_utt;
                SSP_TRY(gmm->sub_scores.reserve((work_rows + (size_t)n_flag) * M * sizeof(float)));
                if (!d_sc) d_sc_work = gmm->sub_scores.as<float>();
                float* sub_sc = gmm->sub_scores.as<float>() + work_rows * M;
                SSP_TRY(gmm->sub_argmax.reserve((size_t)n_flag * sizeof(int32_t)));
                bool used_lists = false;
                SSP_TRY(score_fused(gmm, gmm->sub_feats.as<float>(), gmm->sub_off_host.data(), gmm->sub_off.as<int64_t>(), n_flag, 0, false, true,
                                    sub_sc, gmm->sub_argmax.as<int32_t>(), nullptr, s, gmm->blk.as<int32_t>(), 1 + BL, &used_lists));
                hipLaunchKernelGGL(gmm_scatter_cand_kernel, dim3((unsigned)n_flag), dim3(64), 0, s, gmm->flag_list.as<int32_t>(),
                                   gmm->cand.as<int32_t>(), M, gmm->has_ubm, sub_sc, gmm->sub_argmax.as<int32_t>(), d_sc, d_am, used_lists ? 0 : 1);
                SSP_HIP(hipGetLastError());
                }  // (n_flag > 0 after the late decision)
            }
        }
    } else {
        SSP_TRY(tm.start(kernel_ms != nullptr, s));
    }
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(sll.back(ctx, loglik_out, ll_bytes, where));
    SSP_TRY(ssc.back(ctx, scores_out, (size_t)n_utt * M * sizeof(float), where));
    SSP_TRY(sam.back(ctx, argmax_out, (size_t)n_utt * sizeof(int32_t), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));
    return SSP_OK;
}

/* precision = 4: what the last such call's pilot saw and chose */
int ssp_gmm_last_auto(const ssp_gmm* gmm, int32_t* precision_used, int32_t* pilot_utts, int32_t* pilot_listed, float* predicted_cost) {
    if (!gmm) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_last_auto: null");
    if (precision_used) *precision_used = gmm->auto_choice;
    if (pilot_utts) *pilot_utts = gmm->auto_pilot_utts;
    if (pilot_listed) *pilot_listed = gmm->auto_pilot_listed;
    if (predicted_cost) *predicted_cost = gmm->auto_predicted;
    return SSP_OK;
}

/* diagnostics: utterances the last precision = 1 call re-scored on the fp32 path */
int ssp_gmm_last_rescored(const ssp_gmm* gmm, int32_t* n_out) {
    if (!gmm || !n_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_gmm_last_rescored: null");
    *n_out = gmm->last_rescored;
    return SSP_OK;
}

}  // extern "C"

// d-vector cosine scoring on gfx950 MFMA.
//
// Replaces the N x S Python loop  distance[i,j] = scipy.spatial.distance.cosine(X_val[i], avg[j])  followed by
// argmin(axis=1) (d_vector.py:315-319) and the linear scan of nn_model.eval (d_vector.py:346-361):
//   dist[i,j] = clip(1 - x_i.c_j / (|x_i| |c_j|), 0, 2)          (sp:spatial/distance.py:602-687)
// as one (centroids x d) . (d x embeddings) fp32 MFMA GEMM with the normalisation, clip and a running arg-min
// (numpy first-index tie rule) fused into the epilogue.  Centroids are the MFMA rows, embeddings the columns, so
// each lane owns ONE embedding and scans centroids in its accumulator registers.
#include <cmath>

#include "common.hpp"

namespace ssp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 128;  // centroid rows per block step
constexpr int BN = 128;  // embedding columns per workgroup
constexpr int BK = 32;   // k-chunk
constexpr int CPL = 128 * 4 + 4;  // floats per (q, h) plane of the operand image (+4 pad: conflict-free staging stores, csrc/dense.hip)

struct __attribute__((packed, aligned(4))) f4u {  // 16-byte load at dword alignment (rows of any length)
    float x, y, z, w;
};

struct CosArgs {
    const float* X;    // [N x d]
    const float* C;    // [S x d]
    const float* inc;  // [S] 1/|c_j|
    float* dist;       // nullable [N x S]
    int32_t* argmin;   // nullable [N]
    float* minval;     // nullable [N]
    int64_t N;
    int32_t d, S;
};

__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ A, int64_t rows, int d,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* __restrict__ p = A + row * d;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) s = fmaf(p[k], p[k], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = 1.0f / sqrtf(s);
}

// stage a [128 x BK] slab of a row-major matrix into the MFMA operand image [q=BK/8][h=2][row=128][e=4]:
// element (row, k = 8q + 2e + h).  Returns the sum of squares of what this thread loaded (for the row norm).
__device__ __forceinline__ float stage_slab(const float* __restrict__ A, int64_t row0, int64_t n_rows, int d, int k0,
                                            float* __restrict__ img, int tid) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;       // 1024 float4 per slab
        const int r = idx >> 3, c4 = idx & 7;  // 8 float4 per row
        const int64_t gr = row0 + r;
        const int k = k0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < n_rows) {
            const float* __restrict__ p = A + gr * d + k;
            if (k + 3 < d) {
                const f4u t4 = *reinterpret_cast<const f4u*>(p);
                v = make_float4(t4.x, t4.y, t4.z, t4.w);
            } else {
                if (k < d) v.x = p[0];
                if (k + 1 < d) v.y = p[1];
                if (k + 2 < d) v.z = p[2];
                if (k + 3 < d) v.w = p[3];
            }
        }
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        const int q = c4 >> 1, e0 = (c4 & 1) * 2;  // kk = (c4&1)*4 + {0,1,2,3}: h = kk&1, e = kk>>1
        float* b0 = img + (size_t)(q * 2 + 0) * CPL + r * 4 + e0;
        float* b1 = img + (size_t)(q * 2 + 1) * CPL + r * 4 + e0;
        *reinterpret_cast<float2*>(b0) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(b1) = make_float2(v.y, v.w);
    }
    return ss;
}

__global__ __launch_bounds__(256) void cosine_kernel(CosArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLAB = (BK / 8) * 2 * CPL;  // floats per operand slab
    float* imgA = reinterpret_cast<float*>(smem);  // centroids
    float* imgB = imgA + SLAB;                     // embeddings
    float* inx = imgB + SLAB;                      // [BN] 1/|x|
    float* redv = inx + BN;                        // [2][BN] cross-wave arg-min exchange
    int* redi = reinterpret_cast<int*>(redv + 2 * BN);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;  // 2 x 2 waves, each 64 rows x 64 cols
    const int64_t col0 = (int64_t)blockIdx.x * BN;
    const int d = a.d;
    const int n_kc = (d + BK - 1) / BK;
    const int n_rb = (a.S + BM - 1) / BM;

    float best[2];
    int besti[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        best[ct] = INFINITY;
        besti[ct] = 0x7fffffff;
    }
    float xss = 0.f;

    for (int rb = 0; rb < n_rb; ++rb) {
        f32x16 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;
        for (int kc = 0; kc < n_kc; ++kc) {
            __syncthreads();  // previous slab fully consumed
            stage_slab(a.C, (int64_t)rb * BM, a.S, d, kc * BK, imgA, tid);
            const float s = stage_slab(a.X, col0, a.N, d, kc * BK, imgB, tid);
            if (rb == 0) xss += s;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                f32x4 av[2], bv[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    av[rt] = *reinterpret_cast<const f32x4*>(imgA + (size_t)(q * 2 + h) * CPL + (wr * 64 + rt * 32 + fl) * 4);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    bv[ct] = *reinterpret_cast<const f32x4*>(imgB + (size_t)(q * 2 + h) * CPL + (wc * 64 + ct * 32 + fl) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rt][e], bv[ct][e], acc[rt][ct], 0, 0, 0);
            }
        }
        if (rb == 0) {  // embedding norms: 8 consecutive threads staged one row (tid>>3 + 32 i)
            // thread t staged rows r_i = (t + 256 i) >> 3 = (t >> 3) + 32 i for i = 0..3 -> four different rows:
            // keep it simple and exact: recompute per-row norms from global by one wave-parallel pass
            __syncthreads();
            for (int c = tid; c < BN; c += 256) {
                const int64_t gc = col0 + c;
                float s = 0.f;
                if (gc < a.N) {
                    const float* __restrict__ p = a.X + gc * d;
                    for (int k = 0; k < d; ++k) s = fmaf(p[k], p[k], s);
                }
                inx[c] = 1.0f / sqrtf(s);
            }
            __syncthreads();
        }
        // epilogue: distances for this 128-row block, running arg-min per column
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int c = wc * 64 + ct * 32 + fl;
            const float ix = inx[c];
            const int64_t gc = col0 + c;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = rb * BM + wr * 64 + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (row < a.S) {
                        float dv = 1.0f - acc[rt][ct][i] * a.inc[row] * ix;
                        // clip to [0, 2] like scipy, but a NaN (zero-norm embedding, empty speaker's centroid) stays a NaN: as the
                        // arg-min key it orders first, like numpy's argmin (d_vector.py:319)
                        const bool fin = dv == dv;
                        dv = fin ? fminf(fmaxf(dv, 0.0f), 2.0f) : dv;
                        if (a.dist && gc < a.N) a.dist[gc * a.S + row] = dv;
                        const float key = fin ? dv : -INFINITY;
                        if (key < best[ct] || (key == best[ct] && row < besti[ct])) {
                            best[ct] = key;
                            besti[ct] = row;
                        }
                    }
                }
        }
    }
    (void)xss;
    // combine the two lane halves, then the two row-waves, per column
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const float ob = __shfl_xor(best[ct], 32);
        const int oi = __shfl_xor(besti[ct], 32);
        if (ob < best[ct] || (ob == best[ct] && oi < besti[ct])) {
            best[ct] = ob;
            besti[ct] = oi;
        }
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int c = wc * 64 + ct * 32 + fl;
            redv[wr * BN + c] = best[ct];
            redi[wr * BN + c] = besti[ct];
        }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
        const int64_t gc = col0 + c;
        if (gc >= a.N) continue;
        float b0 = redv[c], b1 = redv[BN + c];
        int i0 = redi[c], i1 = redi[BN + c];
        if (b1 < b0 || (b1 == b0 && i1 < i0)) {
            b0 = b1;
            i0 = i1;
        }
        if (a.argmin) a.argmin[gc] = i0 == 0x7fffffff ? 0 : i0;
        if (a.minval) a.minval[gc] = b0 == -INFINITY ? __builtin_nanf("") : b0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Register-resident variant for d <= 256 (the d-vector sizes of the reference: 128 / 256): the embeddings of a wave
// (32 columns) stay in registers as the MFMA B operand for the whole sweep, the pre-normalised centroids stream through
// LDS as 32-row A tiles by LDS-DMA (double buffered), so the MFMA pipe sees one ds_read_b128 per 4 MFMAs and no barrier
// inside a tile.  Same epilogue as above.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// centroid image: [tile][q][h][row 32][e 4] = c_hat[tile*32 + row][8q + 2e + h]  (unit-norm rows, zero padded)
__global__ __launch_bounds__(256) void cos_pack_kernel(const float* __restrict__ C, int S, int d, int nq, float* __restrict__ img,
                                                       int normalize) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_rows = ((S + 31) / 32) * 32;
    if (row >= n_rows) return;
    float ss = 0.f;
    if (row < S)
        for (int k = lane; k < d; k += 64) ss = fmaf(C[(size_t)row * d + k], C[(size_t)row * d + k], ss);
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float inv = normalize ? 1.0f / sqrtf(ss) : 1.0f;
    float* tile = img + (size_t)(row >> 5) * nq * 256;
    for (int k = lane; k < nq * 8; k += 64) {
        const float v = (row < S && k < d) ? C[(size_t)row * d + k] * inv : 0.f;
        tile[((size_t)((k >> 3) * 2 + (k & 1)) * 32 + (row & 31)) * 4 + ((k & 7) >> 1)] = v;
    }
}

struct CosRegArgs {
    const float* X;
    const float* img;
    const float* bias;  // DENSE mode: nullable [S]
    int32_t relu;       // DENSE mode
    float* dist;        // [N x S]: cosine distances (nullable) or, in DENSE mode, the layer output
    int32_t* argmin;
    float* minval;
    int64_t N;
    int32_t d, S, n_tiles;
    // re-scoring pass of the split-precision path: the embeddings are X[rows[i]], i < *n_dev (a device-side count: no host round trip),
    // results go to argmin[rows[i]] / minval[rows[i]]; the launch covers the worst case and workgroups past the count leave at once
    const int32_t* rows = nullptr;
    const int32_t* n_dev = nullptr;
};

// DENSE: the same streaming GEMM as a fully connected layer (rows = units of an un-normalised packed weight image, epilogue = bias +
// ReLU + 16-byte stores of Y[sample][unit]) — ssp_dense_forward for d_in <= 256, i.e. the hidden layers of the d-vector network
template <int NQ, bool DENSE>
__global__ __launch_bounds__(256, 3) void cosine_reg_kernel(CosRegArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_FLOATS = NQ * 256;
    float* wbuf = reinterpret_cast<float*>(smem);  // [3][TILE_FLOATS / 2]; the first bytes double as the X staging area
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, h = lane >> 5;
    const int d = a.d;
    const int64_t col0 = (int64_t)blockIdx.x * 128 + wave * 32;  // this wave's 32 embeddings
    const int64_t Nn = a.n_dev ? (int64_t)*a.n_dev : a.N;
    if ((int64_t)blockIdx.x * 128 >= Nn) return;  // (whole workgroup: before any barrier)

    // ---- B operand: b[q][e] = x[col][8q + 2e + h]; staged through LDS in 64-wide k chunks so HBM reads stay coalesced
    float b[NQ][4];
    float ss = 0.f;
    {
        float* xs = wbuf + wave * (32 * 65);  // [32 rows][64 + 1 pad]
#pragma unroll
        for (int kc = 0; kc < NQ / 8; ++kc) {
            // 32 rows x 64 floats of this chunk = 512 float4: 8 per lane, all in flight together (a dword per iteration with a
            // wait after each, as the first version had it, cost a memory latency 128 times per workgroup)
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = lane + 64 * u, r = idx >> 4, k = kc * 64 + 4 * (idx & 15);
                const int64_t gc = col0 + r;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gc < Nn) {
                    const float* __restrict__ p = a.X + (a.rows ? (int64_t)a.rows[gc] : gc) * d + k;
                    if (k + 3 < d) {
                        const f4u t4 = *reinterpret_cast<const f4u*>(p);
                        v[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
                    } else {
                        if (k < d) v[u].x = p[0];
                        if (k + 1 < d) v[u].y = p[1];
                        if (k + 2 < d) v[u].z = p[2];
                        if (k + 3 < d) v[u].w = p[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = lane + 64 * u, r = idx >> 4, c = 4 * (idx & 15);
                float* dst = xs + r * 65 + c;
                dst[0] = v[u].x;
                dst[1] = v[u].y;
                dst[2] = v[u].z;
                dst[3] = v[u].w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = xs[fl * 65 + 8 * q8 + 2 * e + h];
                    b[kc * 8 + q8][e] = v;
                    ss = fmaf(v, v, ss);
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    ss += __shfl_xor(ss, 32);
    const float ix = 1.0f / sqrtf(ss);
    __syncthreads();  // staging area is about to be overwritten by tile 0

    // centroid tiles stream through a ring of THREE half-tile slots (k-halves of a 32-row tile, 16 KiB each at d = 256): 48 KiB per
    // workgroup, so three workgroups share a CU and three waves a SIMD's matrix pipe (two full-tile buffers = 64 KiB allowed two)
    constexpr int HALF = TILE_FLOATS / 2;
    const int n_steps = 2 * a.n_tiles;
    auto stage = [&](int step, int slot) {
        const float* src = a.img + (size_t)step * HALF;
        float* dst = wbuf + slot * HALF;
#pragma unroll
        for (int p = 0; p < (NQ / 2 + 3) / 4; ++p) {
            const int piece = wave + 4 * p;
            if (piece < NQ / 2)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + piece * 256 + lane * 4), (lds_ptr_t)(dst + piece * 256), 16, 0, 0);
        }
    };
    stage(0, 0);
    stage(1, 1);
    __syncthreads();

    float best = INFINITY;
    int besti = 0x7fffffff;
    const int64_t gc = col0 + fl;
    const int64_t go = (a.rows && gc < Nn) ? (int64_t)a.rows[gc] : gc;  // where this lane's results go
    int slot = 0;  // slot of half-step 2 t
    for (int t = 0; t < a.n_tiles; ++t) {
        const int s1 = slot == 2 ? 0 : slot + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // ---- first k half
        if (2 * t + 2 < n_steps) stage(2 * t + 2, s2);
        {
            const float* wcur = wbuf + slot * HALF;
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(wcur + ((q * 2 + h) * 32 + fl) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b[q][e], acc, 0, 0, 0);
            }
        }
        __syncthreads();
        // ---- second k half
        if (2 * t + 3 < n_steps) stage(2 * t + 3, slot);
        {
            const float* wcur = wbuf + s1 * HALF;
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(wcur + ((q * 2 + h) * 32 + fl) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b[NQ / 2 + q][e], acc, 0, 0, 0);
            }
        }
        if (DENSE) {
            if (gc < Nn) {
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int row = t * 32 + 8 * i4 + 4 * h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = acc[i4 * 4 + e];
                        if (row + e < a.S) {
                            if (a.bias) y += a.bias[row + e];
                            if (a.relu) y = fmaxf(y, 0.f);
                        }
                        v[e] = y;
                    }
                    float* yp = a.dist + gc * a.S + row;
                    if ((a.S & 3) == 0 && row + 3 < a.S && (reinterpret_cast<uintptr_t>(a.dist) & 15) == 0) {
                        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row + e < a.S) yp[e] = v[e];
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < a.S) {
                    float dv = 1.0f - acc[i] * ix;
                    const bool fin = dv == dv;  // a NaN stays a NaN and orders first (see cosine_kernel)
                    dv = fin ? fminf(fmaxf(dv, 0.0f), 2.0f) : dv;
                    if (a.dist && gc < Nn) a.dist[go * a.S + row] = dv;
                    const float key = fin ? dv : -INFINITY;
                    if (key < best || (key == best && row < besti)) {
                        best = key;
                        besti = row;
                    }
                }
            }
        }
        __syncthreads();
        slot = s2;
    }
    if (DENSE) return;
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(besti, 32);
    if (ob < best || (ob == best && oi < besti)) {
        best = ob;
        besti = oi;
    }
    if (h == 0 && gc < Nn) {
        if (a.argmin) a.argmin[go] = besti == 0x7fffffff ? 0 : besti;
        if (a.minval) a.minval[go] = best == -INFINITY ? __builtin_nanf("") : best;  // (distances are >= 0: -inf is the NaN key)
    }
}

template <int NQ, bool DENSE = false>
static int launch_cos_reg(const CosRegArgs& a, hipStream_t s) {
    const size_t ring = (size_t)3 * NQ * 128 * sizeof(float), xst = (size_t)4 * 32 * 65 * 4;
    const size_t lds = ring > xst ? ring : xst;
    const int64_t grid = ceil_div<int64_t>(a.N, 128);
    if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "cosine: too many embeddings for one launch");
    if (lds > 64 * 1024)
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_reg_kernel<NQ, DENSE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((cosine_reg_kernel<NQ, DENSE>), dim3((unsigned)grid), dim3(256), lds, s, a);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Split-precision arg-min (precision = 1): the same sweep on v_mfma_f32_32x32x16_bf16 — 16 x the per-clock rate of the fp32-input
// MFMA.  Unit-norm centroids and unit-norm embeddings are split into hi = bf16(v), lo = bf16(v - hi); a cosine is accumulated in fp32
// as  ch.xh + ch.xl + cl.xh  (3 MFMAs per 16 k instead of 8 fp32 ones).  What is dropped or rounded differently from the fp32 path is
// BOUNDED for unit vectors (cos_band below), so the kernel keeps every embedding's two largest cosines: when they are further apart
// than twice the bound the arg-min is the fp32 path's; otherwise (or when anything is NaN) the row goes on a device-side list and the
// fp32 kernel above scores it again (cosine_reg_kernel with rows / n_dev: no host round trip) — the arg-min is the fp32 path's on
// EVERY row, the minimum distance within the band of it.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// centroid image: [tile][half 2][ks 8 (NK/2)][part hi | lo][lane 64][8 bf16]: lane (h = l >> 5, row = l & 31) holds
// c_hat[tile * 32 + row][16 (half * NK/2 + ks) + 8 h + j]; flag[0] |= 1 when a centroid's norm is not a finite positive number
__global__ __launch_bounds__(256) void cos_pack16_kernel(const float* __restrict__ C, int S, int d, int nk, __bf16* __restrict__ img,
                                                         int32_t* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_rows = ((S + 31) / 32) * 32;
    if (row >= n_rows) return;
    float ss = 0.f;
    if (row < S)
        for (int k = lane; k < d; k += 64) ss = fmaf(C[(size_t)row * d + k], C[(size_t)row * d + k], ss);
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float inv = 1.0f / sqrtf(ss);
    if (row < S && lane == 0 && !(inv > 0.f && inv < INFINITY)) atomicOr(flag, 1);
    __bf16* tile = img + (size_t)(row >> 5) * nk * 1024;  // nk k-steps x (hi + lo) x 64 lanes x 8
    for (int k = lane; k < nk * 16; k += 64) {
        const float v = (row < S && k < d) ? C[(size_t)row * d + k] * inv : 0.f;
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        const int ks = k >> 4, hh = (k >> 3) & 1, j = k & 7;
        const size_t at = (((size_t)ks * 2) * 64 + hh * 32 + (row & 31)) * 8 + j;
        tile[at] = hi;
        tile[at + 512] = lo;
    }
}

struct Cos16Args {
    const float* X;
    const __bf16* img;
    int32_t* argmin;
    float* minval;
    uint32_t* mask;        // [ceil(N / 32)] one word per wave: bit r = row 32 w + r of this sweep's input goes to the NEXT stage.  Plain
                           // stores — until round 6 every wave appended its rows to the list behind one atomic counter, and the
                           // same-address atomics serialised (23 ns each: a sweep of 12 500 waves took 292 us instead of 40 once a few
                           // per cent of the rows were close calls); cos_compact_kernel turns the words into the list
    int32_t* list;         // (cos_compact_kernel's output: rows the next stage scores again)
    int32_t* count;        // (... [0] how many)
    const int32_t* cflag;  // [0] != 0: a centroid's norm is not a finite positive number (cos_pack16_kernel): every row goes on the list
    int64_t N;
    int32_t d, S, n_tiles;
    float band2;           // twice the bound on |cos(this sweep) - cos(fp32 path)|
    // later stages of a cascade: the embeddings are X[rows[i]], i < *n_dev (device-side count of the previous stage's list)
    const int32_t* rows;
    const int32_t* n_dev;
    // pilot of precision 3 (auto): also mark the rows whose two best cosines are closer than band2b (nullable; mask words like `mask`,
    // counted by cos_compact_kernel — an atomic per wave on one counter would serialise here as it did for the lists)
    uint32_t* mask2b;
    float band2b;
};

// SPLIT = 3: hi + lo operands, three products per k-step (error 1.3e-4 at d = 256); SPLIT = 1: the hi parts alone, one product per k-step
// (error 4e-3): the first sweep of the cascade (ssp_cosine_identify2 precision 2), whose close calls the SPLIT = 3 sweep takes over
template <int NK, int SPLIT>  // k-steps of 16: d <= 16 NK
__global__ __launch_bounds__(256, 3) void cosine_bf16x3_kernel(Cos16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PARTS = SPLIT == 1 ? 1 : 2;
    constexpr int HALF_BYTES = (NK / 2) * PARTS * 1024;  // a k-half of a 32-row tile in LDS: NK/2 k-steps x (hi [+ lo]) x 1 KiB
    constexpr int HALF_SRC = (NK / 2) * 2048;            // ... in the image (always hi + lo)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, h = lane >> 5;
    const int d = a.d;
    const int64_t col0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t Nn = a.n_dev ? (int64_t)*a.n_dev : a.N;
    if ((int64_t)blockIdx.x * 128 >= Nn) return;  // (whole workgroup: before any barrier)

    // ---- B operand: lane (embedding fl, half h) holds x_hat[col][16 ks + 8 h + j] as hi / lo fragments.  The rows come through LDS in
    //      64-wide k chunks (coalesced HBM reads), first as fp32 (the norm needs all of a row), then split in place.
    float bx[NK][8];
    float ss = 0.f;
    {
        float* xs = reinterpret_cast<float*>(smem) + wave * (32 * 65);  // [32 rows][64 + 1 pad]
#pragma unroll
        for (int kc = 0; kc < NK / 4; ++kc) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = lane + 64 * u, r = idx >> 4, k = kc * 64 + 4 * (idx & 15);
                const int64_t gc = col0 + r;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gc < Nn) {
                    const float* __restrict__ p = a.X + (a.rows ? (int64_t)a.rows[gc] : gc) * d + k;
                    if (k + 3 < d) {
                        const f4u t4 = *reinterpret_cast<const f4u*>(p);
                        v[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
                    } else {
                        if (k < d) v[u].x = p[0];
                        if (k + 1 < d) v[u].y = p[1];
                        if (k + 2 < d) v[u].z = p[2];
                        if (k + 3 < d) v[u].w = p[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = lane + 64 * u, r = idx >> 4, c = 4 * (idx & 15);
                float* dst = xs + r * 65 + c;
                dst[0] = v[u].x;
                dst[1] = v[u].y;
                dst[2] = v[u].z;
                dst[3] = v[u].w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v1 = xs[fl * 65 + 16 * k4 + 8 * h + j];
                    bx[kc * 4 + k4][j] = v1;
                    ss = fmaf(v1, v1, ss);
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    ss += __shfl_xor(ss, 32);
    const float ix = 1.0f / sqrtf(ss);
    bf16x8 bh[NK], bl[SPLIT == 1 ? 1 : NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = bx[ks][j] * ix;
            const __bf16 hi = (__bf16)v;
            bh[ks][j] = hi;
            if (SPLIT != 1) bl[SPLIT == 1 ? 0 : ks][j] = (__bf16)(v - (float)hi);
        }
    __syncthreads();  // the staging area is about to be overwritten by tile 0

    // centroid tiles stream through a ring of THREE half-tile slots, as in cosine_reg_kernel
    const int n_steps = 2 * a.n_tiles;
    auto stage = [&](int step, int slot) {
        const char* src = reinterpret_cast<const char*>(a.img) + (size_t)step * HALF_SRC;
        char* dst = smem + slot * HALF_BYTES;
        constexpr int NP = (NK / 2) * PARTS;  // 1-KiB pieces per half (SPLIT = 1: the hi pieces only — every other piece of the image)
#pragma unroll
        for (int p = 0; p < (NP + 3) / 4; ++p) {
            const int piece = wave + 4 * p;
            if (piece < NP)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + piece * (SPLIT == 1 ? 2048 : 1024) + lane * 16), (lds_ptr_t)(dst + piece * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    stage(1, 1);
    __syncthreads();

    // one k-half of a tile: the A fragments of k-step ks + 1 are in flight while the three products of k-step ks issue (the LDS latency
    // would otherwise sit between every triple: the reads are what the matrix pipe waits for)
    auto half_sweep = [&](const char* wcur, int k0, f32x16& acc) {
        const char* wl = wcur + lane * 16;
        if constexpr (SPLIT == 1) {
#pragma unroll
            for (int ks = 0; ks < NK / 2; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wl + ks * 1024);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[k0 + ks], acc, 0, 0, 0);
            }
        } else {
            bf16x8 ah = *reinterpret_cast<const bf16x8*>(wl), al = *reinterpret_cast<const bf16x8*>(wl + 1024);
#pragma unroll
            for (int ks = 0; ks < NK / 2; ++ks) {
                bf16x8 nh = ah, nl = al;
                if (ks + 1 < NK / 2) {
                    nh = *reinterpret_cast<const bf16x8*>(wl + (ks + 1) * 2048);
                    nl = *reinterpret_cast<const bf16x8*>(wl + (ks + 1) * 2048 + 1024);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[k0 + ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[SPLIT == 1 ? 0 : k0 + ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[k0 + ks], acc, 0, 0, 0);
                ah = nh;
                al = nl;
            }
        }
    };
    // the two largest cosines of this lane's rows and where the largest sits.  Per value FOUR vector instructions — the sweep is otherwise
    // bound by this epilogue, not by the matrix pipe (the first version: 250 per 16-MFMA tile): b2 = med3(b1, b2, v) is the second largest
    // of the three while b1 >= b2, b1 = max(b1, v), and the position is kept tile-relative (an inline constant) with the tile beside it.
    // NaNs need no flag of their own: a NaN embedding makes every cosine of the lane NaN, v > b1 never holds and b1 stays -inf (checked
    // below); a NaN centroid is the pack kernel's flag.
    float b1 = -INFINITY, b2 = -INFINITY;
    int i1 = 0, t1 = 0;  // register index inside the tile, tile
    int slot = 0;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.n_tiles; ++t) {
        const int s1 = slot == 2 ? 0 : slot + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        f32x16 acc = zero16;  // (a literal zero accumulator operand of the tile's first MFMA, not sixteen moves)
        if (2 * t + 2 < n_steps) stage(2 * t + 2, s2);
        half_sweep(smem + slot * HALF_BYTES, 0, acc);
        __syncthreads();
        if (2 * t + 3 < n_steps) stage(2 * t + 3, slot);
        half_sweep(smem + s1 * HALF_BYTES, NK / 2, acc);
        const float b1_in = b1;
        if (t * 32 + 32 <= a.S) {  // (wave-uniform: every row of the tile is a centroid)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[i];
                const bool gt = v > b1;  // ascending rows: a strict > keeps the first row of equal cosines
                b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
                i1 = gt ? i : i1;
                b1 = fmaxf(b1, v);
            }
        } else {  // the last, partly padded tile: rows past S must not take part (their zero vectors would score 0)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = row < a.S ? acc[i] : -INFINITY;
                const bool gt = v > b1;
                b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
                i1 = gt ? i : i1;
                b1 = fmaxf(b1, v);
            }
        }
        t1 = b1 > b1_in ? t : t1;
        __syncthreads();
        slot = s2;
    }
    i1 = t1 * 32 + (i1 & 3) + 8 * (i1 >> 2) + 4 * h;
    // the two lane halves hold different rows of the same embedding
    {
        const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
        const int oi = __shfl_xor(i1, 32);
        const bool take = o1 > b1 || (o1 == b1 && oi < i1);
        const float lose = take ? b1 : o1;
        b2 = fmaxf(fmaxf(b2, o2), lose);
        b1 = take ? o1 : b1;
        i1 = take ? oi : i1;
    }
    const bool bad = !(b1 > -INFINITY);  // nothing ever compared greater on either half: the embedding's cosines are NaN
    const int64_t gc = col0 + fl;
    const bool mine = h == 0 && gc < Nn;
    const int64_t go = (a.rows && gc < Nn) ? (int64_t)a.rows[gc] : gc;  // the row this lane's embedding is
    // close calls (and anything that is not a number: a zero-norm embedding, a NaN centroid) are scored again by the next stage; with a
    // single centroid there is nothing to confuse
    const bool must = bad || a.cflag[0] != 0 || !(ix > 0.f && ix < INFINITY);
    const bool again = mine && (must || (a.S > 1 && !(b1 - b2 >= a.band2)));
    const unsigned long long m = __builtin_amdgcn_ballot_w64(again);   // (`mine` holds on lanes 0..31 only: the word's 32 bits)
    if (lane == 0) a.mask[col0 >> 5] = (uint32_t)m;
    if (a.mask2b) {   // (wave-uniform: a kernel argument)
        const bool close2 = mine && (must || (a.S > 1 && !(b1 - b2 >= a.band2b)));
        const unsigned long long m2 = __builtin_amdgcn_ballot_w64(close2);
        if (lane == 0) a.mask2b[col0 >> 5] = (uint32_t)m2;
    }
    if (mine && !again) {
        if (a.argmin) a.argmin[go] = i1;
        if (a.minval) a.minval[go] = fminf(fmaxf(1.0f - b1, 0.0f), 2.0f);
    }
}

// cos_compact_kernel — the mask words of a sweep (Cos16Args::mask) become the next stage's row list.  One thread per word (32 rows), a
// workgroup's rows go behind ONE atomic on the counter: N / 8192 atomics per sweep instead of the N / 32 of the waves themselves.  The
// list is not in row order (workgroups land as they come); its consumers write results by row id, so the order has no effect.
// in_rows (nullable): the sweep was itself list driven — position p of its input was row in_rows[p]; n_dev (nullable): device-side
// length of that input.  row_base: the sweep covered rows [row_base, row_base + n_rows) of a larger array.
__global__ __launch_bounds__(256) void cos_compact_kernel(const uint32_t* __restrict__ mask, int64_t n_rows, const int32_t* n_dev,
                                                           const int32_t* __restrict__ in_rows, int32_t row_base, int32_t* __restrict__ list,
                                                           int32_t* count) {
    __shared__ int wsum[4];
    __shared__ int wg_base;
    const int64_t Nn = n_dev ? (int64_t)*n_dev : n_rows;
    if ((int64_t)blockIdx.x * 256 * 32 >= Nn) return;  // (whole workgroup, before any barrier)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t w = (int64_t)blockIdx.x * 256 + tid;
    const uint32_t mw = w * 32 < Nn ? mask[w] : 0u;
    const int c = __builtin_popcount(mw);
    int incl = c;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        incl += lane >= o ? t : 0;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        wg_base = tot > 0 ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    int at = wg_base + incl - c;
    for (int k = 0; k < wave; ++k) at += wsum[k];
    uint32_t m = mw;
    while (m) {
        const int b = __builtin_ctz(m);
        m &= m - 1;
        const int64_t pos = w * 32 + b;
        list[at++] = in_rows ? in_rows[pos] : (int32_t)pos + row_base;
    }
}

static int launch_cos_compact(const uint32_t* mask, int64_t n_rows, const int32_t* n_dev, const int32_t* in_rows, int32_t row_base, int32_t* list,
                              int32_t* count, hipStream_t s) {
    const int64_t grid = ceil_div<int64_t>(ceil_div<int64_t>(n_rows, 32), 256);
    if (grid <= 0) return SSP_OK;
    hipLaunchKernelGGL(cos_compact_kernel, dim3((unsigned)grid), dim3(256), 0, s, mask, n_rows, n_dev, in_rows, row_base, list, count);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

template <int NK, int SPLIT>
static int launch_cos16(const Cos16Args& a, hipStream_t s) {
    const size_t ring = (size_t)3 * (NK / 2) * (SPLIT == 1 ? 1024 : 2048), xst = (size_t)4 * 32 * 65 * 4;
    const size_t lds = ring > xst ? ring : xst;
    const int64_t grid = ceil_div<int64_t>(a.N, 128);
    if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "cosine: too many embeddings for one launch");
    if (lds > 64 * 1024)
        SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_bf16x3_kernel<NK, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((cosine_bf16x3_kernel<NK, SPLIT>), dim3((unsigned)grid), dim3(256), lds, s, a);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

template <int SPLIT>
static int launch_cos16_nk(int nk, const Cos16Args& a, hipStream_t s) {
    switch (nk) {
        case 4: return launch_cos16<4, SPLIT>(a, s);
        case 8: return launch_cos16<8, SPLIT>(a, s);
        case 12: return launch_cos16<12, SPLIT>(a, s);
        default: return launch_cos16<16, SPLIT>(a, s);
    }
}

// bound on |cos(bf16 x 3) - cos(fp32 path)| for unit vectors of dimension d.  x = hi + lo + r with |x - hi| <= 2^-9 |x| and
// |r| <= 2^-18 |x| (two round-to-nearest bf16 steps), likewise c: the products the split path leaves out (lo.lo, every r term) sum to at
// most 3.01 2^-18 sum |x_k| |c_k| <= 3.01 2^-18 (Cauchy-Schwarz, unit vectors).  Products of bf16 numbers are exact in fp32; the
// accumulation of 3 d terms whose absolute sum is <= 1 + 2^-7 rounds (or truncates: the matrix core's internal order is not documented, so
// the bound assumes the worst, 2^-23 per term) at most 3 d 2^-23 in total; the fp32 path's own sweep over d terms at most d 2^-23.
// The two paths also NORMALISE differently: the row norm is summed in another lane / k order (1 / |x| differs by up to d 2^-24 relative),
// and x / |x|, C / |c| and the final acc / |x| round once each on either path — (d + 8) 2^-24 covers all of it for cosines <= 1.
static float cos_norm_slack(int d) { return (float)(d + 8) * 0x1p-24f; }
static float cos_band(int d) { return 3.01f * 0x1p-18f + 4.0f * (float)d * 0x1p-23f * 1.01f + cos_norm_slack(d); }
// ... of the hi parts alone: x = hi + r, |r| <= 2^-9 |x|, |hi| <= (1 + 2^-9) |x|: what hi.hi' leaves out is at most
// (2 2^-9 (1 + 2^-9) + 2^-18) sum |x_k| |c_k| <= 2.01 2^-9; d exact products accumulated (2^-23 each, worst case) + the fp32 path's d
static float cos_band1(int d) { return 2.01f * 0x1p-9f + 2.0f * (float)d * 0x1p-23f * 1.01f + cos_norm_slack(d); }

// d_vector.py:310-313 — one workgroup per speaker, rows summed IN ROW ORDER into a float64 accumulator (fixed summation order, like
// numpy's mean on the reference's float64 `avg`).  The label array is scanned 2048 rows at a time with independent coalesced loads;
// the matching rows are compacted, in order, into an LDS list (wave ballots + prefix counts), then every thread adds its columns of
// the listed rows, four row loads in flight.  (The first version walked the labels one dependent load at a time: 82 ms at 1e6 rows.)
constexpr int CEN_K = 8;  // 256-row slices per scan batch

__global__ __launch_bounds__(256) void centroid_kernel(const float* __restrict__ X, const int32_t* __restrict__ labels, int64_t N, int d,
                                                       float* __restrict__ out) {
    __shared__ int32_t list[256 * CEN_K];  // row offsets (relative to the batch base) of this speaker, in row order
    __shared__ int32_t wcnt[CEN_K][4];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int DM = 4;                  // columns per thread and sweep: d <= 1024 in one sweep
    for (int k0 = 0; k0 < d; k0 += 256 * DM) {
        double acc[DM];
#pragma unroll
        for (int j = 0; j < DM; ++j) acc[j] = 0.0;
        int64_t cnt = 0;
        for (int64_t base = 0; base < N; base += 256 * CEN_K) {
            int32_t lab[CEN_K];
#pragma unroll
            for (int k = 0; k < CEN_K; ++k) {
                const int64_t r = base + 256 * k + tid;
                lab[k] = r < N ? labels[r] : -1;
            }
            unsigned long long bal[CEN_K];
#pragma unroll
            for (int k = 0; k < CEN_K; ++k) {
                bal[k] = __ballot(lab[k] == s);
                if (lane == 0) wcnt[k][wave] = __popcll(bal[k]);
            }
            __syncthreads();
            int pos = 0, total = 0;
#pragma unroll
            for (int k = 0; k < CEN_K; ++k) {
                int before = 0, all = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int c = wcnt[k][w];
                    before += w < wave ? c : 0;
                    all += c;
                }
                if (lab[k] == s) {
                    pos = total + before + __popcll(bal[k] & ((1ull << lane) - 1ull));
                    list[pos] = 256 * k + tid;
                }
                total += all;
            }
            __syncthreads();
            cnt += total;
            const float* __restrict__ xb = X + base * d;
            int j = 0;
            for (; j + 3 < total; j += 4) {
                const int r0 = list[j], r1 = list[j + 1], r2 = list[j + 2], r3 = list[j + 3];
#pragma unroll
                for (int c = 0; c < DM; ++c) {
                    const int col = k0 + 256 * c + tid;
                    if (col < d) {
                        const float v0 = xb[(size_t)r0 * d + col], v1 = xb[(size_t)r1 * d + col], v2 = xb[(size_t)r2 * d + col],
                                    v3 = xb[(size_t)r3 * d + col];
                        acc[c] += (double)v0;
                        acc[c] += (double)v1;
                        acc[c] += (double)v2;
                        acc[c] += (double)v3;
                    }
                }
            }
            for (; j < total; ++j) {
                const int r0 = list[j];
#pragma unroll
                for (int c = 0; c < DM; ++c) {
                    const int col = k0 + 256 * c + tid;
                    if (col < d) acc[c] += (double)xb[(size_t)r0 * d + col];
                }
            }
            __syncthreads();  // the list is rewritten by the next batch
        }
#pragma unroll
        for (int c = 0; c < DM; ++c) {
            const int col = k0 + 256 * c + tid;
            if (col < d) out[(size_t)s * d + col] = (float)(acc[c] / (double)cnt);
        }
    }
}

// fully connected layer through the register-resident streaming GEMM (d_in <= 256): called by ssp_dense_forward (dense.hip)
int launch_dense_reg(ssp_ctx* ctx, const float* dX, int64_t N, int d_in, const float* dW, const float* dB, int units, int relu, float* dY,
                     hipStream_t s) {
    const int nq = d_in <= 64 ? 8 : (d_in <= 128 ? 16 : (d_in <= 192 ? 24 : 32));
    const int n_tiles = (units + 31) / 32;
    DevBuf& img = ctx->scratch[0];
    SSP_TRY(img.reserve((size_t)n_tiles * nq * 256 * sizeof(float)));
    hipLaunchKernelGGL(cos_pack_kernel, dim3((unsigned)(n_tiles * 8)), dim3(256), 0, s, dW, units, d_in, nq, img.as<float>(), 0);
    CosRegArgs ra{dX, img.as<float>(), dB, relu, dY, nullptr, nullptr, N, d_in, units, n_tiles};
    switch (nq) {
        case 8: return launch_cos_reg<8, true>(ra, s);
        case 16: return launch_cos_reg<16, true>(ra, s);
        case 24: return launch_cos_reg<24, true>(ra, s);
        default: return launch_cos_reg<32, true>(ra, s);
    }
}

}  // namespace ssp

using namespace ssp;

extern "C" {

int ssp_centroids(ssp_ctx* ctx, const float* X, const int32_t* labels, int64_t N, int32_t d, int32_t S, float* out, int where,
                  float* kernel_ms) {
    ssp::TraceRange trace_("ssp_centroids");
    SSP_TRY(use_ctx(ctx));
    if (N < 0 || d < 1 || S < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_centroids: bad shape");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_centroids: where");
    if (!out || (N > 0 && (!X || !labels))) SSP_FAIL(SSP_ERR_INVALID, "ssp_centroids: null pointer");
    if (kernel_ms) *kernel_ms = 0.f;
    Staged sx, sl, so;
    int rc;
    const float* dX = (const float*)sx.in(ctx, X, (size_t)N * d * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const int32_t* dL = (const int32_t*)sl.in(ctx, labels, (size_t)N * sizeof(int32_t), where, &rc);
    SSP_TRY(rc);
    float* dO = (float*)so.out(ctx, out, (size_t)S * d * sizeof(float), where, &rc);
    SSP_TRY(rc);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, ctx->stream));
    hipLaunchKernelGGL(centroid_kernel, dim3((unsigned)S), dim3(256), 0, ctx->stream, dX, dL, N, d, dO);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(ctx->stream, kernel_ms));
    SSP_TRY(so.back(ctx, out, (size_t)S * d * sizeof(float), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(ctx->stream));
    return SSP_OK;
}

int ssp_cosine_identify(ssp_ctx* ctx, const float* X, int64_t N, int32_t d, const float* C, int32_t S, float* dist_out,
                        int32_t* argmin_out, float* min_out, int where, float* kernel_ms) {
    return ssp_cosine_identify2(ctx, X, N, d, C, S, dist_out, argmin_out, min_out, where, 0, kernel_ms);
}

// the close-call counts of the last split-precision call: read back when somebody asks (a device-pointer call does not wait for them)
static int cos_fetch_counts(const ssp_ctx* ctx) {
    if (!ctx->cos_counts_pending) return SSP_OK;
    int32_t hc[2] = {0, 0};
    SSP_HIP(hipSetDevice(ctx->device));
    SSP_HIP(hipMemcpyAsync(hc, ctx->cos_count.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SSP_HIP(hipStreamSynchronize(ctx->stream));
    ctx->cos_last_split = hc[0];
    ctx->cos_last_rescored = hc[1];
    ctx->cos_counts_pending = false;
    return SSP_OK;
}

int ssp_cosine_last_rescored(const ssp_ctx* ctx, int32_t* n_out) {
    if (!ctx || !n_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_last_rescored: null");
    SSP_TRY(cos_fetch_counts(ctx));
    *n_out = ctx->cos_last_rescored;
    return SSP_OK;
}

int ssp_cosine_last_split_rows(const ssp_ctx* ctx, int32_t* n_out) {
    if (!ctx || !n_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_last_split_rows: null");
    SSP_TRY(cos_fetch_counts(ctx));
    *n_out = ctx->cos_last_split;
    return SSP_OK;
}

static constexpr int64_t COS_AUTO_MIN_ROWS = 8192;   // precision 3: below this a call is launch bound either way

int ssp_cosine_last_auto(const ssp_ctx* ctx, int32_t* precision_used, int32_t* pilot_rows, int32_t* pilot_to_bf16x3, int32_t* pilot_to_fp32) {
    if (!ctx) SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_last_auto: null");
    if (precision_used) *precision_used = ctx->cos_auto_choice;
    if (pilot_rows) *pilot_rows = ctx->cos_auto_pilot_rows;
    if (pilot_to_bf16x3) *pilot_to_bf16x3 = ctx->cos_auto_to_x3;
    if (pilot_to_fp32) *pilot_to_fp32 = ctx->cos_auto_to_f32;
    return SSP_OK;
}

int ssp_cosine_identify2(ssp_ctx* ctx, const float* X, int64_t N, int32_t d, const float* C, int32_t S, float* dist_out,
                         int32_t* argmin_out, float* min_out, int where, int precision, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_cosine_identify");
    SSP_TRY(use_ctx(ctx));
    if (N < 0 || d < 1 || S < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_identify: bad shape N=%lld d=%d S=%d", (long long)N, d, S);
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_identify: where");
    if (precision < 0 || precision > 3)
        SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_identify: precision must be 0 (fp32 MFMA), 1 (bf16x3 MFMA + fp32 re-scoring of close calls), 2 (bf16 sweep, "
                                  "then bf16x3, then fp32 on the respective close calls) or 3 (auto: 2, 1 or 0, whichever a pilot on the first rows predicts to be fastest)");
    const bool want_auto = precision == 3;
    ctx->cos_auto_choice = -1;
    ctx->cos_auto_pilot_rows = ctx->cos_auto_to_x3 = ctx->cos_auto_to_f32 = 0;
    // the distance matrix / wide embeddings: fp32 only; small calls (an fp32 sweep under ~0.25 ms by the cost model below): launch bound, fp32
    if (want_auto && (dist_out || d > 256 || N < COS_AUTO_MIN_ROWS || (double)N * d * (7.8e-13 + 1.62e-14 * S) < 0.25e-3)) {
        precision = 0;
        ctx->cos_auto_choice = 0;
    }
    if (precision >= 1 && (dist_out || d > 256))
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_cosine_identify: precision 1 / 2 give the arg-min / minimum only (dist_out must be NULL) for d <= 256");
    if (kernel_ms) *kernel_ms = 0.f;
    if (N == 0) return SSP_OK;
    if (!X || !C) SSP_FAIL(SSP_ERR_INVALID, "ssp_cosine_identify: null input");
    if (N > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_cosine_identify: more than 2^31 - 1 embeddings in one call");
    hipStream_t s = ctx->stream;
    Staged sx, sc, sd, sa, sm;
    int rc;
    // Host-fed arg-min / minimum on the parity path above two slices of embeddings (d_vector.py:315-319 hands host arrays): the rows go
    // through the ctx's ring, copied in ahead of the sweep that scores them (feed_rows, staging.hpp): a call costs its PCIe time
    // instead of PCIe + sweep.  Rows are independent: bits equal to the one-piece path.
    if (where == SSP_HOST && precision == 0 && !dist_out && d <= 256 && N >= 2 && (size_t)N * d * sizeof(float) >= 2 * host_slice_bytes()) {
        const float* dC = (const float*)sc.in(ctx, C, (size_t)S * d * sizeof(float), where, &rc);
        SSP_TRY(rc);
        int32_t* dA = (int32_t*)sa.out(ctx, argmin_out, (size_t)N * sizeof(int32_t), where, &rc);
        SSP_TRY(rc);
        float* dM = (float*)sm.out(ctx, min_out, (size_t)N * sizeof(float), where, &rc);
        SSP_TRY(rc);
        const int nq = d <= 64 ? 8 : (d <= 128 ? 16 : (d <= 192 ? 24 : 32));
        const int n_tiles = (S + 31) / 32;
        DevBuf& img = ctx->cos_img;
        SSP_TRY(img.reserve((size_t)n_tiles * nq * 256 * sizeof(float)));
        ctx->cos_last_rescored = ctx->cos_last_split = 0;
        ctx->cos_counts_pending = false;
        const int64_t per = std::max<int64_t>(1, (int64_t)(host_slice_bytes() / ((size_t)d * sizeof(float))));
        std::vector<int64_t> cuts;
        for (int64_t r = 0; r < N; r += per) cuts.push_back(r);
        cuts.push_back(N);
        Timer tms;
        SSP_TRY(tms.start(kernel_ms != nullptr, s));
        hipLaunchKernelGGL(cos_pack_kernel, dim3((unsigned)(n_tiles * 8)), dim3(256), 0, s, dC, S, d, nq, img.as<float>(), 1);
        SSP_HIP(hipGetLastError());
        SSP_TRY(feed_rows(ctx, X, (size_t)d * sizeof(float), cuts, [&](int i, void* dev) -> int {
            const int64_t r0 = cuts[(size_t)i], r1 = cuts[(size_t)i + 1];
            CosRegArgs ra{static_cast<const float*>(dev), img.as<float>(), nullptr, 0, nullptr, dA ? dA + r0 : nullptr, dM ? dM + r0 : nullptr, r1 - r0, d, S, n_tiles};
            switch (nq) {
                case 8: return launch_cos_reg<8>(ra, s);
                case 16: return launch_cos_reg<16>(ra, s);
                case 24: return launch_cos_reg<24>(ra, s);
                default: return launch_cos_reg<32>(ra, s);
            }
        }));
        SSP_TRY(tms.stop(s, kernel_ms));
        SSP_TRY(sa.back(ctx, argmin_out, (size_t)N * sizeof(int32_t), where));
        SSP_TRY(sm.back(ctx, min_out, (size_t)N * sizeof(float), where));
        SSP_HIP(hipStreamSynchronize(s));
        return SSP_OK;
    }
    const float* dX = (const float*)sx.in(ctx, X, (size_t)N * d * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const float* dC = (const float*)sc.in(ctx, C, (size_t)S * d * sizeof(float), where, &rc);
    SSP_TRY(rc);
    float* dD = (float*)sd.out(ctx, dist_out, (size_t)N * S * sizeof(float), where, &rc);
    SSP_TRY(rc);
    int32_t* dA = (int32_t*)sa.out(ctx, argmin_out, (size_t)N * sizeof(int32_t), where, &rc);
    SSP_TRY(rc);
    float* dM = (float*)sm.out(ctx, min_out, (size_t)N * sizeof(float), where, &rc);
    SSP_TRY(rc);
    // (scratch lives on the ctx, grow-only: no allocation and no implicit synchronisation per call; calls on one ctx are stream-ordered)
    DevBuf &inc = ctx->cos_inc, &img = ctx->cos_img;
    Timer tm;
    bool timer_on = false;
    ctx->cos_last_rescored = ctx->cos_last_split = 0;
    ctx->cos_counts_pending = false;
    if (precision >= 1) {
        // split-precision sweep(s) keeping the two best cosines per embedding, then the fp32 kernel on the rows whose call is closer than
        // the error bound — device-side lists and counts: nothing comes back to the host in between.  precision 2 puts a hi-parts-only
        // sweep (one MFMA per k-step, bound 4e-3) in front: its close calls go to the bf16 x 3 sweep, that one's to fp32
        const int nk = d <= 64 ? 4 : (d <= 128 ? 8 : (d <= 192 ? 12 : 16)), nq = 2 * nk;
        const int n_tiles = (S + 31) / 32;
        DevBuf &img16 = ctx->cos_img16, &list1 = ctx->cos_list1, &list2 = ctx->cos_list2, &count = ctx->cos_count;
        SSP_TRY(img16.reserve((size_t)n_tiles * nk * 2048));
        SSP_TRY(img.reserve((size_t)n_tiles * nq * 256 * sizeof(float)));
        // a list of up to N rows + the mask words of the sweep that fills it (one per 32 rows, whole 128-row workgroups) behind it
        const size_t n_words = (size_t)ceil_div<int64_t>(N, 128) * 4, list_ints = (size_t)N + n_words + 64;
        SSP_TRY(list2.reserve(list_ints * sizeof(int32_t)));
        if (precision >= 2) SSP_TRY(list1.reserve(list_ints * sizeof(int32_t)));
        uint32_t* mask2 = reinterpret_cast<uint32_t*>(list2.as<int32_t>() + N);
        uint32_t* mask1 = precision >= 2 ? reinterpret_cast<uint32_t*>(list1.as<int32_t>() + N) : nullptr;
        SSP_TRY(count.reserve(4 * sizeof(int32_t)));  // [0] rows for the bf16 x 3 sweep (precision 2), [1] rows for fp32, [2] centroid flag
        int32_t* cnt = count.as<int32_t>();
        SSP_TRY(tm.start(kernel_ms != nullptr, s));
        timer_on = true;
        SSP_HIP(hipMemsetAsync(count.p, 0, 4 * sizeof(int32_t), s));
        hipLaunchKernelGGL(cos_pack16_kernel, dim3((unsigned)(n_tiles * 8)), dim3(256), 0, s, dC, S, d, nk, img16.as<__bf16>(), cnt + 2);
        hipLaunchKernelGGL(cos_pack_kernel, dim3((unsigned)(n_tiles * 8)), dim3(256), 0, s, dC, S, d, nq, img.as<float>(), 1);
        SSP_HIP(hipGetLastError());
        int64_t done1 = 0;   // rows of the bf16 sweep (stage 1 of the cascade) already swept by the pilot
        if (want_auto) {
            // ---- precision 3 (auto).  The pilot IS the first round of the cascade's first stage: the bf16 sweep over the first rows (one
            // machine-filling round of waves, so it costs its share of the full sweep and nothing more), listing its close calls as the
            // full sweep would and counting beside them the rows closer than the bf16 x 3 band (an estimate of what that sweep would hand
            // to fp32: a call that close at bf16 accuracy is, with few exceptions, that close at any).  One host wait, then the cost model
            // below prices the three ways on; the smallest wins (near-ties to the cascade: it continues with the remaining rows of its
            // first stage — nothing the pilot did is thrown away — while the other two start over).
            int64_t n_p = std::min<int64_t>((int64_t)ctx->num_cu * 12 * 32, std::max<int64_t>(2048, N / 8));
            if (const char* e = getenv("SSP_COS_AUTO_PILOT")) n_p = std::max<int64_t>(1, atoll(e));
            n_p = std::min(n_p, N) & ~(int64_t)127;   // (whole workgroups: the sweep behind the pilot starts on a mask-word boundary)
            Cos16Args p1{dX, img16.as<__bf16>(), dA, dM, mask1, list1.as<int32_t>(), cnt, cnt + 2, n_p, d, S, n_tiles, 2.0f * cos_band1(d), nullptr, nullptr};
            p1.mask2b = mask2;   // (the second sweep's words and list are free until that sweep runs)
            p1.band2b = 2.0f * cos_band(d);
            SSP_TRY(launch_cos16_nk<1>(nk, p1, s));
            SSP_TRY(launch_cos_compact(mask1, n_p, nullptr, nullptr, 0, list1.as<int32_t>(), cnt, s));       // (the counts are what the host reads)
            SSP_TRY(launch_cos_compact(mask2, n_p, nullptr, nullptr, 0, list2.as<int32_t>(), cnt + 3, s));
            if (!ctx->pinned_words) SSP_HIP(hipHostMalloc((void**)&ctx->pinned_words, 64, hipHostMallocDefault));   // (pinned: the 16-byte read-back is a plain DMA)
            int32_t* h = ctx->pinned_words;
            SSP_HIP(hipMemcpyAsync(h, count.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            SSP_HIP(hipStreamSynchronize(s));
            const float f1 = (float)h[0] / (float)n_p, f2 = (float)h[3] / (float)n_p;
            // cost model (seconds; fitted to tools/auto_sweep.py's grid after the lists moved to mask words + compaction): a sweep over n
            // rows costs n d (A + B S) — A: reading and normalising a row, B: the products per centroid (fp32 / bf16 x 3 / bf16 MFMA) —, a
            // LIST-driven sweep pays the row term twice (rows gathered through the list, 32 to a wave, partly filled)
            const double e = (double)N * d, A = 7.8e-13, B32 = 1.62e-14, BX3 = 4.55e-15, B16 = 1.74e-15;
            const double again32 = f2 * e * (2 * A + B32 * S);
            const double t32 = e * (A + B32 * S), tx3 = e * (A + BX3 * S) + again32, tcasc = e * (A + B16 * S) + f1 * e * (2 * A + BX3 * S) + again32;
            // (the cascade keeps the pilot's rows, the other two start over: it is taken unless the prediction says it loses clearly)
            precision = (tcasc <= 1.1 * std::min(tx3, t32)) ? 2 : (tx3 < 0.95 * t32 ? 1 : 0);
            // (a non-finite centroid makes the sweep list every row: f1 = f2 = 1 and the fp32 sweep is chosen)
            ctx->cos_auto_choice = precision;
            ctx->cos_auto_pilot_rows = (int32_t)n_p;
            ctx->cos_auto_to_x3 = h[0];
            ctx->cos_auto_to_f32 = h[3];
            if (precision == 2) done1 = n_p;                                   // the pilot's mask words stand: the sweep goes on behind them
            SSP_HIP(hipMemsetAsync(count.p, 0, 2 * sizeof(int32_t), s));        // (the lists are compacted afresh; the centroid flag at [2] stays)
        }
        if (precision >= 1) {
            Cos16Args c3{dX, img16.as<__bf16>(), dA, dM, mask2, list2.as<int32_t>(), cnt + 1, cnt + 2, N, d, S, n_tiles, 2.0f * cos_band(d), nullptr, nullptr};
            CosRegArgs ra{dX, img.as<float>(), nullptr, 0, nullptr, dA, dM, N, d, S, n_tiles};
            ra.rows = list2.as<int32_t>();
            ra.n_dev = cnt + 1;
            if (precision == 2) {
                Cos16Args c1 = c3;
                c1.mask = mask1;
                c1.list = list1.as<int32_t>();
                c1.count = cnt;
                c1.band2 = 2.0f * cos_band1(d);
                if (done1 > 0) {   // (precision 3: rows [0, done1) were the pilot; its mask words are words [0, done1 / 32))
                    c1.X = dX + (size_t)done1 * d;
                    c1.argmin = dA ? dA + done1 : nullptr;
                    c1.minval = dM ? dM + done1 : nullptr;
                    c1.N = N - done1;
                    c1.mask = mask1 + done1 / 32;
                }
                if (c1.N > 0) SSP_TRY(launch_cos16_nk<1>(nk, c1, s));
                SSP_TRY(launch_cos_compact(mask1, N, nullptr, nullptr, 0, list1.as<int32_t>(), cnt, s));
                c3.rows = list1.as<int32_t>();
                c3.n_dev = cnt;
            }
            SSP_TRY(launch_cos16_nk<3>(nk, c3, s));
            // (list driven: position p of the sweep's input was row list1[p], *cnt of them)
            SSP_TRY(launch_cos_compact(mask2, N, precision == 2 ? cnt : nullptr, precision == 2 ? list1.as<int32_t>() : nullptr, 0, list2.as<int32_t>(), cnt + 1, s));
            switch (nq) {
                case 8: SSP_TRY(launch_cos_reg<8>(ra, s)); break;
                case 16: SSP_TRY(launch_cos_reg<16>(ra, s)); break;
                case 24: SSP_TRY(launch_cos_reg<24>(ra, s)); break;
                default: SSP_TRY(launch_cos_reg<32>(ra, s)); break;
            }
            SSP_TRY(tm.stop(s, kernel_ms));
            SSP_TRY(sa.back(ctx, argmin_out, (size_t)N * sizeof(int32_t), where));
            SSP_TRY(sm.back(ctx, min_out, (size_t)N * sizeof(float), where));
            ctx->cos_counts_pending = true;  // (ssp_cosine_last_rescored / _split_rows fetch them)
            if (where == SSP_HOST) SSP_TRY(cos_fetch_counts(ctx));  // (host outputs: the call waits for its copies anyway)
            return SSP_OK;
        }
        // (the pilot chose the fp32 sweep: it follows, inside the same timed region)
    }
    if (d <= 256) {  // register-resident embeddings, LDS-DMA streamed centroid tiles
        const int nq = d <= 64 ? 8 : (d <= 128 ? 16 : (d <= 192 ? 24 : 32));
        const int n_tiles = (S + 31) / 32;
        SSP_TRY(img.reserve((size_t)n_tiles * nq * 256 * sizeof(float)));
        CosRegArgs ra{dX, img.as<float>(), nullptr, 0, dD, dA, dM, N, d, S, n_tiles};
        if (!timer_on) SSP_TRY(tm.start(kernel_ms != nullptr, s));
        hipLaunchKernelGGL(cos_pack_kernel, dim3((unsigned)(n_tiles * 8)), dim3(256), 0, s, dC, S, d, nq, img.as<float>(), 1);
        SSP_HIP(hipGetLastError());
        switch (nq) {
            case 8: SSP_TRY(launch_cos_reg<8>(ra, s)); break;
            case 16: SSP_TRY(launch_cos_reg<16>(ra, s)); break;
            case 24: SSP_TRY(launch_cos_reg<24>(ra, s)); break;
            default: SSP_TRY(launch_cos_reg<32>(ra, s)); break;
        }
        SSP_TRY(tm.stop(s, kernel_ms));
    } else {
        SSP_TRY(inc.reserve((size_t)S * sizeof(float)));
        const int64_t grid = ceil_div<int64_t>(N, BN);
        if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "cosine: too many embeddings for one launch");
        CosArgs a{dX, dC, inc.as<float>(), dD, dA, dM, N, d, S};
        constexpr size_t lds = (size_t)(2 * (BK / 8) * 2 * CPL + BN + 2 * BN) * sizeof(float) + 2 * BN * sizeof(int);
        SSP_TRY(tm.start(kernel_ms != nullptr, s));
        hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)ceil_div(S, 4)), dim3(256), 0, s, dC, (int64_t)S, d, inc.as<float>());
        SSP_HIP(hipGetLastError());
        hipLaunchKernelGGL(cosine_kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
        SSP_HIP(hipGetLastError());
        SSP_TRY(tm.stop(s, kernel_ms));
    }
    SSP_TRY(sd.back(ctx, dist_out, (size_t)N * S * sizeof(float), where));
    SSP_TRY(sa.back(ctx, argmin_out, (size_t)N * sizeof(int32_t), where));
    SSP_TRY(sm.back(ctx, min_out, (size_t)N * sizeof(float), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));  // (device pointers: asynchronous on the ctx stream, like every other entry point)
    return SSP_OK;
}

}  // extern "C"

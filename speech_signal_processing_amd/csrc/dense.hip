// Dense layer forward on gfx950 MFMA: Y = act(X W + b), the building block of the d-vector speaker network's predict()
// (d_vector.py:171-189: Dense(256) x 4 with ReLU between, spkModel.predict at d_vector.py:298-299 / 327 / 348).
// (units x d_in) . (d_in x samples) fp32 MFMA GEMM, output units as the MFMA rows and samples as the columns (the layout
// of the cosine scorer, csrc/cosine.hip, whose operand image this kernel shares), bias + ReLU fused into the epilogue, four
// consecutive units per 16-byte store.
#include <cstdlib>

#include "common.hpp"

namespace ssp {

int launch_dense_reg(ssp_ctx* ctx, const float* dX, int64_t N, int d_in, const float* dW, const float* dB, int units, int relu, float* dY,
                     hipStream_t s);  // cosine.hip

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DBM = 128;  // units per block step
constexpr int DBN = 128;  // samples per workgroup
constexpr int DBK = 32;   // k-chunk
constexpr int DPL = 128 * 4 + 4;  // floats per (q, h) plane of the operand image: 128 rows x 4 + 4 pad, so that the 8 planes the
                                  // 16 lanes of a ds_write_b64 group touch start on different banks (unpadded: 4-way conflicts)

struct DenseArgs {
    const float* X;     // [N x d_in]
    const float* Wt;    // [units x d_in]  (the Keras kernel transposed)
    const float* bias;  // [units] (nullable)
    float* Y;           // [N x units]
    int64_t N;
    int32_t d_in, units, relu;
};

struct __attribute__((packed, aligned(4))) f4u {
    float x, y, z, w;
};

// a [128 x DBK] slab of a row-major matrix goes global -> registers -> MFMA operand image [q=DBK/8][h=2] planes of [row=128][e=4]
// (element (row, k = 8q + 2e + h)) in two steps: the loads of slab kc + 1 are issued before the MFMAs of slab kc and land under them
__device__ __forceinline__ void dense_load(const float* __restrict__ A, int64_t row0, int64_t n_rows, int d, int k0, int tid,
                                           float4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 3, c4 = idx & 7;
        const int64_t gr = row0 + r;
        const int k = k0 + c4 * 4;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < n_rows) {
            const float* __restrict__ p = A + gr * d + k;
            if (k + 3 < d) {  // one 16-byte load at dword alignment (rows of d_in = 98 x 13 = 1274 floats are only 8-byte aligned)
                const f4u t = *reinterpret_cast<const f4u*>(p);
                v[i] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                if (k < d) v[i].x = p[0];
                if (k + 1 < d) v[i].y = p[1];
                if (k + 2 < d) v[i].z = p[2];
                if (k + 3 < d) v[i].w = p[3];
            }
        }
    }
}

__device__ __forceinline__ void dense_store(float* __restrict__ img, int tid, const float4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 3, c4 = idx & 7;
        const int q = c4 >> 1, e0 = (c4 & 1) * 2;
        *reinterpret_cast<float2*>(img + (size_t)(q * 2 + 0) * DPL + r * 4 + e0) = make_float2(v[i].x, v[i].z);
        *reinterpret_cast<float2*>(img + (size_t)(q * 2 + 1) * DPL + r * 4 + e0) = make_float2(v[i].y, v[i].w);
    }
}

__global__ __launch_bounds__(256, 2) void dense_kernel(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLAB = (DBK / 8) * 2 * DPL;
    float* imgA = reinterpret_cast<float*>(smem);  // weights (units)
    float* imgB = imgA + SLAB;                     // samples
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;  // 2 x 2 waves, each 64 units x 64 samples
    const int64_t col0 = (int64_t)blockIdx.x * DBN;
    const int d = a.d_in;
    const bool vst = (a.units & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0;
    const int n_kc = (d + DBK - 1) / DBK;
    const int n_rb = (a.units + DBM - 1) / DBM;
    for (int rb = 0; rb < n_rb; ++rb) {
        f32x16 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;
        // global -> register prefetch: the weights (L2 resident) one slab ahead, the samples (HBM, a TLB miss away) TWO slabs ahead —
        // one slab of MFMAs (~2 us) did not cover their latency (PMC: matrix pipe 48 % busy, waves waiting on memory)
        float4 va[4], vb0[4], vb1[4];
        dense_load(a.Wt, (int64_t)rb * DBM, a.units, d, 0, tid, va);
        dense_load(a.X, col0, a.N, d, 0, tid, vb0);
        dense_load(a.X, col0, a.N, d, DBK, tid, vb1);  // (beyond d_in: zeros)
        auto slab = [&](int kc, float4 (&vb)[4]) {
            __syncthreads();  // the previous slab is consumed
            dense_store(imgA, tid, va);
            dense_store(imgB, tid, vb);
            __syncthreads();
            if (kc + 1 < n_kc) dense_load(a.Wt, (int64_t)rb * DBM, a.units, d, (kc + 1) * DBK, tid, va);
            if (kc + 2 < n_kc) dense_load(a.X, col0, a.N, d, (kc + 2) * DBK, tid, vb);
#pragma unroll
            for (int q = 0; q < DBK / 8; ++q) {
                f32x4 av[2], bv[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    av[rt] = *reinterpret_cast<const f32x4*>(imgA + (size_t)(q * 2 + h) * DPL + (wr * 64 + rt * 32 + fl) * 4);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    bv[ct] = *reinterpret_cast<const f32x4*>(imgB + (size_t)(q * 2 + h) * DPL + (wc * 64 + ct * 32 + fl) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rt][e], bv[ct][e], acc[rt][ct], 0, 0, 0);
            }
        };
        for (int kc = 0; kc < n_kc; kc += 2) {
            slab(kc, vb0);
            if (kc + 1 < n_kc) slab(kc + 1, vb1);
        }
        // epilogue: accumulator register i of a lane = unit (i & 3) + 8 (i >> 2) + 4 h of the 32-unit tile, sample fl
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int64_t gc = col0 + wc * 64 + ct * 32 + fl;
            if (gc >= a.N) continue;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int row = rb * DBM + wr * 64 + rt * 32 + 8 * i4 + 4 * h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[rt][ct][i4 * 4 + e];
                        if (row + e < a.units) {
                            if (a.bias) t += a.bias[row + e];
                            if (a.relu) t = fmaxf(t, 0.f);
                        }
                        v[e] = t;
                    }
                    float* y = a.Y + gc * a.units + row;
                    if (vst && row + 3 < a.units) {
                        *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row + e < a.units) y[e] = v[e];
                    }
                }
        }
    }
}

}  // namespace ssp

using namespace ssp;

extern "C" int ssp_dense_forward(ssp_ctx* ctx, const float* X, int64_t N, int32_t d_in, const float* Wt, const float* bias,
                                 int32_t units, int32_t relu, float* Y, int where, float* kernel_ms) {
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (N < 0 || d_in < 1 || units < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_dense_forward: bad shape");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_dense_forward: where");
    if (N == 0) return SSP_OK;
    if (!X || !Wt || !Y) SSP_FAIL(SSP_ERR_INVALID, "ssp_dense_forward: null array");
    const int64_t grid = (N + DBN - 1) / DBN;
    if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_dense_forward: too many samples");
    hipStream_t s = ctx->stream;
    Staged sx, sw, sb, sy;
    int rc;
    const float* dX = (const float*)sx.in(ctx, X, (size_t)N * d_in * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const float* dW = (const float*)sw.in(ctx, Wt, (size_t)units * d_in * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const float* dB = (const float*)sb.in(ctx, bias, (size_t)units * sizeof(float), where, &rc);
    SSP_TRY(rc);
    float* dY = (float*)sy.out(Y, (size_t)N * units * sizeof(float), where, &rc);
    SSP_TRY(rc);
    if (d_in <= 256 && !getenv("SSP_DENSE_NO_REG")) {  // samples held in registers, weight tiles streamed (cosine.hip): the network's hidden layers
        Timer tr;
        SSP_TRY(tr.start(kernel_ms != nullptr, s));
        SSP_TRY(launch_dense_reg(ctx, dX, N, d_in, dW, dB, units, relu ? 1 : 0, dY, s));
        SSP_TRY(tr.stop(s, kernel_ms));
        SSP_TRY(sy.back(ctx, Y, (size_t)N * units * sizeof(float), where));
        if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));
        return SSP_OK;
    }
    DenseArgs a{dX, dW, dB, dY, N, d_in, units, relu ? 1 : 0};
    constexpr size_t lds = (size_t)2 * (DBK / 8) * 2 * DPL * sizeof(float);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    hipLaunchKernelGGL(dense_kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(sy.back(ctx, Y, (size_t)N * units * sizeof(float), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));
    return SSP_OK;
}

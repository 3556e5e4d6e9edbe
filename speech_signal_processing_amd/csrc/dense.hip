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
// (element (row, k = 8q + 2e + h)) in two steps: the loads of slab kc + 1 are issued before the MFMAs of slab kc and land under them.
// The loads are branch-free 16-byte buffer loads through a resource that covers exactly this workgroup's rows (rows past the end read
// as zero; rows of d_in = 98 x 13 = 1274 floats are only 8-byte aligned, which buffer loads allow); the columns past d_in inside the
// last slab are cleared in registers.  (Per-load bounds branches cut the k-loop into ~20 basic blocks and kept the compiler from
// placing any load among the MFMAs.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dense_rsrc(const float* A, int64_t row0, int64_t n_rows, int d) {
    const int64_t rows = n_rows - row0 < 128 ? n_rows - row0 : 128;
    const uint64_t addr = reinterpret_cast<uint64_t>(A + row0 * d);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane((int)(rows > 0 ? rows * d * 4 : 0));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}

__device__ __forceinline__ void dense_load(__amdgpu_buffer_rsrc_t rs, int d, int k0, int tid, float4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 3, c4 = idx & 7;
        const int k = k0 + c4 * 4;
        // (the whole vector is bit-cast before any element is taken: element reads straight off the builtin's result are narrowed
        //  to a one-dword load by this compiler)
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, k < d ? (r * d + k) * 4 : 0x7ffffff0, 0, 0));
        v[i].x = t.x;
        v[i].y = k + 1 < d ? t.y : 0.f;
        v[i].z = k + 2 < d ? t.z : 0.f;
        v[i].w = k + 3 < d ? t.w : 0.f;
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
    // two slabs of each operand image: slab kc + 1 is written (from registers loaded one or two iterations earlier) in the middle of
    // the MFMAs of slab kc, so a k-step costs ONE workgroup barrier and no wave waits on a store it has just issued (3.25 -> 2.95 ms
    // on the 1274-wide input layer at 5e5 samples against store, barrier, MFMAs, barrier on a single buffer).
    float* imgA = reinterpret_cast<float*>(smem);  // weights (units)  [2][SLAB]
    float* imgB = imgA + 2 * SLAB;                 // samples          [2][SLAB]
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
        // register staging: the weights (L2 resident) run one slab ahead of the LDS image, the samples (HBM, a TLB miss away) two
        float4 va[4], vp[4], vq[4];
        const __amdgpu_buffer_rsrc_t rw = dense_rsrc(a.Wt, (int64_t)rb * DBM, a.units, d), rx = dense_rsrc(a.X, col0, a.N, d);
        dense_load(rw, d, 0, tid, va);
        dense_load(rx, d, 0, tid, vq);
        dense_load(rx, d, DBK, tid, vp);  // (beyond d_in: zeros)
        __syncthreads();  // the previous unit block's last slab is consumed
        dense_store(imgA, tid, va);
        dense_store(imgB, tid, vq);
        dense_load(rw, d, DBK, tid, va);
        dense_load(rx, d, 2 * DBK, tid, vq);
        __syncthreads();
        // iteration kc: MFMAs on image kc & 1; `vn` holds sample slab kc + 1 and is refilled with slab kc + 3 once stored
        auto slab = [&](int kc, float4 (&vn)[4]) {
            const float* cA = imgA + (kc & 1) * SLAB;
            const float* cB = imgB + (kc & 1) * SLAB;
#pragma unroll
            for (int q = 0; q < DBK / 8; ++q) {
                if (q == DBK / 16 && kc + 1 < n_kc) {
                    dense_store(imgA + ((kc + 1) & 1) * SLAB, tid, va);
                    dense_store(imgB + ((kc + 1) & 1) * SLAB, tid, vn);
                    dense_load(rw, d, (kc + 2) * DBK, tid, va);  // (slabs past the last one: every offset out of range, zeros)
                    dense_load(rx, d, (kc + 3) * DBK, tid, vn);
                }
                f32x4 av[2], bv[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    av[rt] = *reinterpret_cast<const f32x4*>(cA + (size_t)(q * 2 + h) * DPL + (wr * 64 + rt * 32 + fl) * 4);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    bv[ct] = *reinterpret_cast<const f32x4*>(cB + (size_t)(q * 2 + h) * DPL + (wc * 64 + ct * 32 + fl) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rt][e], bv[ct][e], acc[rt][ct], 0, 0, 0);
            }
            __syncthreads();  // image kc is consumed, image kc + 1 is complete
        };
        for (int kc = 0; kc < n_kc; kc += 2) {
            slab(kc, vp);
            if (kc + 1 < n_kc) slab(kc + 1, vq);
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
    ssp::TraceRange trace_("ssp_dense_forward");
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (N < 0 || d_in < 1 || units < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_dense_forward: bad shape");
    if (d_in > (1 << 21)) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_dense_forward: d_in above 2^21 (32-bit offsets inside a 128-row block)");
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
    float* dY = (float*)sy.out(ctx, Y, (size_t)N * units * sizeof(float), where, &rc);
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
    constexpr size_t lds = (size_t)4 * (DBK / 8) * 2 * DPL * sizeof(float);  // two slabs of two operand images: 66 KB
    SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    hipLaunchKernelGGL(dense_kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(sy.back(ctx, Y, (size_t)N * units * sizeof(float), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));
    return SSP_OK;
}

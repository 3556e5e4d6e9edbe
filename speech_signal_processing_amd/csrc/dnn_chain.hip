// Forward pass of the reference's fully connected d-vector network as ONE call (d_vector.py:171-189: Dense(256)+ReLU x 3, Dense(256);
// spkModel.predict at d_vector.py:298-299, 327, 348): the input layer (98 x 13 = 1274 inputs) runs on the tiled MFMA GEMM of dense.hip,
// every following layer whose input and output widths are <= 256 runs inside dnn_chain_kernel, where the activations never leave
// the registers:
//   * wave = 16 samples, v_mfma_f32_16x16x4_f32 (exact fp32): units are the MFMA rows, samples the columns
//   * the layer input is the B operand, 64 registers per lane: lane (kq = l >> 4, n = l & 15) holds x[n][16 t + 4 kq + r] in
//     register 4 t + r
//   * an output tile of 16 units lands as 4 accumulator registers per lane: lane (kq, n), register r = unit 16 t' + 4 kq + r of
//     sample n — which IS the B-operand register 4 t' + r of the next layer.  Bias + ReLU are applied in place; no LDS round trip,
//     no HBM round trip between layers
//   * the weights stream through a 3-slot LDS ring by LDS-DMA, one 16-KiB tile (16 units x 256 inputs, packed in operand order by
//     ssp_dnn_create) per workgroup barrier, shared by the workgroup's 4 waves
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace ssp {

int launch_dense_reg(ssp_ctx* ctx, const float* dX, int64_t N, int d_in, const float* dW, const float* dB, int units, int relu, float* dY,
                     hipStream_t s);  // cosine.hip

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int CH_W = 256;              // widest layer the chain holds in registers
constexpr int CH_TILE = 16 * CH_W;     // floats per weight tile: 16 units x 256 inputs
constexpr int CH_MAXL = 8;

struct ChainArgs {
    const float* X;       // [N x d_in] input of the first chained layer
    float* Y;             // [N x d_out]
    const float* img;     // [n_layers][16 tiles][16 t][64 lanes][4 r] packed weights
    const float* bias;    // [n_layers][256] zero padded
    int64_t N;
    int32_t n_layers, d_in, d_out;
    int32_t relu[CH_MAXL];
};

struct __attribute__((packed, aligned(4))) f4u {
    float x, y, z, w;
};

__global__ __launch_bounds__(256, 3) void dnn_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);                    // [3][CH_TILE]
    float* s_bias = ring + 3 * CH_TILE;                              // [n_layers][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + n;    // this lane's sample
    for (int i = tid; i < a.n_layers * CH_W; i += 256) s_bias[i] = a.bias[i];

    auto stage = [&](int g, int slot) {  // tile g of the linear (layer, tile) stream -> ring slot
        const float* src = a.img + (size_t)g * CH_TILE;
        float* dst = ring + slot * CH_TILE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wave + 4 * p;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + piece * 256 + lane * 4), (lds_ptr_t)(dst + piece * 256), 16, 0, 0);
        }
    };
    const int n_tiles = a.n_layers * 16;
    stage(0, 0);
    if (n_tiles > 1) stage(1, 1);

    // layer input: 16-byte loads of x[row][16 t + 4 kq .. + 3] (rows of d_in floats are only dword aligned in general)
    float hb[64];
    {
        const float* __restrict__ xr = a.X + row * a.d_in;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int k = 16 * t + 4 * kq;
            f4u v = {0.f, 0.f, 0.f, 0.f};
            if (row < a.N) {
                if (k + 3 < a.d_in) {
                    v = *reinterpret_cast<const f4u*>(xr + k);
                } else {
                    if (k < a.d_in) v.x = xr[k];
                    if (k + 1 < a.d_in) v.y = xr[k + 1];
                    if (k + 2 < a.d_in) v.z = xr[k + 2];
                }
            }
            hb[4 * t + 0] = v.x;
            hb[4 * t + 1] = v.y;
            hb[4 * t + 2] = v.z;
            hb[4 * t + 3] = v.w;
        }
    }
    __syncthreads();  // bias table + tiles 0, 1 (the barrier drains the LDS-DMA)

    int g = 0, slot = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        const bool last = l + 1 == a.n_layers;
        const bool relu = a.relu[l] != 0;
        float nb[64];
#pragma unroll
        for (int tp = 0; tp < 16; ++tp) {
            const int s2 = slot >= 1 ? slot - 1 : 2;  // slot of tile g + 2 = the one tile g - 1 just left
            if (g + 2 < n_tiles) stage(g + 2, s2);
            const float* wcur = ring + slot * CH_TILE;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(wcur + (t * 64 + lane) * 4);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], hb[4 * t + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], hb[4 * t + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], hb[4 * t + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], hb[4 * t + 3], acc1, 0, 0, 0);
            }
            const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + l * CH_W + 16 * tp + 4 * kq);
            f32x4 v = (acc0 + acc1) + bv;
            if (relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (last) {
                const int u = 16 * tp + 4 * kq;
                if (row < a.N && u < a.d_out) {
                    float* y = a.Y + row * a.d_out + u;
                    if (u + 3 < a.d_out && (a.d_out & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0) {
                        *reinterpret_cast<f32x4*>(y) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (u + r < a.d_out) y[r] = v[r];
                    }
                }
            } else {
                nb[4 * tp + 0] = v[0];
                nb[4 * tp + 1] = v[1];
                nb[4 * tp + 2] = v[2];
                nb[4 * tp + 3] = v[3];
            }
            __syncthreads();  // tile g is consumed by every wave; tiles g + 1 (and g + 2) have landed
            ++g;
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (!last) {
#pragma unroll
            for (int i = 0; i < 64; ++i) hb[i] = nb[i];
        }
    }
}

// packs one layer: img[tp][t][lane][r] = W[unit 16 tp + (lane & 15)][input 16 t + 4 (lane >> 4) + r], zero beyond the layer's shape
__global__ __launch_bounds__(256) void dnn_pack_kernel(const float* __restrict__ Wt, int units, int d_in, float* __restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // over 16 x 16 x 64 x 4
    if (idx >= 16 * CH_TILE) return;
    const int r = idx & 3, lane = (idx >> 2) & 63, t = (idx >> 8) & 15, tp = idx >> 12;
    const int u = 16 * tp + (lane & 15), k = 16 * t + 4 * (lane >> 4) + r;
    img[idx] = (u < units && k < d_in) ? Wt[(size_t)u * d_in + k] : 0.f;
}

}  // namespace ssp

struct ssp_dnn {
    ssp_ctx* ctx = nullptr;
    int32_t n_layers = 0;
    std::vector<int32_t> dims, relu;
    std::vector<ssp::DevBuf> Wt, bias;   // raw layers (device), for the layers in front of the chain
    int32_t chain_from = 0;              // first layer that runs inside dnn_chain_kernel (n_layers: none)
    ssp::DevBuf img, cbias;              // packed chain weights / zero-padded biases
    ssp::DevBuf act[2];                  // activations between the un-chained layers (grow-only)
};

using namespace ssp;

extern "C" {

int ssp_dnn_create(ssp_ctx* ctx, int32_t n_layers, const int32_t* dims, const float* const* Wt, const float* const* bias,
                   const int32_t* relu, ssp_dnn** out) {
    if (!out) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_create: null out");
    *out = nullptr;
    SSP_TRY(use_ctx(ctx));
    if (n_layers < 1 || !dims || !Wt || !relu) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_create: bad arguments");
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_create: layer width < 1");
    for (int l = 0; l < n_layers; ++l)
        if (!Wt[l]) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_create: null kernel of layer %d", l);
    ssp_dnn* d = new (std::nothrow) ssp_dnn;
    if (!d) SSP_FAIL(SSP_ERR_NOMEM, "dnn: host alloc");
    d->ctx = ctx;
    d->n_layers = n_layers;
    d->dims.assign(dims, dims + n_layers + 1);
    d->relu.assign(relu, relu + n_layers);
    d->Wt = std::vector<DevBuf>((size_t)n_layers);
    d->bias = std::vector<DevBuf>((size_t)n_layers);
    hipStream_t s = ctx->stream;
    int rc = SSP_OK;
    for (int l = 0; l < n_layers && rc == SSP_OK; ++l) {
        const size_t wb = (size_t)dims[l + 1] * dims[l] * sizeof(float);
        rc = d->Wt[(size_t)l].alloc(wb);
        if (rc == SSP_OK && hipMemcpyAsync(d->Wt[(size_t)l].p, Wt[l], wb, hipMemcpyHostToDevice, s) != hipSuccess) rc = SSP_ERR_HIP;
        if (rc == SSP_OK && bias && bias[l]) {
            rc = d->bias[(size_t)l].alloc((size_t)dims[l + 1] * sizeof(float));
            if (rc == SSP_OK && hipMemcpyAsync(d->bias[(size_t)l].p, bias[l], (size_t)dims[l + 1] * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)
                rc = SSP_ERR_HIP;
        }
    }
    // the chain = the longest tail of layers whose input and output widths fit the register-resident form (at most CH_MAXL of them)
    int from = n_layers;
    while (from > 0 && dims[from - 1] <= CH_W && dims[from] <= CH_W && n_layers - (from - 1) <= CH_MAXL) --from;
    if (getenv("SSP_DNN_NO_CHAIN")) from = n_layers;
    d->chain_from = from;
    const int nc = n_layers - from;
    if (rc == SSP_OK && nc > 0) {
        rc = d->img.alloc((size_t)nc * 16 * CH_TILE * sizeof(float));
        if (rc == SSP_OK) rc = d->cbias.alloc((size_t)nc * CH_W * sizeof(float));
        if (rc == SSP_OK) {
            std::vector<float> cb((size_t)nc * CH_W, 0.f);
            for (int l = from; l < n_layers; ++l)
                if (bias && bias[l])
                    for (int u = 0; u < dims[l + 1]; ++u) cb[(size_t)(l - from) * CH_W + u] = bias[l][u];
            if (hipMemcpyAsync(d->cbias.p, cb.data(), cb.size() * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) rc = SSP_ERR_HIP;
            for (int l = from; l < n_layers && rc == SSP_OK; ++l) {
                hipLaunchKernelGGL(dnn_pack_kernel, dim3(16 * CH_TILE / 256), dim3(256), 0, s, d->Wt[(size_t)l].as<float>(), dims[l + 1], dims[l],
                                   d->img.as<float>() + (size_t)(l - from) * 16 * CH_TILE);
                if (hipGetLastError() != hipSuccess) rc = SSP_ERR_HIP;
            }
            if (rc == SSP_OK && hipStreamSynchronize(s) != hipSuccess) rc = SSP_ERR_HIP;  // cb (host) dies at return
        }
    } else if (rc == SSP_OK && hipStreamSynchronize(s) != hipSuccess) {
        rc = SSP_ERR_HIP;
    }
    if (rc != SSP_OK) {
        if (rc == SSP_ERR_HIP) set_error("ssp_dnn_create: upload / pack failed");
        delete d;
        return rc;
    }
    *out = d;
    return SSP_OK;
}

int ssp_dnn_destroy(ssp_dnn* dnn) {
    if (!dnn) return SSP_OK;
    ssp::quiesce_ctx(dnn->ctx);  // (the ctx may already be gone: common.hpp)
    delete dnn;
    return SSP_OK;
}

int ssp_dnn_forward(ssp_dnn* dnn, const float* X, int64_t N, float* Y, int where, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_dnn_forward");
    if (!dnn) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_forward: null handle");
    ssp_ctx* ctx = dnn->ctx;
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_forward: where");
    if (N < 0) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_forward: N < 0");
    if (N == 0) return SSP_OK;
    if (!X || !Y) SSP_FAIL(SSP_ERR_INVALID, "ssp_dnn_forward: null array");
    if ((N + 63) / 64 > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_dnn_forward: too many samples");
    hipStream_t s = ctx->stream;
    const int L = dnn->n_layers, from = dnn->chain_from;
    Staged sx, sy;
    int rc;
    const float* dX = (const float*)sx.in(ctx, X, (size_t)N * dnn->dims[0] * sizeof(float), where, &rc);
    SSP_TRY(rc);
    float* dY = (float*)sy.out(ctx, Y, (size_t)N * dnn->dims[(size_t)L] * sizeof(float), where, &rc);
    SSP_TRY(rc);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    // layers in front of the chain: one GEMM launch each (ssp_dense_forward's kernels), activations through a scratch pair
    const float* cur = dX;
    for (int l = 0; l < from; ++l) {
        float* dst = (l + 1 == L) ? dY : nullptr;
        if (!dst) {
            DevBuf& b = dnn->act[l & 1];
            SSP_TRY(b.reserve((size_t)N * dnn->dims[(size_t)l + 1] * sizeof(float)));
            dst = b.as<float>();
        }
        float ms_unused = 0.f;
        (void)ms_unused;
        SSP_TRY(ssp_dense_forward(ctx, cur, N, dnn->dims[(size_t)l], dnn->Wt[(size_t)l].as<float>(),
                                  dnn->bias[(size_t)l].p ? dnn->bias[(size_t)l].as<float>() : nullptr, dnn->dims[(size_t)l + 1], dnn->relu[(size_t)l], dst,
                                  SSP_DEVICE, nullptr));
        cur = dst;
    }
    if (from < L) {
        ChainArgs a{};
        a.X = cur;
        a.Y = dY;
        a.img = dnn->img.as<float>();
        a.bias = dnn->cbias.as<float>();
        a.N = N;
        a.n_layers = L - from;
        a.d_in = dnn->dims[(size_t)from];
        a.d_out = dnn->dims[(size_t)L];
        for (int l = from; l < L; ++l) a.relu[l - from] = dnn->relu[(size_t)l];
        const size_t lds = (size_t)(3 * CH_TILE + (L - from) * CH_W) * sizeof(float);
        if (lds > 64 * 1024)
            SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dnn_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dnn_chain_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), lds, s, a);
        SSP_HIP(hipGetLastError());
    }
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(sy.back(ctx, Y, (size_t)N * dnn->dims[(size_t)L] * sizeof(float), where));
    if (where == SSP_HOST) SSP_HIP(hipStreamSynchronize(s));
    return SSP_OK;
}

}  // extern "C"

// DTW distances of the reference's template matcher on gfx950 (MFCC_DTW.py:57-108, 187-217):
//   d, C, D1, path = dtw.accelerated_dtw(x, y, dist='euclidean')        (dtw package, warp = 1)
//   D1[i][j] = ||x_i - y_j||_2 + min(D1[i-1][j-1], D1[i-1][j], D1[i][j-1]),  borders +inf, D0[0][0] = 0;  d = D1[r-1][c-1]
// for every (test sequence, template) pair of distance_test / test().  The reference feeds FLATTENED MFCCs (_MFCC,
// MFCC_DTW.py:54: 1-D sequences of frames x 13 scalars, reshaped to (-1, 1) by the dtw package), so dim = 1 with sequences of
// ~1.2k elements is the common case; dim > 1 (2-D MFCC rows) is supported by the same kernel.
//
// One wave per pair, skewed wavefront: lane l owns W consecutive template columns and walks the rows one step behind
// lane l - 1 (row i = step - l), so the left and diagonal neighbours of its first column are the values lane l - 1
// produced one and two steps ago (one cross-lane shift per step); its own previous row stays in W registers.  Templates
// longer than 64 W columns are swept in super-blocks of 64 W columns with the boundary column parked in a per-wave scratch.
#include <cmath>

#include "common.hpp"

namespace ssp {

struct DtwArgs {
    const float* xq;          // [rows_q x dim]
    const float* xt;          // [rows_t x dim]
    const int64_t* q_off;     // [n_q + 1]
    const int64_t* t_off;     // [n_t + 1]
    void* out;                // [n_q x n_t]  (float; double in the path variant)
    void* bnd;                // [n_waves x max_r] boundary columns
    void* dmat;               // path variant: the accumulated cost matrix D1 [r x c] (double), else null
    int64_t n_pairs;
    int32_t n_q, n_t, dim, normalize, max_r;
};

// T = float: the all-pairs matcher.  T = double, STORE: the single-pair variant that also leaves D1 for the traceback
// (generate_template needs the warping path; float64 like the reference, so that ties break the same way).
template <int W, typename T, bool STORE>
__global__ __launch_bounds__(256) void dtw_kernel(DtwArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_id >= a.n_pairs) return;
    const int q = (int)(wave_id / a.n_t), p = (int)(wave_id - (int64_t)q * a.n_t);
    const int r = (int)(a.q_off[q + 1] - a.q_off[q]), c = (int)(a.t_off[p + 1] - a.t_off[p]);
    const int dim = a.dim;
    const float* __restrict__ x = a.xq + a.q_off[q] * dim;
    const float* __restrict__ y = a.xt + a.t_off[p] * dim;
    T* __restrict__ bnd = static_cast<T*>(a.bnd) + wave_id * a.max_r;
    T* __restrict__ dmat = static_cast<T*>(a.dmat);
    const T INF = (T)INFINITY;
    T result = INF;  // empty sequences: the package would fail; report +inf
    if (r > 0 && c > 0) {
        for (int cb0 = 0; cb0 < c; cb0 += 64 * W) {
            const int cb = min(64 * W, c - cb0);        // columns of this super-block
            const int lanes = (cb + W - 1) / W;         // lanes that own at least one column
            const int j0 = cb0 + lane * W;              // this lane's first column
            const bool more = cb0 + 64 * W < c;         // another super-block follows: park the last column
            T yreg[W];
            if (dim == 1) {
#pragma unroll
                for (int k = 0; k < W; ++k) yreg[k] = j0 + k < c ? (T)y[j0 + k] : (T)0;
            }
            T prev[W];
#pragma unroll
            for (int k = 0; k < W; ++k) prev[k] = INF;  // row -1
            T last = INF;    // this lane's value in its LAST column at its previous step (row i - 1)
            T diagl = INF;   // D[i-1][j0-1]
            for (int s = 0; s < r + lanes - 1; ++s) {
                const int i = s - lane;
                // lane l - 1's last-column value for row i (it finished that row in the previous step)
                T left = __shfl_up(last, 1);
                const bool act = i >= 0 && i < r && lane < lanes;
                if (lane == 0) {
                    if (cb0 == 0) {
                        left = INF;
                        diagl = i == 0 ? (T)0 : INF;
                    } else if (act) {
                        // agent-scope loads: the values were stored by another lane of this wave in the previous super-block
                        left = __hip_atomic_load(&bnd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        diagl = i > 0 ? __hip_atomic_load(&bnd[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INF;
                    }
                }
                if (act) {
                    T cur[W];
                    T xi = 0;
                    if (dim == 1) xi = (T)x[i];
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int j = j0 + k;
                        T cost;
                        if (dim == 1) {
                            cost = xi > yreg[k] ? xi - yreg[k] : yreg[k] - xi;
                        } else {
                            T ss = 0;
                            if (j < c)
                                for (int e = 0; e < dim; ++e) {
                                    const T df = (T)x[(size_t)i * dim + e] - (T)y[(size_t)j * dim + e];
                                    ss += df * df;
                                }
                            cost = sqrt(ss);
                        }
                        const T up = prev[k];
                        const T dg = k == 0 ? diagl : prev[k - 1];
                        const T lf = k == 0 ? left : cur[k - 1];
                        const T mn = dg < up ? (dg < lf ? dg : lf) : (up < lf ? up : lf);
                        cur[k] = j < c ? cost + mn : INF;
                        if (STORE && j < c) dmat[(size_t)i * c + j] = cur[k];
                    }
                    diagl = left;  // D[i][j0-1] is the diagonal neighbour of row i + 1
                    const int kl = min(W, cb - lane * W) - 1;  // this lane's last valid column
                    T lv = cur[0];
#pragma unroll
                    for (int k = 1; k < W; ++k) lv = k == kl ? cur[k] : lv;
#pragma unroll
                    for (int k = 0; k < W; ++k) prev[k] = cur[k];
                    last = lv;
                    if (lane == lanes - 1) {
                        if (more) __hip_atomic_store(&bnd[i], lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (i == r - 1 && !more) result = lv;
                    }
                }
            }
            if (more) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the parked column is read back by lane 0 of this wave
        }
    }
    // the value sits in the lane that owned the last column
    T v = result;
    for (int o = 32; o > 0; o >>= 1) {
        const T w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    if (lane == 0) static_cast<T*>(a.out)[wave_id] = (a.normalize && r + c > 0) ? v / (T)(r + c) : v;
}


// dim == 1 all-pairs matcher, tuned: the lane's previous row is updated in place (three scalars carry diagonal / left / new
// value along the row), x[i + 1] is fetched one step ahead (the plain kernel stalls a full memory latency per step on x[i]);
// three instructions per cell (subtract, min3, add with |.| on its operand).
template <int W>
__global__ __launch_bounds__(256) void dtw1_kernel(DtwArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_id >= a.n_pairs) return;
    const int q = (int)(wave_id / a.n_t), p = (int)(wave_id - (int64_t)q * a.n_t);
    const int r = (int)(a.q_off[q + 1] - a.q_off[q]), c = (int)(a.t_off[p + 1] - a.t_off[p]);
    const float* __restrict__ x = a.xq + a.q_off[q];
    const float* __restrict__ y = a.xt + a.t_off[p];
    float* __restrict__ bnd = static_cast<float*>(a.bnd) + wave_id * a.max_r;
    float result = INFINITY;
    if (r > 0 && c > 0) {
        for (int cb0 = 0; cb0 < c; cb0 += 64 * W) {
            const int cb = min(64 * W, c - cb0);
            const int lanes = (cb + W - 1) / W;
            const int j0 = cb0 + lane * W;
            const int nvalid = max(0, min(W, c - j0));  // this lane's columns inside the template
            const bool more = cb0 + 64 * W < c;
            float yreg[W], prev[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                yreg[k] = k < nvalid ? y[j0 + k] : 0.f;
                prev[k] = INFINITY;  // row -1
            }
            float last = INFINITY, diagl = INFINITY;
            float xnext = lane < lanes && r > 0 ? x[0] : 0.f;  // this lane starts with row 0 at step `lane`
            for (int s = 0; s < r + lanes - 1; ++s) {
                const int i = s - lane;
                float left = __shfl_up(last, 1);
                const bool act = i >= 0 && i < r && lane < lanes;
                if (lane == 0) {
                    if (cb0 == 0) {
                        left = INFINITY;
                        diagl = i == 0 ? 0.f : INFINITY;
                    } else if (act) {
                        left = __hip_atomic_load(&bnd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        diagl = i > 0 ? __hip_atomic_load(&bnd[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFINITY;
                    }
                }
                if (act) {
                    const float xi = xnext;
                    if (i + 1 < r) xnext = x[i + 1];  // in flight during this row's W cells
                    float d = diagl, l = left;
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const float up = prev[k];
                        const float v = fabsf(xi - yreg[k]) + fminf(d, fminf(up, l));
                        d = up;
                        l = v;  // (columns past the template's end compute on: a cell depends on columns <= its own, so what they hold never
                        prev[k] = v;  //  reaches a valid one; a per-cell select to freeze them was a quarter of the loop's instructions)
                    }
                    diagl = left;
                    last = l;
                    // (a block that hands over to a next one is full: its last lane's last column is valid)
                    if (more && lane == lanes - 1) __hip_atomic_store(&bnd[i], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!more && lane == lanes - 1) {  // D[r - 1][c - 1]: the last valid column of the last lane, after its last row
                float lv = prev[0];
#pragma unroll
                for (int k = 1; k < W; ++k) lv = k == nvalid - 1 ? prev[k] : lv;
                result = lv;
            }
            if (more) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        }
    }
    float v = result;
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    if (lane == 0) static_cast<float*>(a.out)[wave_id] = (a.normalize && r + c > 0) ? v / (float)(r + c) : v;
}

// the package's _traceback over D0 (= D1 with a +inf border and D0[0][0] = 0): from (r-1, c-1) back to (0, 0), at each step the
// first minimum of (diagonal, up, left).  One thread; writes the path backwards into the END of path_i / path_j (capacity r + c)
__global__ void dtw_traceback_kernel(const double* __restrict__ D1, int r, int c, int32_t* path_i, int32_t* path_j, int32_t* path_len) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto D0 = [&](int i, int j) -> double {  // D0[i][j], i in 0..r, j in 0..c
        if (i == 0 && j == 0) return 0.0;
        if (i == 0 || j == 0) return INFINITY;
        return D1[(size_t)(i - 1) * c + (j - 1)];
    };
    int i = r - 1, j = c - 1, pos = r + c - 1;
    path_i[pos] = i;
    path_j[pos] = j;
    while (i > 0 || j > 0) {
        const double dg = D0(i, j), up = D0(i, j + 1), lf = D0(i + 1, j);
        int tb = 0;
        double m = dg;
        if (up < m) { m = up; tb = 1; }
        if (lf < m) { tb = 2; }
        if (tb == 0) { --i; --j; }
        else if (tb == 1) --i;
        else --j;
        --pos;
        path_i[pos] = i;
        path_j[pos] = j;
    }
    *path_len = r + c - pos;
}

}  // namespace ssp

using namespace ssp;

extern "C" int ssp_dtw_distances(ssp_ctx* ctx, const float* xq, const ssp_segments* q_seg, const float* xt,
                                 const ssp_segments* t_seg, int32_t dim, int32_t normalize, float* dist_out, int where,
                                 float* kernel_ms) {
    ssp::TraceRange trace_("ssp_dtw_distances");
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (!q_seg || !t_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_dtw_distances: null segments");
    if (dim < 1) SSP_FAIL(SSP_ERR_INVALID, "ssp_dtw_distances: dim");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_dtw_distances: where");
    const int64_t n_q = q_seg->n, n_t = t_seg->n, n_pairs = n_q * n_t;
    if (n_pairs == 0) return SSP_OK;
    if (!dist_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_dtw_distances: null output");
    const int64_t rows_q = q_seg->host.back(), rows_t = t_seg->host.back();
    if ((rows_q > 0 && !xq) || (rows_t > 0 && !xt)) SSP_FAIL(SSP_ERR_INVALID, "ssp_dtw_distances: null data");
    const int64_t max_r = std::max<int64_t>(q_seg->max_len(), 1), max_c = t_seg->max_len();
    if (max_r > INT32_MAX / 2 || max_c > INT32_MAX / 2 || n_q > INT32_MAX || n_t > INT32_MAX)
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_dtw_distances: sequence too long");
    hipStream_t s = ctx->stream;
    Staged sq, st, so;
    int rc;
    const float* dq = (const float*)sq.in(ctx, xq, (size_t)rows_q * dim * sizeof(float), where, &rc);
    SSP_TRY(rc);
    const float* dt = (const float*)st.in(ctx, xt, (size_t)rows_t * dim * sizeof(float), where, &rc);
    SSP_TRY(rc);
    float* dout = (float*)so.out(ctx, dist_out, (size_t)n_pairs * sizeof(float), where, &rc);
    SSP_TRY(rc);
    // column block per lane: the smallest of {4, 8, (12,) 16, (20, 24,) 32} that covers the longest template in one super-block, else 32
    const int need = (int)((max_c + 63) / 64);
    int W = need <= 4 ? 4 : need <= 8 ? 8 : need <= 16 ? 16 : 32;
    if (dim == 1) W = need <= 4 ? 4 : need <= 8 ? 8 : need <= 12 ? 12 : need <= 16 ? 16 : need <= 20 ? 20 : need <= 24 ? 24 : 32;
    const bool multi = max_c > 64 * (int64_t)W;
    DevBuf bnd;
    SSP_TRY(bnd.alloc(multi ? (size_t)n_pairs * max_r * sizeof(float) : 16));
    DtwArgs a{dq, dt, q_seg->dev.as<int64_t>(), t_seg->dev.as<int64_t>(), dout, bnd.p, nullptr, n_pairs,
              (int32_t)n_q, (int32_t)n_t, dim, normalize ? 1 : 0, (int32_t)max_r};
    const int64_t grid = (n_pairs + 3) / 4;
    if (grid > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_dtw_distances: too many pairs");
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    if (dim == 1) {
        switch (W) {
            case 4: hipLaunchKernelGGL(dtw1_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 8: hipLaunchKernelGGL(dtw1_kernel<8>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 12: hipLaunchKernelGGL(dtw1_kernel<12>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 20: hipLaunchKernelGGL(dtw1_kernel<20>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 24: hipLaunchKernelGGL(dtw1_kernel<24>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 16: hipLaunchKernelGGL(dtw1_kernel<16>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL(dtw1_kernel<32>, dim3((unsigned)grid), dim3(256), 0, s, a); break;
        }
    } else {
        switch (W) {
            case 4: hipLaunchKernelGGL((dtw_kernel<4, float, false>), dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 8: hipLaunchKernelGGL((dtw_kernel<8, float, false>), dim3((unsigned)grid), dim3(256), 0, s, a); break;
            case 16: hipLaunchKernelGGL((dtw_kernel<16, float, false>), dim3((unsigned)grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((dtw_kernel<32, float, false>), dim3((unsigned)grid), dim3(256), 0, s, a); break;
        }
    }
    SSP_HIP(hipGetLastError());
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(so.back(ctx, dist_out, (size_t)n_pairs * sizeof(float), where));
    SSP_HIP(hipStreamSynchronize(s));  // `bnd` and the staging buffers are freed at return
    return SSP_OK;
}

extern "C" int ssp_dtw_path(ssp_ctx* ctx, const float* x, int64_t r, const float* y, int64_t c, int32_t dim, double* dist_out,
                            int32_t* path_i_out, int32_t* path_j_out, int32_t* path_len_out) {
    SSP_TRY(use_ctx(ctx));
    if (r < 1 || c < 1 || dim < 1 || !x || !y || !dist_out || !path_i_out || !path_j_out || !path_len_out)
        SSP_FAIL(SSP_ERR_INVALID, "ssp_dtw_path: bad argument");
    if (r > (1 << 20) || c > (1 << 20) || (double)r * (double)c > 5.0e8) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_dtw_path: sequences too long");
    hipStream_t s = ctx->stream;
    const int64_t off_q[2] = {0, r}, off_t[2] = {0, c};
    DevBuf dx, dy, doq, dot, dD, dbnd, dout, dpi, dpj, dlen;
    SSP_TRY(dx.alloc((size_t)r * dim * sizeof(float)));
    SSP_TRY(dy.alloc((size_t)c * dim * sizeof(float)));
    SSP_TRY(doq.alloc(sizeof(off_q)));
    SSP_TRY(dot.alloc(sizeof(off_t)));
    SSP_TRY(dD.alloc((size_t)r * c * sizeof(double)));
    SSP_TRY(dbnd.alloc((size_t)r * sizeof(double)));
    SSP_TRY(dout.alloc(sizeof(double)));
    SSP_TRY(dpi.alloc((size_t)(r + c) * sizeof(int32_t)));
    SSP_TRY(dpj.alloc((size_t)(r + c) * sizeof(int32_t)));
    SSP_TRY(dlen.alloc(sizeof(int32_t)));
    SSP_HIP(hipMemcpyAsync(dx.p, x, (size_t)r * dim * sizeof(float), hipMemcpyHostToDevice, s));
    SSP_HIP(hipMemcpyAsync(dy.p, y, (size_t)c * dim * sizeof(float), hipMemcpyHostToDevice, s));
    SSP_HIP(hipMemcpyAsync(doq.p, off_q, sizeof(off_q), hipMemcpyHostToDevice, s));
    SSP_HIP(hipMemcpyAsync(dot.p, off_t, sizeof(off_t), hipMemcpyHostToDevice, s));
    DtwArgs a{dx.as<float>(), dy.as<float>(), doq.as<int64_t>(), dot.as<int64_t>(), dout.p, dbnd.p, dD.p, 1, 1, 1, dim, 0, (int32_t)r};
    hipLaunchKernelGGL((dtw_kernel<16, double, true>), dim3(1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(dtw_traceback_kernel, dim3(1), dim3(64), 0, s, dD.as<double>(), (int)r, (int)c, dpi.as<int32_t>(), dpj.as<int32_t>(),
                       dlen.as<int32_t>());
    SSP_HIP(hipGetLastError());
    std::vector<int32_t> pi((size_t)(r + c)), pj((size_t)(r + c));
    int32_t len = 0;
    SSP_HIP(hipMemcpyAsync(dist_out, dout.p, sizeof(double), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipMemcpyAsync(pi.data(), dpi.p, pi.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipMemcpyAsync(pj.data(), dpj.p, pj.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipMemcpyAsync(&len, dlen.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipStreamSynchronize(s));
    if (len < 1 || len > r + c) SSP_FAIL(SSP_ERR_HIP, "ssp_dtw_path: traceback produced an invalid length %d", len);
    memcpy(path_i_out, pi.data() + (r + c - len), (size_t)len * sizeof(int32_t));
    memcpy(path_j_out, pj.data() + (r + c - len), (size_t)len * sizeof(int32_t));
    *path_len_out = len;
    return SSP_OK;
}

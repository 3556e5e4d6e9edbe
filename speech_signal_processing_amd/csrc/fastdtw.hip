// FastDTW distances on gfx950 — the dtw_method = 2 branch of the reference's matcher (MFCC_DTW.py:69-70:
// `d, path = fastdtw(sample_x, sample_y, dist=euclidean)`, radius 1).  The fastdtw package is absent from the reference tree and this
// image (parity unpinned); its published algorithm (Salvador & Chan, as the package's `__fastdtw` states it) is restated here:
//   fastdtw(x, y):  if len(x) < radius + 2 or len(y) < radius + 2: full DTW
//                   else: path = fastdtw(halve(x), halve(y)); window = expand(path, radius); DTW restricted to the window
//   halve(x)[i] = (x[2 i] + x[2 i + 1]) / 2  (a trailing odd element is dropped);  float64 throughout, like the package
//   window: every path cell grows to its (2 radius + 1)^2 neighbourhood, every cell of that set becomes the 2 x 2 block of fine cells;
//           row i of the fine grid keeps ONE contiguous run of columns — because the path is monotone that run is
//           [2 (min j' - radius), 2 (max j' + radius) + 1] over the path cells of coarse rows (i >> 1) - radius .. (i >> 1) + radius
//   DTW step: D[i][j] = |x_i - y_j| + min(D[i-1][j], D[i][j-1], D[i-1][j-1]) (outside the window: +inf), the FIRST minimum in that
//           order names the predecessor (Python's min over the tuple list) — the order decides the path and with it the next window
// The recursion is data dependent and O(N): one THREAD walks one (query, template) pair through all its levels with per-pair scratch
// in global memory (levels of halved series, the per-row column range of the path, one predecessor byte per window cell, two rolling
// rows of D).  Pairs are independent, so a test set against all templates is one launch.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace ssp {

struct FdtwArgs {
    const float* xq;
    const float* xt;
    const int64_t* q_off;
    const int64_t* t_off;
    double* out;           // [n_q x n_t]
    char* scratch;         // per pair: pair_bytes
    int64_t pair_bytes;
    int64_t n_pairs, pair0;
    int32_t n_t, radius, max_r, max_c, cell_cap;
    int32_t* err;          // set to 1 when a pair's window outgrew cell_cap
};

__global__ __launch_bounds__(64) void fastdtw_kernel(FdtwArgs a) {
    const int64_t pid = a.pair0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (pid >= a.n_pairs) return;
    const int q = (int)(pid / a.n_t), p = (int)(pid - (int64_t)q * a.n_t);
    const int nx0 = (int)(a.q_off[q + 1] - a.q_off[q]), ny0 = (int)(a.t_off[p + 1] - a.t_off[p]);
    const float* __restrict__ x0 = a.xq + a.q_off[q];
    const float* __restrict__ y0 = a.xt + a.t_off[p];
    if (nx0 <= 0 || ny0 <= 0) {
        a.out[pid] = INFINITY;
        return;
    }
    // ---- per-pair scratch carve
    char* base = a.scratch + (pid - a.pair0) * a.pair_bytes;
    double* xs = reinterpret_cast<double*>(base);                 // levels of x back to back: 2 max_r doubles
    double* ys = xs + 2 * (size_t)a.max_r;                        // levels of y: 2 max_c doubles
    double* rowA = ys + 2 * (size_t)a.max_c;                      // two rolling rows of D, max_c + 2 each
    double* rowB = rowA + (a.max_c + 2);
    int32_t* pmin = reinterpret_cast<int32_t*>(rowB + (a.max_c + 2));  // per row of the level below: min / max column of its path cells
    int32_t* pmax = pmin + (a.max_r + 2);
    int32_t* lo = pmax + (a.max_r + 2);                           // window of the current level
    int32_t* hi = lo + (a.max_r + 2);
    int32_t* off = hi + (a.max_r + 2);                            // first cell of every row in the predecessor array
    unsigned char* bp = reinterpret_cast<unsigned char*>(off + (a.max_r + 2));  // [cell_cap] 0 = up, 1 = left, 2 = diagonal
    // ---- levels: halve until one side is shorter than radius + 2
    const int min_size = a.radius + 2;
    int nxl[32], nyl[32], xo[32], yo[32];
    int L = 0;
    nxl[0] = nx0;
    nyl[0] = ny0;
    xo[0] = yo[0] = 0;
    for (int i = 0; i < nx0; ++i) xs[i] = (double)x0[i];
    for (int j = 0; j < ny0; ++j) ys[j] = (double)y0[j];
    while (nxl[L] >= min_size && nyl[L] >= min_size && L < 31) {
        const int nx = nxl[L], ny = nyl[L];
        xo[L + 1] = xo[L] + nx;
        yo[L + 1] = yo[L] + ny;
        nxl[L + 1] = nx / 2;
        nyl[L + 1] = ny / 2;
        for (int i = 0; i < nx / 2; ++i) xs[xo[L + 1] + i] = (xs[xo[L] + 2 * i] + xs[xo[L] + 2 * i + 1]) / 2;
        for (int j = 0; j < ny / 2; ++j) ys[yo[L + 1] + j] = (ys[yo[L] + 2 * j] + ys[yo[L] + 2 * j + 1]) / 2;
        ++L;
    }
    // ---- from the coarsest level (full DTW) up to level 0 (windowed DTW)
    double dist = 0.0;
    bool overflow = false;
    for (int lev = L; lev >= 0 && !overflow; --lev) {
        const int nx = nxl[lev], ny = nyl[lev];
        const double* __restrict__ x = xs + xo[lev];
        const double* __restrict__ y = ys + yo[lev];
        // window rows
        if (lev == L) {
            for (int i = 0; i < nx; ++i) {
                lo[i] = 0;
                hi[i] = ny - 1;
            }
        } else {
            const int nxc = nxl[lev + 1], r = a.radius;
            int prev_lo = 0;
            for (int i = 0; i < nx; ++i) {
                const int ic = i >> 1;
                int mn = 0x7fffffff, mx = -0x7fffffff;
                for (int k = max(ic - r, 0); k <= min(ic + r, nxc - 1); ++k) {
                    mn = min(mn, pmin[k]);
                    mx = max(mx, pmax[k]);
                }
                int l = max(0, 2 * (mn - r)), h = min(ny - 1, 2 * (mx + r) + 1);
                // the package scans a row from the previous row's first column: a run that starts before it is cut there
                l = max(l, prev_lo);
                lo[i] = l;
                hi[i] = h;
                prev_lo = l;
            }
        }
        int64_t cells = 0;
        for (int i = 0; i < nx; ++i) {
            off[i] = (int32_t)cells;
            cells += hi[i] >= lo[i] ? hi[i] - lo[i] + 1 : 0;
        }
        if (cells > a.cell_cap) {
            overflow = true;
            break;
        }
        // DP over the window, row by row; prev = row i - 1 (indexed by column + 1), cur = row i
        double* prev = rowA;
        double* cur = rowB;
        int plo = 0, phi = -1;  // window of the previous row (none before row 0)
        for (int i = 0; i < nx; ++i) {
            const int l = lo[i], h = hi[i];
            const double xi = x[i];
            double left = INFINITY;  // D[i][l - 1]: outside the window
            for (int j = l; j <= h; ++j) {
                const double dt = fabs(xi - y[j]);
                double up = (j >= plo && j <= phi) ? prev[j] : INFINITY;
                double dg = (j - 1 >= plo && j - 1 <= phi) ? prev[j - 1] : INFINITY;
                if (i == 0) {
                    up = INFINITY;
                    dg = j == 0 ? 0.0 : INFINITY;  // D[0, 0] = 0 sits diagonally before cell (0, 0)
                }
                if (j == 0 && i > 0) dg = INFINITY;
                // first minimum in the order up, left, diagonal
                double best = up + dt;
                unsigned char who = 0;
                if (left + dt < best) {
                    best = left + dt;
                    who = 1;
                }
                if (dg + dt < best) {
                    best = dg + dt;
                    who = 2;
                }
                cur[j] = best;
                bp[off[i] + (j - l)] = who;
                left = best;
            }
            double* t = prev;
            prev = cur;
            cur = t;
            plo = l;
            phi = h;
        }
        dist = prev[ny - 1];  // D[nx][ny]
        if (lev > 0) {
            // walk the path back from (nx - 1, ny - 1), recording the column range of its cells per row
            for (int i = 0; i < nx; ++i) {
                pmin[i] = 0x7fffffff;
                pmax[i] = -0x7fffffff;
            }
            int i = nx - 1, j = ny - 1;
            for (;;) {
                pmin[i] = min(pmin[i], j);
                pmax[i] = max(pmax[i], j);
                if (i == 0 && j == 0) break;
                if (j < lo[i] || j > hi[i]) {  // (cannot happen for a finite distance)
                    overflow = true;
                    break;
                }
                const unsigned char who = bp[off[i] + (j - lo[i])];
                if (who == 0) --i;
                else if (who == 1) --j;
                else {
                    --i;
                    --j;
                }
                if (i < 0 || j < 0) {
                    overflow = true;
                    break;
                }
            }
        }
    }
    if (overflow) {
        *a.err = 1;
        a.out[pid] = NAN;
    } else {
        a.out[pid] = dist;
    }
}

}  // namespace ssp

using namespace ssp;

extern "C" int ssp_fastdtw_distances(ssp_ctx* ctx, const float* xq, const ssp_segments* q_seg, const float* xt, const ssp_segments* t_seg,
                                     int32_t radius, double* dist_out, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_fastdtw_distances");
    SSP_TRY(use_ctx(ctx));
    if (kernel_ms) *kernel_ms = 0.f;
    if (!q_seg || !t_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_fastdtw_distances: null segments");
    if (radius < 1 || radius > 16) SSP_FAIL(SSP_ERR_INVALID, "ssp_fastdtw_distances: radius must be in [1, 16] (the package itself fails on odd lengths at radius 0)");
    const int64_t n_q = q_seg->n, n_t = t_seg->n, n_pairs = n_q * n_t;
    if (n_pairs == 0) return SSP_OK;
    if (!dist_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_fastdtw_distances: null output");
    const int64_t rows_q = q_seg->host.back(), rows_t = t_seg->host.back();
    if ((rows_q > 0 && !xq) || (rows_t > 0 && !xt)) SSP_FAIL(SSP_ERR_INVALID, "ssp_fastdtw_distances: null data");
    const int64_t max_r = std::max<int64_t>(q_seg->max_len(), 1), max_c = std::max<int64_t>(t_seg->max_len(), 1);
    if (max_r > (1 << 24) || max_c > (1 << 24) || n_t > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_fastdtw_distances: sequence too long");
    hipStream_t s = ctx->stream;
    Staged sq, st;
    int rc;
    const float* dq = (const float*)sq.in(ctx, xq, (size_t)rows_q * sizeof(float), SSP_HOST, &rc);
    SSP_TRY(rc);
    const float* dt = (const float*)st.in(ctx, xt, (size_t)rows_t * sizeof(float), SSP_HOST, &rc);
    SSP_TRY(rc);
    // window cells of a level: every path cell (at most nx + ny of them one level down) becomes <= (2 radius + 1) columns over
    // (2 radius + 1) rows, doubled in both directions; the coarsest level is a full matrix of at most (2 radius + 4)^2 x ... cells
    const int64_t cell_cap = std::max<int64_t>(4 * (2 * radius + 2) * (max_r + max_c) + 64, (int64_t)(2 * radius + 4) * std::max(max_r, max_c) + 64);
    const int64_t pair_bytes =
        ((int64_t)(2 * max_r + 2 * max_c + 2 * (max_c + 2)) * 8 + (int64_t)5 * (max_r + 2) * 4 + cell_cap + 15) & ~(int64_t)15;
    const int64_t budget = (int64_t)2 << 30;
    const int64_t per_launch = std::max<int64_t>(64, std::min<int64_t>(n_pairs, (budget / pair_bytes) / 64 * 64));
    DevBuf scratch, dout, derr;
    SSP_TRY(scratch.alloc((size_t)std::min(per_launch, n_pairs) * pair_bytes));
    SSP_TRY(dout.alloc((size_t)n_pairs * sizeof(double)));
    SSP_TRY(derr.alloc(sizeof(int32_t)));
    SSP_HIP(hipMemsetAsync(derr.p, 0, sizeof(int32_t), s));
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    for (int64_t p0 = 0; p0 < n_pairs; p0 += per_launch) {
        FdtwArgs a{dq, dt, q_seg->dev.as<int64_t>(), t_seg->dev.as<int64_t>(), dout.as<double>(), scratch.as<char>(), pair_bytes,
                   std::min(n_pairs, p0 + per_launch), p0, (int32_t)n_t, radius, (int32_t)max_r, (int32_t)max_c, (int32_t)std::min<int64_t>(cell_cap, INT32_MAX),
                   derr.as<int32_t>()};
        const int64_t grid = (std::min(per_launch, n_pairs - p0) + 63) / 64;
        hipLaunchKernelGGL(fastdtw_kernel, dim3((unsigned)grid), dim3(64), 0, s, a);
        SSP_HIP(hipGetLastError());
    }
    SSP_TRY(tm.stop(s, kernel_ms));
    int32_t err = 0;
    SSP_HIP(hipMemcpyAsync(dist_out, dout.p, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipMemcpyAsync(&err, derr.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SSP_HIP(hipStreamSynchronize(s));
    if (err) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_fastdtw_distances: a warping window outgrew its scratch (pathological path)");
    return SSP_OK;
}

/*
 * score_shard.c — the C-ABI of libsspgpu.so from plain C: what one rank of configs[3] does.
 *
 *   utterances of this rank's shard  ->  ssp_mfcc_run (39-d MFCC + delta + delta-delta, device resident)
 *                                    ->  ssp_gmm_score against a UBM + S speaker models (GMM_UBM.py:181-197)
 *                                    ->  12-byte decision records (int32 argmax, float best - ubm, float ubm)
 *                                    ->  ssp_allgather over RCCL (a world of one here; with N ranks every rank
 *                                        passes the 128-byte id made by rank 0 to ssp_comm_init)
 *
 * Build:  gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/score_shard.c -o score_shard -L speech_signal_processing_amd -lsspgpu \
 *             -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/speech_signal_processing_amd -Wl,-rpath,/opt/rocm/lib
 * Run  :  ./score_shard            (needs an MI355X; prints "OK ..." and exits 0)
 *
 * The dialect tables (window, filterbank, DCT matrix) are built here the way frontend.preset_sidekit builds them — they are host
 * metadata, not part of the library.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssp.h"

#define CHECK(expr)                                                                   \
    do {                                                                              \
        int rc_ = (expr);                                                             \
        if (rc_ != SSP_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, ssp_last_error());    \
            return 1;                                                                 \
        }                                                                             \
    } while (0)
#define HIP(expr)                                                                     \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_));         \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static double hz2mel(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel2hz(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

int main(void) {
    enum { FS = 16000, WIN = 400, HOP = 160, NFFT = 512, NB = NFFT / 2 + 1, NFILT = 24, NCEPS = 13, D = 39 };
    enum { U = 64, NSAMP = 16000, K = 8, S = 5, M = S + 1 };

    /* ---- dialect tables: sidekit mfcc as called at GMM_UBM.py:89 (hanning window, 24 HTK-mel triangles 100..8000 Hz, c1..c13) */
    static float window[WIN], fbank[NFILT * NB], dct[NCEPS * NFILT];
    for (int n = 0; n < WIN; ++n) window[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / (WIN - 1)));
    double edges[NFILT + 2];
    for (int i = 0; i < NFILT + 2; ++i) edges[i] = mel2hz(hz2mel(100.0) + (hz2mel(8000.0) - hz2mel(100.0)) * i / (NFILT + 1));
    memset(fbank, 0, sizeof fbank);
    for (int i = 0; i < NFILT; ++i) {
        const double lo = edges[i], ce = edges[i + 1], hi = edges[i + 2], h = 2.0 / (hi - lo);
        const int b_lo = (int)floor(lo * NFFT / FS) + 1, b_ce = (int)floor(ce * NFFT / FS), b_hi = (int)floor(hi * NFFT / FS) + 1;
        for (int k = b_lo; k <= b_ce && k < NB; ++k) fbank[i * NB + k] = (float)(h * (k * (double)FS / NFFT - lo) / (ce - lo));
        for (int k = b_ce + 1; k < (b_hi < NFFT ? b_hi : NFFT) - 1 && k < NB; ++k) fbank[i * NB + k] = (float)(h * (hi - k * (double)FS / NFFT) / (hi - ce));
    }
    for (int q = 0; q < NCEPS; ++q)
        for (int j = 0; j < NFILT; ++j) dct[q * NFILT + j] = (float)(sqrt(2.0 / NFILT) * cos(M_PI * (q + 1) * (2 * j + 1) / (2.0 * NFILT)));

    ssp_mfcc_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.sample_rate = FS; cfg.win_len = WIN; cfg.hop = HOP; cfg.n_fft = NFFT; cfg.n_filt = NFILT; cfg.n_ceps = NCEPS;
    cfg.frame_mode = 0; cfg.preemph_mode = 1; cfg.preemph = 0.97f; cfg.spec_power = 2; cfg.spec_scale = 1.0f;
    cfg.log_mode = 0; cfg.floor_mode = 0; cfg.eps = 0.f; cfg.top_db = -1.f; cfg.delta_order = 2; cfg.delta_N = 2; cfg.cmvn = 0;

    ssp_ctx* ctx = NULL;
    CHECK(ssp_ctx_create(0, NULL, 0, &ctx));
    ssp_mfcc_plan* plan = NULL;
    CHECK(ssp_mfcc_plan_create(ctx, &cfg, window, fbank, dct, &plan));

    /* ---- this rank's shard: U utterances of 1 s (host metadata: offsets), samples resident on the device */
    int64_t off[U + 1];
    for (int u = 0; u <= U; ++u) off[u] = (int64_t)u * NSAMP;
    ssp_segments *sseg = NULL, *fseg = NULL;
    CHECK(ssp_segments_create(ctx, off, U, &sseg));
    CHECK(ssp_mfcc_frame_segments(plan, sseg, &fseg));
    int64_t n_seg = 0, n_frames = 0;
    CHECK(ssp_segments_count(fseg, &n_seg, &n_frames));

    float* h_x = (float*)malloc(sizeof(float) * U * NSAMP);
    unsigned rs = 12345u;
    for (int u = 0; u < U; ++u)
        for (int n = 0; n < NSAMP; ++n) {
            rs = rs * 1664525u + 1013904223u;
            h_x[u * NSAMP + n] = (float)(0.3 * sin(2.0 * M_PI * (100.0 + 7.0 * (u % S)) * n / FS) + 0.05 * ((rs >> 8) / 8388608.0 - 1.0));
        }
    float *d_x = NULL, *d_feat = NULL, *d_scores = NULL;
    int32_t* d_arg = NULL;
    HIP(hipMalloc((void**)&d_x, sizeof(float) * U * NSAMP));
    HIP(hipMalloc((void**)&d_feat, sizeof(float) * n_frames * D));
    HIP(hipMalloc((void**)&d_scores, sizeof(float) * U * M));
    HIP(hipMalloc((void**)&d_arg, sizeof(int32_t) * U));
    HIP(hipMemcpy(d_x, h_x, sizeof(float) * U * NSAMP, hipMemcpyHostToDevice));
    float ms_mfcc = 0.f, ms_gmm = 0.f;
    CHECK(ssp_mfcc_run(plan, sseg, fseg, d_x, d_feat, SSP_DEVICE, 0, &ms_mfcc));

    /* ---- models: a UBM and S speaker models with shifted means (diag covariance), parameters from the features' statistics */
    float* h_feat = (float*)malloc(sizeof(float) * n_frames * D);
    HIP(hipMemcpy(h_feat, d_feat, sizeof(float) * n_frames * D, hipMemcpyDeviceToHost));
    double mean[D], var[D];
    for (int d = 0; d < D; ++d) {
        double s1 = 0, s2 = 0;
        for (int64_t t = 0; t < n_frames; ++t) { s1 += h_feat[t * D + d]; s2 += (double)h_feat[t * D + d] * h_feat[t * D + d]; }
        mean[d] = s1 / n_frames; var[d] = s2 / n_frames - mean[d] * mean[d] + 1e-3;
    }
    static double w[M * K], mu[M * K * D], cov[M * K * D];
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            w[m * K + k] = 1.0 / K;
            for (int d = 0; d < D; ++d) {
                rs = rs * 1664525u + 1013904223u;
                const double z = (rs >> 8) / 8388608.0 - 1.0;
                mu[(m * K + k) * D + d] = mean[d] + sqrt(var[d]) * (0.8 * sin(1.7 * k + 0.3 * d) + (m ? 0.3 * z : 0.0));
                cov[(m * K + k) * D + d] = var[d];
            }
        }
    ssp_gmm* gmm = NULL;
    CHECK(ssp_gmm_pack(ctx, M, K, D, w, mu, cov, 1 /* model 0 is the UBM */, &gmm));
    CHECK(ssp_gmm_score(gmm, d_feat, fseg, NULL, d_scores, d_arg, SSP_DEVICE, 0 /* fp32 parity path */, &ms_gmm));

    /* ---- decision records and the one exchange step of the path */
    static float h_scores[U * M];
    static int32_t h_arg[U], rec[U * 3], all[U * 3];
    HIP(hipMemcpy(h_scores, d_scores, sizeof h_scores, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_arg, d_arg, sizeof h_arg, hipMemcpyDeviceToHost));
    for (int u = 0; u < U; ++u) {
        const float ubm = h_scores[u * M], best = h_scores[u * M + 1 + h_arg[u]] - ubm;
        rec[3 * u] = h_arg[u];
        memcpy(&rec[3 * u + 1], &best, 4);
        memcpy(&rec[3 * u + 2], &ubm, 4);
    }
    int32_t *d_rec = NULL, *d_all = NULL;
    HIP(hipMalloc((void**)&d_rec, sizeof rec));
    HIP(hipMalloc((void**)&d_all, sizeof all));
    HIP(hipMemcpy(d_rec, rec, sizeof rec, hipMemcpyHostToDevice));
    unsigned char uid[SSP_COMM_ID_BYTES];
    CHECK(ssp_comm_unique_id(uid));           /* rank 0; with N ranks: ship these 128 bytes to the others */
    CHECK(ssp_comm_init(ctx, 0, 1, uid));     /* rank, nranks */
    CHECK(ssp_allgather(ctx, d_rec, d_all, sizeof rec));
    CHECK(ssp_ctx_sync(ctx));
    HIP(hipMemcpy(all, d_all, sizeof all, hipMemcpyDeviceToHost));
    CHECK(ssp_comm_destroy(ctx));
    int bad = memcmp(all, rec, sizeof rec) != 0;
    /* the arg-max must be the arg-max of the score differences */
    for (int u = 0; u < U && !bad; ++u) {
        int am = 0;
        for (int s = 1; s < S; ++s)
            if (h_scores[u * M + 1 + s] - h_scores[u * M] > h_scores[u * M + 1 + am] - h_scores[u * M]) am = s;
        if (am != h_arg[u]) bad = 1;
        if (!(h_scores[u * M] == h_scores[u * M]) ) bad = 1; /* NaN */
    }
    printf("%s: %d utterances, %lld frames x %d-d in %.3f ms, %d models x %d mixtures scored in %.3f ms, %d records gathered\n",
           bad ? "MISMATCH" : "OK", U, (long long)n_frames, D, ms_mfcc, M, K, ms_gmm, U);

    ssp_gmm_destroy(gmm);
    ssp_segments_destroy(fseg);
    ssp_segments_destroy(sseg);
    ssp_mfcc_plan_destroy(plan);
    ssp_ctx_destroy(ctx);
    hipFree(d_x); hipFree(d_feat); hipFree(d_scores); hipFree(d_arg); hipFree(d_rec); hipFree(d_all);
    free(h_x); free(h_feat);
    return bad;
}

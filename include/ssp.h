/*
 * ssp.h — C-ABI of libsspgpu.so: the MI355X (gfx950) drop-in for the MFCC -> GMM-UBM /
 * d-vector scoring hot path of kleinzcy/speech_signal_processing.
 *
 * The reference is pure Python and has no FFI for this path; each entry
 * point below names the reference code it replaces.  Host code (Python, ctypes) keeps the
 * reference's call surface and dispatches here.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns SSP_OK (0) or a negative ssp_status; ssp_last_error() returns a
 *     thread-local message for the last failure on the calling thread.  Nothing aborts.
 *   - `where` = SSP_HOST (0): bulk arrays are host pointers, the library stages them through
 *     device scratch the ctx keeps between calls (up to 8 buffers of at most 64 MiB; larger
 *     operands get a buffer of their own for the call — except the batches of ssp_mfcc_run(_i16),
 *     ssp_gmm_score (precision 0 / 2) and ssp_cosine_identify(2) (precision 0, arg-min / minimum)
 *     above two slices (SSP_HOST_SLICE_MB, 64 MiB): those go through a ring of three slice-sized
 *     slots, copied in ahead of the kernels that consume them);  SSP_DEVICE (1): bulk arrays are
 *     device pointers on the ctx's device.
 *   - segment offsets (per-utterance sample / frame offsets) are small host-side metadata:
 *     they are always HOST int64 arrays and are uploaded once into an ssp_segments handle.
 *   - one ssp_ctx = one HIP device + one stream.  A ctx is not thread-safe; distinct ctxs are.
 *   - all kernels are asynchronous on the ctx stream; SSP_HOST calls and calls that return
 *     kernel_ms synchronise before returning.  kernel_ms (nullable) receives the device time
 *     of the call's kernels measured with hipEvents on the ctx stream.
 */
#ifndef SSP_H_
#define SSP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSP_ABI_VERSION 4 /* 4: ssp_mfcc_run_i16, sliced host-fed ssp_mfcc_run, ssp_calibrate, precision = auto (ssp_gmm_score 4, ssp_cosine_identify2 3); 3: ssp_cosine_identify2 (precision), ssp_mfcc_plan_set_flags; 2: ssp_comm_* / ssp_allgather / ssp_allreduce_sum; ssp_gmm_score precision = 1 re-scores close calls in fp32 (host sync) */

typedef enum {
    SSP_OK = 0,
    SSP_ERR_INVALID = -1,     /* bad argument / shape */
    SSP_ERR_UNSUPPORTED = -2, /* valid request the kernels do not cover */
    SSP_ERR_HIP = -3,         /* HIP runtime error (message has hipGetErrorString) */
    SSP_ERR_NOMEM = -4,
    SSP_ERR_NODEVICE = -5
} ssp_status;

enum { SSP_HOST = 0, SSP_DEVICE = 1 };

typedef struct ssp_ctx ssp_ctx;
typedef struct ssp_segments ssp_segments;
typedef struct ssp_mfcc_plan ssp_mfcc_plan;
typedef struct ssp_gmm ssp_gmm;
typedef struct ssp_dnn ssp_dnn;           /* a fully connected network packed for the MFMA forward pass */

/* MFCC dialect knobs.  Three presets are built by the host side:
 *   in-repo  utils/processing.py:19-144  (Hamming, |X|/L, 40 talkbox filters folded, log10(.+1e-8), c0..c12)
 *   sidekit  sidekit.frontend.features.mfcc as called at GMM_UBM.py:89 / d_vector.py:91
 *   librosa  librosa.feature.mfcc as called at MFCC_DTW.py:28-31 */
typedef struct {
    int32_t sample_rate;
    int32_t win_len;      /* samples per analysis window (<= n_fft) */
    int32_t hop;
    int32_t n_fft;        /* power of two, 64..4096 */
    int32_t n_filt;       /* rows of the filterbank table */
    int32_t n_ceps;       /* rows of the DCT table */
    int32_t frame_mode;   /* 0 floor, no pad (sidekit) | 1 ceil, zero-pad tail (utils/processing.py:27) | 2 centred, reflect (librosa) */
    int32_t preemph_mode; /* 0 none | 1 per frame: y[0]=x[0]-a*x[0], y[n]=x[n]-a*x[n-1] */
    float preemph;
    int32_t spec_power;   /* 1 |X| | 2 |X|^2 */
    float spec_scale;     /* multiplies the magnitude / power (1/L for the in-repo dialect, utils/processing.py:139) */
    int32_t log_mode;     /* 0 ln | 1 log10 | 2 10*log10 */
    int32_t floor_mode;   /* 0 none | 1 log(x+eps) (utils/processing.py:105) | 2 log(max(eps,x)) (librosa power_to_db) */
    float eps;
    float top_db;         /* <0 off; else clamp log-mel to (utterance max - top_db) (librosa power_to_db) */
    int32_t delta_order;  /* 0 | 1 [c,dc] (GMM_UBM.py:90-91) | 2 [c,dc,ddc] */
    int32_t delta_N;      /* regression half-width (GMM_UBM.py:53: N=2) */
    int32_t cmvn;         /* 1: per-utterance sklearn.preprocessing.scale (GMM_UBM.py:93) */
} ssp_mfcc_cfg;

int ssp_abi_version(void);
const char* ssp_last_error(void);

/* ---- context: device + stream ------------------------------------------------------- */
/* borrow_stream != 0: `stream` is the caller's hipStream_t (e.g. torch's current stream; NULL = the HIP default
 * stream) and all work is enqueued on it;  borrow_stream == 0: the library creates and owns a stream. */
int ssp_ctx_create(int device, void* stream, int borrow_stream, ssp_ctx** out);
int ssp_ctx_destroy(ssp_ctx* ctx);
/* test aid: fill the LDS of every CU with one 32-bit pattern (e.g. a NaN), so that a kernel reading LDS it never wrote shows up
 * deterministically in the parity tests instead of depending on what the previous kernel left behind */
int ssp_debug_poison_lds(ssp_ctx* ctx, uint32_t pattern);
int ssp_ctx_sync(ssp_ctx* ctx);
/* measurement aid (bench.py `env.calibration`): what this box sustains on two textbook loads, each run for about target_ms
 * milliseconds (<= 0: 20) on the ctx stream and timed with HIP events — a float4 copy of 1 GiB buffers (GB/s, read + write) and
 * eight packed-fp32 FMA chains per lane on every SIMD (TFLOP/s).  Blocks until both are measured; copy_ms / fma_ms may be NULL. */
int ssp_calibrate(ssp_ctx* ctx, double target_ms, double* copy_gbs, double* fma_tflops, double* copy_ms, double* fma_ms);
/* ordering against another stream of the same device without a host wait (a ctx that owns its stream, called with device pointers
 * produced / consumed on the caller's stream): wait = the ctx stream waits for everything queued on other_stream so far;
 * signal = other_stream waits for everything queued on the ctx stream so far.  other_stream: hipStream_t (NULL = the default stream) */
int ssp_ctx_wait_stream(ssp_ctx* ctx, void* other_stream);
int ssp_ctx_signal_stream(ssp_ctx* ctx, void* other_stream);

/* ---- collectives (multi-GPU: one process and one ctx per GPU) ----------
 * Utterances shard across ranks with no data-path collective (GMM_UBM.py:183-197 and d_vector.py:315-318 have no cross-utterance
 * term); models / centroids are replicated.  The one exchange step is the all-gather of the per-utterance decision records after
 * scoring — (int32 argmax, float best, float ubm) = 12 bytes per utterance.  RCCL (rings over xGMI inside a node) is loaded at
 * run time the first time a communicator is asked for; a ctx without a communicator is a world of one.
 *   rank 0: ssp_comm_unique_id(id) -> ship the 128 bytes to every rank (file, socket, MPI, torch store: the caller's transport)
 *   every rank: ssp_comm_init(ctx, rank, nranks, id)    (blocks until all ranks have called it)
 *   ssp_allgather(ctx, send, recv, bytes): DEVICE pointers; recv holds nranks * bytes, rank r's block at r * bytes; asynchronous on
 *   the ctx stream (ssp_ctx_sync before the host reads recv).  Ragged shards: gather the counts first, then padded blocks.
 *   ssp_allreduce_sum(ctx, buf, count, is_f64): in-place sum of float / double DEVICE arrays (centroid sums d_vector.py:310-313,
 *   EM sufficient statistics over utterance shards). */
#define SSP_COMM_ID_BYTES 128
int ssp_comm_unique_id(void* id_out /* HOST byte[128] */);
int ssp_comm_init(ssp_ctx* ctx, int rank, int nranks, const void* unique_id /* HOST byte[128] */);
int ssp_comm_destroy(ssp_ctx* ctx); /* (also done by ssp_ctx_destroy) */
int ssp_comm_info(const ssp_ctx* ctx, int* rank, int* nranks);
int ssp_allgather(ssp_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank);
int ssp_allreduce_sum(ssp_ctx* ctx, void* buf, size_t count, int is_f64);

/* ---- segments: per-utterance offsets (host metadata -> device-resident) ------------- */
/* offsets: HOST int64[n_seg+1], non-decreasing, offsets[0] >= 0. */
int ssp_segments_create(ssp_ctx* ctx, const int64_t* offsets, int64_t n_seg, ssp_segments** out);
int ssp_segments_destroy(ssp_segments* seg);
int ssp_segments_count(const ssp_segments* seg, int64_t* n_seg, int64_t* total);
int ssp_segments_read(const ssp_segments* seg, int64_t* offsets_out /* HOST int64[n_seg+1] */);

/* ---- MFCC: replaces utils.processing.MFCC (utils/processing.py:110-144), sidekit mfcc call
 *      sites (GMM_UBM.py:89, d_vector.py:91), librosa call site (MFCC_DTW.py:29), the delta
 *      loop (GMM_UBM.py:53-69) and preprocessing.scale (GMM_UBM.py:93) as ONE fused pass ---- */
/* window: HOST float[win_len]; fbank: HOST float[n_filt x (n_fft/2+1)] row-major (already folded
 * for the in-repo dialect); dct: HOST float[n_ceps x n_filt] row-major. */
int ssp_mfcc_plan_create(ssp_ctx* ctx, const ssp_mfcc_cfg* cfg, const float* window, const float* fbank,
                         const float* dct, ssp_mfcc_plan** out);
int ssp_mfcc_plan_destroy(ssp_mfcc_plan* plan);
/* SSP_MFCC_REPRODUCIBLE: the float32 bits of an utterance's features do not depend on the batch it is in, its place in it, or the
 * machine's CU count.  By default small batches cut utterances into short chunks for latency (each agrees with the uncut utterance to
 * rounding in the delta-delta block) and machine-filling batches scale inside the kernel (cmvn; rounding again): with the flag every
 * cut chunk recomputes 16 frames of history, which reproduces the uncut bits, scaling always runs as the stand-alone kernel and the
 * 2048-point dialects always take the second-pass clamp + DCT.  Costs latency on single-utterance calls (about 30 % more frames per
 * chunk) and about 15 % on machine-filling batches with cmvn.  (Holds for finite features: a step whose window holds a non-finite
 * cepstrum takes a term-by-term path whose finite rows agree with the matrix-core path to rounding.) */
#define SSP_MFCC_REPRODUCIBLE 1u
int ssp_mfcc_plan_set_flags(ssp_mfcc_plan* plan, uint32_t flags);
int ssp_mfcc_num_frames(const ssp_mfcc_cfg* cfg, int64_t n_samples, int64_t* n_frames);
int ssp_mfcc_out_dim(const ssp_mfcc_cfg* cfg, int32_t* d_out);
/* frame segments derived from sample segments with the plan's framing rule */
int ssp_mfcc_frame_segments(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, ssp_segments** frame_seg_out);
/* samples: float[total samples]; feats_out: float[total frames x d_out] row-major.
 * variant: 0 auto | 1 generic kernel (any cfg) | 2 fused n_fft == 512 kernel, one workgroup per utterance chunk | 3 n_fft == 512 wave-stream
 * kernel (every wave walks its own chunk; DCT / delta / delta-delta on the matrix cores; 13 cepstra, <= 40 filters, N = 2 deltas: the
 * sidekit call sites with or without scaling and the in-repo MFCC; a dense-band instance for the PLP front end; utterances may start at
 * any sample) | 4 n_fft == 2048 wave-stream kernel (no deltas: MFCC_DTW.py:28-31's librosa dialect and frameSize 2048; log filterbank rows
 * and utterance maxima, then the clamp + DCT in the same wave for single-chunk utterances or as a second pass).  An explicit 2 / 3 / 4
 * answers SSP_ERR_UNSUPPORTED when the cfg is not covered; auto picks 3, then 2 (n_fft == 512), 4 (n_fft == 2048), then 1. */
int ssp_mfcc_run(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg,
                 const float* samples, float* feats_out, int where, int variant, float* kernel_ms);
/* Host-fed batches (where = SSP_HOST — what the reference-shaped callers hand over: GMM_UBM.py:24-50 reads the wav files into host
 * arrays, :86-93 loops over them).  Up to two slices (SSP_HOST_SLICE_MB, default 64 MiB of fp32 samples) a batch is staged whole; larger
 * ones run as a pipeline over runs of whole utterances: slice i + 1 copies in while slice i computes and slice i - 1's features copy
 * back (three streams, a ring of three slots kept on the ctx).  Pinned host memory (hipHostMalloc / torch pin_memory) makes the copies
 * asynchronous and full rate; pageable memory works at the runtime's staging rate.  kernel_ms then spans the pipeline on the ctx stream.
 *
 * ssp_mfcc_run_i16: the same pass on int16 PCM — what utils.tools.read (utils/tools.py:45-47, scipy.io.wavfile) returns and the
 * reference's extractors receive (GMM_UBM.py:86-93, d_vector.py:80-98).  Samples are taken at their integer value (no 1/32768: sidekit's
 * mfcc computes on the integers as they are); half the bytes cross PCIe and a widening kernel on the device feeds the same MFCC
 * kernels (bit-identical to ssp_mfcc_run on the float32 of the same integers).  where = SSP_DEVICE: int16 device array, widened
 * slice by slice through the same ring. */
int ssp_mfcc_run_i16(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg,
                     const int16_t* samples, float* feats_out, int where, int variant, float* kernel_ms);

/* ---- stand-alone framing and cepstrum steps of the in-repo dialect (kept for API parity; ssp_mfcc_run fuses them) ---- */
/* utils.processing.enframe (utils/processing.py:19-38): frame i = x[i*step : i*step+frame_size], zero padded tail, times
 * window; n_frames = ceil(n/step).  frames_out: float[frame_size x n_frames] row-major, exactly the reference's
 * ndarray (element (k, i) = windowed sample k of frame i). */
int ssp_enframe(ssp_ctx* ctx, const float* samples, int64_t n, int32_t frame_size, int32_t step, const float* window /* HOST float[frame_size] */,
                float* frames_out, int where, float* kernel_ms);
/* utils.processing.stMFCC (utils/processing.py:91-107) on a batch of spectra: out = DCT(log(X . fbank^T (+eps | floor)))
 * X: float[n_rows x n_bins]; fbank: HOST float[n_filt x n_bins]; dct: HOST float[n_ceps x n_filt]; out: float[n_rows x n_ceps].
 * log_mode / floor_mode / eps as in ssp_mfcc_cfg. */
int ssp_cepstrum(ssp_ctx* ctx, const float* X, int64_t n_rows, int32_t n_bins, const float* fbank, int32_t n_filt,
                 const float* dct, int32_t n_ceps, int32_t log_mode, int32_t floor_mode, float eps, float* out, int where,
                 float* kernel_ms);

/* magnitude (power = 1) or power (2) spectrum, scaled, from rows of [re(0..n_bins) | im(0..n_bins)] — the |FFT|/L of
 * utils/processing.py:137-139 when the transform itself ran as a DFT-matrix product (ssp_dense_forward) for frame sizes
 * that are not powers of two.  reim: float[n_rows x 2 n_bins]; out: float[n_rows x n_bins]. */
int ssp_spectrum_abs(ssp_ctx* ctx, const float* reim, int64_t n_rows, int32_t n_bins, float scale, int32_t power, float* out,
                     int where, float* kernel_ms);

/* ---- stand-alone delta / CMVN on feature matrices (GMM_UBM.delta, preprocessing.scale) - */
int ssp_delta(ssp_ctx* ctx, const float* feats, const ssp_segments* frame_seg, int32_t dim, int32_t N,
              float* out, int where, float* kernel_ms);
int ssp_cmvn(ssp_ctx* ctx, const float* feats, const ssp_segments* frame_seg, int32_t dim,
             float* out, int where, float* kernel_ms);

/* ---- PLP back end: replaces sidekit.frontend.features.plp after its power spectrum -> Bark bands -> ln stage (call sites
 * GMM_UBM.py:95, d_vector.py:93, UI/GMM_UBM_GUI.py:93, UI/tmp.py:315-318; that front stage is ssp_mfcc_run with a Bark table and
 * an identity DCT).  logspec: float[F x n_bands] ln critical-band energies laid out by frame_seg (which must start at frame 0);
 * rasta != 0: RASTA filtering along time per utterance (first four frames of an utterance come out as the flat-spectrum
 * cepstrum, as in rastamat); then equal-loudness (band centres 0..fmax_hz in Bark), ^0.33, autocorrelation, Levinson-Durbin of
 * order plp_order - 1, LPC -> cepstrum, lifter n^lift.  ceps_out: float[F x plp_order] (c0 first).  sidekit's source is absent
 * from the reference tree: the arithmetic follows the published rastamat algorithm it ports (parity unpinned). */
int ssp_plp_post(ssp_ctx* ctx, const float* logspec, const ssp_segments* frame_seg, int32_t n_bands, float fmax_hz,
                 int32_t plp_order, int32_t rasta, float lift, float* ceps_out, int where, float* kernel_ms);

/* ---- GMM-UBM scoring: replaces the GMM[i].score(x_j) - UBM.score(x_j) double loop
 *      (GMM_UBM.py:181-197) and sklearn GaussianMixture.score_samples/score for diag models ---- */
/* weights: HOST double[n_models x K]; means, covars: HOST double[n_models x K x D].
 * has_ubm != 0: model 0 is the UBM; argmax/score differences are taken over models 1.. */
int ssp_gmm_pack(ssp_ctx* ctx, int32_t n_models, int32_t K, int32_t D, const double* weights,
                 const double* means, const double* covars, int32_t has_ubm, ssp_gmm** out);
int ssp_gmm_destroy(ssp_gmm* gmm);
/* feats: float[total frames x D]; loglik_out (nullable): float[n_models x total frames] (model-major,
 * = score_samples per model); scores_out (nullable): float[n_utt x n_models] mean log-likelihood
 * (= GaussianMixture.score); argmax_out (nullable): int32[n_utt] = argmax_i(score_i - score_ubm)
 * over speaker models (index 0 = first speaker model); precision: 0 fp32 MFMA (parity path) |
 * 1 bf16x3 split MFMA (fast path, same tolerance class) with every utterance whose top-2 margin lies inside the split-precision
 * error band scored again on the fp32 path, so the arg-max equals precision 0's | 2 bf16x3 split MFMA alone | 3 as 1 with the
 * calibrated band 8e-5 (|UBM score| + 1) (eight times the error measured at K = 64 / 512, D = 39): a HEURISTIC, about 100 times narrower
 * than 1's bound (rounding errors do not conspire and they average over an utterance's frames), so far fewer utterances are scored twice.
 * (precision 1: the band is a BOUND, not a calibration: with every operand split hi + lo the exponent of a mixture is off by at most
 *  eps * S(x), S(x) = sum_d |x_d| max|mu P|_d + x_d^2 max(P/2)_d (maxima over every mixture of every model, taken at ssp_gmm_pack),
 *  eps = 3.01 * 2^-18 + 8 D 2^-23 (the products a two-term split leaves out + worst-case fp32 accumulation on both paths); the
 *  log-sum-exp is 1-Lipschitz and the mean a mean, so a margin between two models is resolved when it exceeds
 *  2 (eps mean_t S(x_t) + 2^-20 (max_m |score_m| + 1)): gmm.hip gmm_band_kernel.  The call reads the flag count on the host, so it
 *  synchronises the stream even with device pointers and cannot be captured in a graph.  precision 0 is the parity path.
 *  What comes back for a close call (precision 1 / 3): the utterance's CANDIDATE models — those within the band of its best score, and
 *  the UBM — are scored again in fp32 and only THEIR entries of its scores_out row are replaced; the row's other entries keep their
 *  bf16x3 values (within the band of the fp32 path's).  The arg-max is the fp32 path's in every case.  A re-scoring pass too large for
 *  one launch scores every model and replaces the whole row.)
 * Without loglik_out the per-utterance means are formed inside the scoring kernel (the [n_models x frames] matrix never exists). */
int ssp_gmm_score(ssp_gmm* gmm, const float* feats, const ssp_segments* frame_seg, float* loglik_out,
                  float* scores_out, int32_t* argmax_out, int where, int precision, float* kernel_ms);
/* utterances the last precision = 1 call scored again on the fp32 path (diagnostics) */
int ssp_gmm_last_rescored(const ssp_gmm* gmm, int32_t* n_out);
/* precision = 4 (auto; GMM_UBM.py:183-187's arg-max with the fp32 path's result on every utterance, never dearer than the cheaper of the
 * two ways to get it): precision 1's guarantee scores close calls twice, which costs more than precision 0 once most utterances are
 * close calls.  A pilot — split-precision pass, band and candidate lists on the first ~2 % of the utterances (>= 256) — prices the
 * re-scoring (listed frames x candidate models); the call then runs as precision 1 when that predicts less than the fp32 pass, else as
 * precision 0.  Batches of fewer than ~16 machine-filling rounds of frames (3.1 M at 256 CUs) skip the pilot — it would cost a round of
 * its own — and decide LATE: the split pass runs on everything and, when re-scoring the close calls it lists would cost more than a
 * whole fp32 pass, that pass runs instead (at worst 1.33 x the fp32 path).  Batches under 1024 utterances and calls that ask for
 * loglik_out run as precision 0.  One extra host wait.
 * ssp_gmm_last_auto: what the last such call chose (precision_used; -1: none yet) and saw (predicted_cost: of precision 1, in units of
 * the fp32 pass). */
int ssp_gmm_last_auto(const ssp_gmm* gmm, int32_t* precision_used, int32_t* pilot_utts, int32_t* pilot_listed, float* predicted_cost);

/* ---- GMM training (EM): the O(frames x K x D) part of one iteration of sklearn GaussianMixture(covariance_type='diag').fit as
 *      the reference trains its speaker models and UBM (GMM_UBM.py:158-170; sklearn mixture/_base.py:_e_step,
 *      mixture/_gaussian_mixture.py:_estimate_gaussian_parameters) ---- */
/* Current parameters: HOST double weights[K], means[K x D], covars[K x D].  feats: float[n_frames x D].
 * Outputs (HOST double): nk_out[K] = sum_t resp[t,k];  sx_out[K x D] = sum_t resp[t,k] x[t,d];  sxx_out[K x D] = sum_t resp[t,k] x[t,d]^2;
 * loglik_sum_out = sum_t logsumexp_k(log w_k + log N(x_t | k))  (n_frames x sklearn's lower bound of the E step).
 * The O(K x D) closing arithmetic of the M step and the convergence test stay with the caller (float64). */
int ssp_gmm_em_stats(ssp_ctx* ctx, int32_t K, int32_t D, const double* weights, const double* means, const double* covars,
                     const float* feats, int64_t n_frames, double* nk_out, double* sx_out, double* sxx_out,
                     double* loglik_sum_out, int where, float* kernel_ms);

/* ---- DTW template matching: replaces the distance_dtw double loop of MFCC_DTW.py:57-108,187-217
 *      (dtw.accelerated_dtw(x, y, dist='euclidean'), warp 1) for every (query, template) pair ---- */
/* xq: float[total query rows x dim] with q_seg row offsets; xt, t_seg likewise for the templates (dim = 1: the reference's
 * flattened _MFCC sequences).  dist_out: float[n_q x n_t] = D1[r-1][c-1]; normalize != 0 divides by (r + c) (dtw <= 1.3.3). */
int ssp_dtw_distances(ssp_ctx* ctx, const float* xq, const ssp_segments* q_seg, const float* xt, const ssp_segments* t_seg,
                      int32_t dim, int32_t normalize, float* dist_out, int where, float* kernel_ms);

/* The dtw_method = 2 branch of the same matcher (MFCC_DTW.py:69-70: fastdtw(x, y, dist=euclidean), radius 1): the FastDTW approximation
 * for every (query, template) pair of 1-D sequences, float64 like the package.  Host arrays in, dist_out: HOST double[n_q x n_t]. */
int ssp_fastdtw_distances(ssp_ctx* ctx, const float* xq, const ssp_segments* q_seg, const float* xt, const ssp_segments* t_seg,
                          int32_t radius, double* dist_out, float* kernel_ms);

/* One pair WITH the warping path — what generate_template (MFCC_DTW.py:187-217) takes from accelerated_dtw: d and
 * path = _traceback(D0) (first minimum of diagonal / up / left at every step), computed in float64 like the package.
 * x: HOST float[r x dim], y: HOST float[c x dim]; path_i_out / path_j_out: HOST int32[r + c] (path_len_out entries are written). */
int ssp_dtw_path(ssp_ctx* ctx, const float* x, int64_t r, const float* y, int64_t c, int32_t dim, double* dist_out,
                 int32_t* path_i_out, int32_t* path_j_out, int32_t* path_len_out);

/* ---- d-vector network forward: one Dense layer Y = act(X W + b) of the speaker network the reference runs with
 *      spkModel.predict (d_vector.py:171-189 builds Dense(256) x 4 with ReLU between; predict at d_vector.py:298-299,327,348) ---- */
/* X: float[N x d_in]; Wt: float[units x d_in] = the Keras kernel (d_in x units) TRANSPOSED; bias: float[units] (nullable);
 * relu != 0 applies max(0, .); Y: float[N x units].  All four arrays live on the side `where` names. */
int ssp_dense_forward(ssp_ctx* ctx, const float* X, int64_t N, int32_t d_in, const float* Wt, const float* bias,
                      int32_t units, int32_t relu, float* Y, int where, float* kernel_ms);

/* The whole network as one object: n_layers Dense layers, layer l = (Wt[l]: HOST float[dims[l+1] x dims[l]] = Keras kernel transposed,
 * bias[l]: HOST float[dims[l+1]] or NULL, relu[l]).  ssp_dnn_forward = spkModel.predict: X float[N x dims[0]] -> Y float[N x dims[n_layers]].
 * Layers whose input and output widths are <= 256 (the reference's three hidden-to-hidden / output layers) run inside one kernel with
 * the activations kept in registers between layers; wider layers in front of them (the 1274-input layer) run one GEMM launch each. */
int ssp_dnn_create(ssp_ctx* ctx, int32_t n_layers, const int32_t* dims, const float* const* Wt, const float* const* bias,
                   const int32_t* relu, ssp_dnn** out);
int ssp_dnn_destroy(ssp_dnn* dnn);
int ssp_dnn_forward(ssp_dnn* dnn, const float* X, int64_t N, float* Y, int where, float* kernel_ms);

/* ---- d-vector cosine scoring: replaces the scipy cosine double loop + argmin
 *      (d_vector.py:315-319, 346-361) ---- */
/* X: float[N x d]; C: float[S x d]; dist_out (nullable): float[N x S] = clip(1 - cos, 0, 2);
 * argmin_out (nullable): int32[N] (first index on ties); min_out (nullable): float[N]. */
/* per-speaker centroids avg[s] = mean(X[labels == s]) with a float64 accumulator in row order (d_vector.py:310-313,
 * and nn_model.enroll d_vector.py:331 with one label).  X: float[N x d]; labels: int32[N] in [0, S); out: float[S x d]
 * (a speaker without rows gives NaN like numpy's mean of an empty slice). */
int ssp_centroids(ssp_ctx* ctx, const float* X, const int32_t* labels, int64_t N, int32_t d, int32_t S, float* out,
                  int where, float* kernel_ms);

int ssp_cosine_identify(ssp_ctx* ctx, const float* X, int64_t N, int32_t d, const float* C, int32_t S,
                        float* dist_out, int32_t* argmin_out, float* min_out, int where, float* kernel_ms);
/* the same with a choice of arithmetic.  precision 0: fp32-input MFMA (the parity path; ssp_cosine_identify).  precision 1: the sweep on
 * bf16 MFMA with every operand split hi + lo (three products per k-step, fp32 accumulation) keeping each embedding's two largest
 * cosines; rows whose two best are closer than twice a PROVEN bound on |cos(bf16x3) - cos(fp32 path)| for unit vectors
 * (3.01 * 2^-18 + 4 d 2^-23 + (d + 8) 2^-24, the last term for the two paths' different normalisation roundings: cosine.hip cos_band),
 * and every row that meets a NaN or a zero norm, are scored again by the fp32 kernel from a device-side list (no host round trip in
 * between; with device pointers the call is asynchronous on the ctx stream — its scratch lives on the ctx — and the diagnostics
 * below fetch their counts when asked).  argmin_out is therefore the fp32 path's on EVERY row; min_out is within the bound
 * of it (exact on the re-scored rows).  precision 2: a cascade — a sweep on the hi parts alone (one product per k-step, bound
 * 2.01 * 2^-9 + 2 d 2^-23 + (d + 8) 2^-24 = 4e-3) first, its close calls to the bf16x3 sweep, that one's to fp32: the same arg-min guarantee, three times
 * fewer matrix instructions on well-separated data (how many rows each later stage takes depends on the data); min_out is then only within
 * 4e-3 of the fp32 path's on rows the first sweep decided.  Arg-min / minimum only: dist_out must be NULL; d <= 256. */
int ssp_cosine_identify2(ssp_ctx* ctx, const float* X, int64_t N, int32_t d, const float* C, int32_t S, float* dist_out,
                         int32_t* argmin_out, float* min_out, int where, int precision, float* kernel_ms);
/* diagnostics: rows the last precision >= 1 call scored again in fp32; rows its precision-2 cascade handed to the bf16x3 sweep
 * (these wait for the ctx stream when the counts of a device-pointer call have not been read yet) */
int ssp_cosine_last_rescored(const ssp_ctx* ctx, int32_t* n_out);
int ssp_cosine_last_split_rows(const ssp_ctx* ctx, int32_t* n_out);
/* precision = 3 (auto; d_vector.py:315-319's arg-min with the fp32 path's result on every row, at the cost of the cheapest path): the
 * pilot is the first round of the cascade's bf16 sweep (one machine-filling set of waves, at most N / 8 rows): it lists its close calls
 * as the full sweep would and counts the rows closer than the bf16x3 band beside them; from those two shares the call goes on as the
 * cascade (2: the sweep continues behind the pilot's rows, nothing is computed twice), or starts over as the bf16x3 sweep (1) or — when
 * nearly every row is a close call — the fp32 sweep (0).  N < 8192, d > 256 and dist_out requests run as precision 0 without a pilot.
 * One host wait per call (the counts), also with device pointers.  ssp_cosine_last_auto: what the last such call chose and saw
 * (precision_used -1: no auto call yet). */
int ssp_cosine_last_auto(const ssp_ctx* ctx, int32_t* precision_used, int32_t* pilot_rows, int32_t* pilot_to_bf16x3, int32_t* pilot_to_fp32);

#ifdef __cplusplus
}
#endif
#endif /* SSP_H_ */

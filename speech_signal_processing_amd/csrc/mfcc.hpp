// Kernel argument block + plan object of the fused MFCC pass.
#pragma once
#include "common.hpp"

namespace ssp {

struct MfccChunk {
    int32_t utt;  // utterance index
    int32_t t0;   // first frame of the chunk inside the utterance
    int32_t n;    // frames in the chunk
    int32_t pad;  // wave-stream kernel: extra frames of halo in front of the chunk (0, or 12: the chunk then reproduces the uncut utterance's bits)
};

struct MfccArgs {
    // data
    const float* samples;
    const int64_t* sample_off;  // [n_utt+1] device
    const int64_t* frame_off;   // [n_utt+1] device
    float* out;                 // [total_frames x d_out]
    float* lm_out;              // non-null: the generic kernel only writes log-mel rows [total_frames x n_filt] here (first pass of
                                // the two-pass top_db path for utterances longer than one workgroup's LDS)
    const MfccChunk* chunks;
    float* utt_max;             // [n_utt] two-pass top_db: largest log-mel value of each utterance (-inf before the first pass)
    // tables (device)
    const float* window;    // [n_fft] zero padded
    const float2* twiddle;  // [n_fft]  W_nfft^k = exp(-2 pi i k / n_fft)
    const int32_t* filt_lo4;   // [n_filt] first non-zero bin of each filter, rounded down to a multiple of 4
    const float* filt_wT;      // per group of 64 filters [gsteps][filters in the group][4]: taps 4 s .. 4 s + 3 (from filt_lo4), zero padded
    const int32_t* filt_grp;   // [2][8] per group: 16-byte steps of its widest filter (even); offset of its block in filt_wT (16-byte units)
    int32_t filt_w4_total;     // 16-byte entries in filt_wT
    int32_t dct_identity;      // n_ceps == n_filt and the DCT matrix is the identity: the log filterbank row is the output
    int32_t dct_ncp;           // DCT lanes per coefficient block: power of two >= min(n_ceps, 64)
    const float* dct;          // [n_ceps x n_filt]
    const float* dctT;         // [n_filt x n_ceps]
    // cfg
    int32_t win_len, hop, n_fft, n_filt, n_ceps, d_out;
    int32_t frame_mode, preemph_mode, spec_power, log_mode, floor_mode, delta_order, delta_N, cmvn;
    float preemph, spec_scale, eps, top_db, delta_inv_denom;
    // LDS carve (bytes from the dynamic LDS base; all multiples of 16)
    int32_t lds_logmel_off, lds_ceps_off, lds_dlt_off, lds_ddl_off, lds_lmrows_off, lds_stats_off, lds_tw_off;
    int32_t lds_dct_off;  // transposed DCT matrix staged in LDS (-1: read from global memory)
    int32_t lds_wt_off;   // filter taps staged in LDS (-1: read from global memory)
    int32_t lm_stride;    // floats per log-mel row kept in LDS (4 x odd: 16-byte reads down a column are conflict free)
    int32_t n_chunks;
    int32_t prefetch;     // touch the next frame's cache lines ahead of its staging loads
};

constexpr int MFCC_FAST_MAX_PASS = 4;  // <= 64 filters in the fused n_fft == 512 kernel

// extra tables / LDS carve of the fused n_fft == 512 kernel (mfcc_fast.hip)
struct FastArgs {
    const float2* tw16;     // [16][16]  W_256^(k1*n2)
    const float2* wpost;    // [9][16]   W_512^k, k = p + 16 i (i < 8); row 8 unused
    const float* melw;      // [total_steps][16]
    const int32_t* mel_lo;  // [n_pass*16]
    const int32_t* mel_id;  // [n_pass*16] filter id, -1 = empty slot
    const float* dctT;      // [n_filt][q_pass*16]
    int32_t mel_steps[MFCC_FAST_MAX_PASS];
    int32_t mel_blocks[MFCC_FAST_MAX_PASS];      // 4-step blocks per pass
    int32_t lm_pad;                               // zero-padded log-mel entries behind the n_filt real ones
    int32_t n_pass, q_pass, total_steps, n_filt4;  // mel steps are 4-tap (16-byte) steps
    float ddw[17];          // delta-delta of interior frames as one convolution over 4N+1 cepstra
    int32_t slen;           // staged samples per quad = 3*hop + 32*NZ
    int32_t stage_floats;   // per-wave stage buffer (>= slen, >= 4*PSTR)
    int32_t ceps_rows;      // capacity of the cepstra buffer (rows)
    int32_t off_win, off_tw16, off_wpost, off_melw, off_mello, off_melid, off_dct, off_ceps, off_stats, off_wave, wave_bytes;
    float pscale;           // 0.25 * spec_scale (power) or 0.5 * spec_scale (magnitude)
    float one_minus_a;
    // register-resident "piece" filterbank (mfcc_fast.hip step 7, MELV > 0): every lane owns up to 4*MELV consecutive taps
    // of ONE filter; a filter's pieces sit in consecutive lanes of one 16-lane row and are summed by a masked DPP scan
    const float* pc_w;        // [64][melv][4] weights in the lane's read order
    const int32_t* pc_ofs;    // [64][melv]    byte offset of each 16-byte read inside a frame's P row
    const float* pc_mask;     // [64][4]       scan masks: 1 when lane + (1 << s) holds a piece of the same filter
    const int32_t* pc_fid;    // [64]          filter id on the first lane of a filter's run, else -1
    int32_t melv;             // 16-byte reads per lane and frame (0: the banded sweep is used instead)
    int32_t mel_ns;           // scan steps = ceil(log2(longest run))
    float log_add, log_max, log_k;
    // persistent workgroups
    float* ceps_scratch;      // [grid][ceps_stride] cepstra of the chunk each workgroup has in flight
    int32_t* work_counter;    // [0] next chunk to claim (zeroed before every launch)
    int32_t* redo_flags;      // (unused by the workgroup kernel — it forms non-finite steps inline —; kept so that the argument block's
                              //  layout, which the stream kernels share, stays what their instances were tuned with)
    int32_t ceps_stride;      // floats per workgroup slot (multiple of 4)
    int32_t n_chunks;  // log(max(v + log_add, log_max)) * log_k  (floor_mode / log_mode, branch free)
};

// wave-stream kernel (mfcc_stream.hip): every wave walks its own chunk; DCT / delta / delta-delta on the matrix cores
struct StreamArgs {
    const float* dctA;        // [KS][64] DCT matrix as the MFMA A operand: lane (ceps = l & 15, kq = l >> 4), k-step s <-> filter KS kq + s
    int32_t* work_counter;    // [0] next chunk to claim | [1] != 0: a chunk was flagged (64 bytes, zeroed before every launch)
    int32_t* redo_flags;      // = work_counter + 16: [n_chunks], zeroed by the launch, set by mfcc_stream_scan_kernel (the SECOND kernel) for every
                              // chunk whose stored rows show a non-finite cepstrum in a time step's window; the THIRD kernel (WALK = 1) walks those again
    int32_t n_chunks;
    int32_t wave_bytes;       // LDS per wave: 4 frame images + sample stage + cepstrum ring
    int32_t stage_bytes;      // sample stage (whole 1-KiB DMA pieces + a trailing 512-B half piece)
    int32_t table_bytes;      // workgroup-shared DCT operand table in front of the wave regions
    int32_t tstep;            // the instance has the transposed step form (scan kernel: without deltas one row per step tells)
    const float* dense_w;     // dense-band instances: [lane][6 bands][20] weights (16 bins of the lane's chunk, bin 256, pad), pscale folded in
};

int launch_mfcc_generic(const MfccArgs& args, int n_chunks, size_t lds_bytes, int n_waves, int num_cu, hipStream_t stream);
// per-utterance CMVN over a feature matrix in global memory (feat_ops.hip; in == out allowed), any utterance length
int launch_cmvn(const float* in, float* out, const int64_t* frame_off_dev, int64_t n_utt, int dim, int64_t max_T, hipStream_t stream);
// second pass of the two-pass top_db path: per-utterance max of the log-mel rows, clamp at max - top_db, DCT rows -> out [F x n_ceps]
int launch_topdb_dct(const float* logmel, const int64_t* frame_off_dev, int64_t n_utt, const MfccChunk* chunks, int n_chunks,
                     const float* utt_max, int n_filt, int n_ceps, const float* dct, float top_db, float* out, hipStream_t stream);
    

}  // namespace ssp

struct ssp_mfcc_plan {
    ssp_ctx* ctx = nullptr;
    ssp_mfcc_cfg cfg{};
    int32_t d_out = 0;
    bool reproducible = false;  // SSP_MFCC_REPRODUCIBLE: chunking / scaling choices that depend on the batch or the machine are pinned
    ssp::DevBuf window, twiddle, filt_lo4, filt_grp, filt_wT, dct, dctT, fbank_dense;
    int32_t max_filt_len = 0;
    // cached work table for the last (sample_seg, frame_seg, variant) seen
    uint64_t checked_sseg = 0, checked_fseg = 0;  // last (sample, frame) segment pair validated against the framing rule
    uint64_t cache_sseg = 0;  // ssp_segments::serial
    uint64_t cache_fseg = 0;
    int cache_variant = -1;   // kernel the cached work table was laid out for (1 generic | 2 workgroup-fused | 3 wave-stream)
    int cache_request = -1;   // variant argument that table answers (0 = auto: cache_variant is what auto resolved to)
    bool cache_split_cmvn = false;  // CMVN as a second kernel (an utterance exceeds one workgroup's chunk)
    bool cache_split_topdb = false; // top_db as a second kernel (log-mel rows through a global scratch)
    ssp::DevBuf lm_scratch, umax_scratch;
    int cache_chunk_frames = 0;
    size_t cache_lds = 0;
    int cache_waves = 4;  // waves per workgroup of the generic kernel
    int32_t cache_n_chunks = 0;
    std::vector<int32_t> cache_chunk_first;  // [n_utt + 1] first chunk of every utterance (chunks are laid out in utterance order): a
                                             // range of utterances is a range of chunks (the sliced host-fed path launches such ranges)
    ssp::DevBuf chunks;
    ssp::MfccArgs args{};
    // fused n_fft == 512 kernel
    bool fast_ready = false;
    int64_t fast_max_samples = 0;  // longest utterance of the cached work table (32-bit offsets in the fast kernel)
    ssp::FastArgs fast{};
    ssp::DevBuf f_tw16, f_wpost, f_melw, f_mello, f_melid, f_dct, f_pcw, f_pcofs, f_pcmask, f_pcfid, f_scratch, f_counter;
    // wave-stream kernel
    bool stream_ready = false;
    ssp::DevBuf s_dctA, s_dense;
    bool cache_s2k_fused = false;  // (with the cached work table of a variant-4 run)
    bool s2k_ready = false;   // tables of the 2048-point wave-stream kernel (mfcc_stream2k.hip)
    int32_t s2k_steps = 0, s2k_steps1 = 0;
    ssp::DevBuf s2k_twA, s2k_twB, s2k_twS, s2k_mel, s2k_minfo;
};

namespace ssp {
bool mfcc_fast_supported(const ssp_mfcc_cfg& cfg);
int build_fast_tables(ssp_mfcc_plan* plan);
size_t mfcc_fast_lds(const ssp_mfcc_cfg& cfg, FastArgs& f, int chunk_frames);
int mfcc_fast_max_chunk(const ssp_mfcc_cfg& cfg, const FastArgs& f);  // most frames one workgroup can take at once
int launch_mfcc_fast(const MfccArgs& args, ssp_mfcc_plan* plan, int n_chunks, int chunk_frames, hipStream_t stream);
bool mfcc_stream_supported(const ssp_mfcc_plan* plan);  // cfg covered by the wave-stream kernel
bool mfcc_s2k_supported(const ssp_mfcc_plan* plan);    // n_fft == 2048 plans without deltas: first pass on the 2048-point wave-stream kernel
int build_s2k_tables(ssp_mfcc_plan* plan);
bool mfcc_s2k_fuses(const ssp_mfcc_plan* plan, int64_t max_T, int chunk_frames);  // the first pass also clamps and takes the DCT (single-chunk utterances)
int launch_mfcc_s2k(const MfccArgs& args, ssp_mfcc_plan* plan, int n_chunks, hipStream_t stream);
bool mfcc_stream_dense(const ssp_mfcc_plan* plan);      // ... by the dense-band instance (identity DCT over <= 24 dense filterbank rows: the PLP front end)
bool mfcc_stream_fuses_cmvn(const ssp_mfcc_plan* plan); // ... by an instance that scales the features itself (cmvn) when every utterance is one chunk
int build_stream_tables(ssp_mfcc_plan* plan);
int launch_mfcc_stream(const MfccArgs& args, ssp_mfcc_plan* plan, int n_chunks, hipStream_t stream, bool dry_run = false);
}  // namespace ssp

// Host side of the MFCC pass: plan (device tables), framing rules, chunk work table, launch.
#include <cmath>
#include <cstdlib>

#include "mfcc.hpp"

namespace ssp {

static int validate_cfg(const ssp_mfcc_cfg* c) {
    if (!c) SSP_FAIL(SSP_ERR_INVALID, "mfcc: null cfg");
    if (c->n_fft < 64 || c->n_fft > 2048 || (c->n_fft & (c->n_fft - 1)))
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: n_fft=%d must be a power of two in [64, 2048]", c->n_fft);
    if (c->win_len < 1 || c->win_len > c->n_fft) SSP_FAIL(SSP_ERR_INVALID, "mfcc: win_len=%d not in [1, n_fft]", c->win_len);
    if (c->hop < 1) SSP_FAIL(SSP_ERR_INVALID, "mfcc: hop < 1");
    if (c->n_filt < 1 || c->n_filt > 512) SSP_FAIL(SSP_ERR_INVALID, "mfcc: n_filt=%d not in [1,512]", c->n_filt);
    if (c->n_ceps < 1 || c->n_ceps > 128) SSP_FAIL(SSP_ERR_INVALID, "mfcc: n_ceps=%d not in [1,128]", c->n_ceps);
    if (c->frame_mode < 0 || c->frame_mode > 2) SSP_FAIL(SSP_ERR_INVALID, "mfcc: frame_mode");
    if (c->frame_mode == 2 && c->win_len != c->n_fft)
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: centred framing needs win_len == n_fft");
    if (c->preemph_mode < 0 || c->preemph_mode > 1) SSP_FAIL(SSP_ERR_INVALID, "mfcc: preemph_mode");
    if (c->spec_power != 1 && c->spec_power != 2) SSP_FAIL(SSP_ERR_INVALID, "mfcc: spec_power must be 1 or 2");
    if (c->log_mode < 0 || c->log_mode > 2) SSP_FAIL(SSP_ERR_INVALID, "mfcc: log_mode");
    if (c->floor_mode < 0 || c->floor_mode > 2) SSP_FAIL(SSP_ERR_INVALID, "mfcc: floor_mode");
    if (c->delta_order < 0 || c->delta_order > 2) SSP_FAIL(SSP_ERR_INVALID, "mfcc: delta_order must be 0, 1 or 2");
    if (c->delta_order > 0 && (c->delta_N < 1 || c->delta_N > 8))
        SSP_FAIL(SSP_ERR_INVALID, "mfcc: delta_N must be in [1, 8] (GMM_UBM.py:59 raises for N < 1)");
    return SSP_OK;
}

static int64_t frames_for(const ssp_mfcc_cfg& c, int64_t n) {
    switch (c.frame_mode) {
        case 0: return n < c.win_len ? 0 : (n - c.win_len) / c.hop + 1;  // sidekit framing: floor, no padding
        case 1: return (n + c.hop - 1) / c.hop;                          // utils/processing.py:27 ceil(wlen/step)
        default: return n <= 0 ? 0 : 1 + n / c.hop;                      // librosa stft(center=True)
    }
}

template <class T>
static int upload(DevBuf& b, const std::vector<T>& v, hipStream_t s) {
    SSP_TRY(b.alloc(sizeof(T) * v.size()));
    if (!v.empty()) SSP_HIP(hipMemcpyAsync(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s));
    return SSP_OK;
}

static size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

// LDS carve of the generic kernel for chunks of `ch` frames and `nw` waves per workgroup; returns total bytes
static size_t generic_lds_layout(const ssp_mfcc_cfg& c, int ch, int nw, MfccArgs* a) {
    const int H = c.delta_order * c.delta_N;
    const int M = c.n_fft / 2;
    size_t off = (size_t)nw * (M + M / 16) * sizeof(float2);  // per wave: M complex points, one pad slot per 16
    a->lds_logmel_off = (int32_t)off;
    off = align16(off + (size_t)nw * c.n_filt * sizeof(float));
    a->lds_ceps_off = (int32_t)off;
    off = align16(off + (size_t)(ch + 2 * H) * c.n_ceps * sizeof(float));
    a->lds_dlt_off = (int32_t)off;
    if (c.delta_order >= 1) off = align16(off + (size_t)(ch + 2 * (c.delta_order - 1) * c.delta_N) * c.n_ceps * sizeof(float));
    a->lds_ddl_off = (int32_t)off;
    if (c.delta_order >= 2) off = align16(off + (size_t)ch * c.n_ceps * sizeof(float));
    a->lds_lmrows_off = (int32_t)off;
    a->lm_stride = 4 * (((c.n_filt + 3) / 4) | 1);
    if (c.top_db >= 0.f) off = align16(off + (size_t)ch * a->lm_stride * sizeof(float));
    a->lds_stats_off = (int32_t)off;
    off = align16(off + (size_t)(2 * c.n_ceps * (1 + c.delta_order) + 16) * sizeof(float));
    a->lds_tw_off = (int32_t)off;
    off = align16(off + (size_t)c.n_fft * sizeof(float2));
    a->lds_dct_off = a->lds_wt_off = -1;
    const size_t wt_bytes = (size_t)a->filt_w4_total * 16;
    if (wt_bytes <= 32 * 1024) {
        a->lds_wt_off = (int32_t)off;
        off = align16(off + wt_bytes);
    }
    const size_t dct_bytes = (size_t)((c.n_filt + 3) & ~3) * c.n_ceps * sizeof(float);
    if (dct_bytes <= 16 * 1024) {
        a->lds_dct_off = (int32_t)off;
        off = align16(off + dct_bytes);
    }
    return off;
}

static int build_work(ssp_mfcc_plan* p, const ssp_segments* sseg, const ssp_segments* fseg, int variant) {
    const ssp_mfcc_cfg& c = p->cfg;
    bool whole = c.cmvn || c.top_db >= 0.f;  // needs utterance-level statistics inside one workgroup
    bool split_cmvn = false;                 // utterances too long for that: features un-normalised, then the CMVN kernel in place
    bool split_topdb = false;                // ... or log-mel rows through a global scratch, then the clamp + DCT kernel
    const int64_t max_T = fseg->max_len();
    const size_t lds_cap = 160 * 1024;
    int ch, nw = 4;
    size_t lds = 0;
    bool small_batch = false;
    const bool repro = p->reproducible;  // ssp_mfcc_plan_set_flags(SSP_MFCC_REPRODUCIBLE): an utterance's bits must not depend on the batch
    // a failed build must not leave a half-written layout behind a valid cache key (the next run of the cached segment pair would launch
    // with it): the key is dropped first and only restored at the end
    p->cache_sseg = p->cache_fseg = 0;
    p->cache_variant = p->cache_request = -1;
    if (variant == 4) {
        // 2048-point wave-stream kernel: first pass only (log filterbank rows + utterance maxima); chunks of 64 frames, one wave each
        // (small batches: shorter chunks, so that a single utterance is spread over many waves instead of walked by one; they always
        //  take the second-pass kernel for the clamp + DCT: the same bits for an utterance alone and in any small batch)
        // (the kernel forms 32-bit byte offsets from sample indices: longer utterances stay with the generic kernel)
        if (sseg->max_len() * 4 > (int64_t)INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream2048): utterance too long for 32-bit offsets");
        ch = 128;
        bool s2k_small = false;
        {
            const int64_t total = fseg->host.back() - fseg->host.front(), want = (int64_t)p->ctx->num_cu * 12;
            if ((total + ch - 1) / ch < want) {
                ch = (int)std::min<int64_t>(128, std::max<int64_t>(4, (total + want - 1) / want));
                s2k_small = true;
            }
        }
        split_topdb = true;  // (the rows always go through the global scratch; whether a second pass follows: cache_s2k_fused)
        whole = false;
        p->cache_s2k_fused = !s2k_small && !repro && mfcc_s2k_fuses(p, max_T, ch);
    } else if (variant == 3) {
        // wave-stream kernel: a chunk is a run of frames one WAVE walks alone (no LDS bound); whole utterances up to 512 frames, longer ones
        // in 512-frame chunks with a recomputed 4-frame halo.  CMVN needs the whole utterance: inside the kernel when every utterance is a
        // single chunk (and the dialect has a scaling instance), the stand-alone kernel afterwards otherwise.
        // (utterances may start at any sample: the 16-byte LDS-DMA loads of the sample stage only need dword-aligned addresses)
        ch = (int)std::min<int64_t>(std::max<int64_t>(max_T, 1), 512);
        {
            // small batches (the reference calls these functions one utterance at a time): a chunk per wave would leave most of the
            // machine idle, so utterances are cut into shorter chunks (multiples of 16 frames, >= 32; each recomputes a 4-frame halo)
            // until there are about 12 waves' worth per CU.  Chunks cut by this rule agree with each other bit for bit, with the uncut utterance to rounding (mfcc_stream.hip: H)
            const int64_t total = fseg->host.back() - fseg->host.front(), want = (int64_t)p->ctx->num_cu * 12;
            if ((total + ch - 1) / ch < want) {
                const int64_t per = (total + want - 1) / want;
                ch = (int)std::min<int64_t>(ch, std::max<int64_t>(32, (per + 15) / 16 * 16));
                small_batch = true;
            }
        }
        // (small batches always scale with the stand-alone kernel: an utterance then gets the same bits alone and in any small batch;
        //  reproducible plans in every batch)
        split_cmvn = c.cmvn != 0 && !(!small_batch && !repro && max_T <= ch && mfcc_stream_fuses_cmvn(p));
        whole = false;
        {
            // mfcc_stream_supported() is wider than the set of compiled instances (e.g. win <= 416 with hop > 160 needs the two-per-CU
            // LDS layout, which only some dialects have): ask the launcher itself; auto mode then falls back to the workgroup kernel
            MfccArgs probe = p->args;
            probe.cmvn = (c.cmvn != 0 && !split_cmvn) ? 1 : 0;
            const int prc = launch_mfcc_stream(probe, p, 0, nullptr, /*dry_run=*/true);
            if (prc != SSP_OK) return prc;
        }
    } else if (variant == 2) {
        // persistent workgroups with a fixed LDS footprint; a chunk's cepstra (+ delta halo) and a block of output rows
        // share the wave regions in the delta tail, which bounds the chunk: whole utterances up to 512 frames, else chunks
        FastArgs tmp = p->fast;
        const int cap = mfcc_fast_max_chunk(c, tmp);
        ch = (int)std::min<int64_t>(std::max<int64_t>(max_T, 1), std::min(512, cap));
        if (whole) {
            ch = (int)std::max<int64_t>(max_T, 1);
            if (ch > cap) {  // (top_db never reaches the fused kernel)
                split_cmvn = true;
                whole = false;
                ch = (int)std::min<int64_t>(std::max<int64_t>(max_T, 1), std::min(512, cap));
            }
        }
        lds = mfcc_fast_lds(c, tmp, ch);
        if (lds > lds_cap) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): LDS footprint %zu B exceeds 160 KiB", lds);
    } else {
        // chunked work: 4 waves per workgroup; the LDS cap sets how many workgroups share a CU (2 at n_fft 2048, whose registers allow
        // no more; 3 at 1024; 4 below — measured, tools/dialect_bench.py); whole-utterance work (CMVN / top_db inside the kernel):
        // 8 waves when the utterance's rows still fit the 160 KiB, else 4
        MfccArgs tmp = p->args;  // (the table sizes the layout depends on)
        size_t half_cap = (size_t)(c.n_fft >= 2048 ? 80 : (c.n_fft >= 1024 ? 53 : 40)) * 1024;
        if (const char* e = getenv("SSP_GENERIC_LDS_CAP_KB")) half_cap = (size_t)atoi(e) * 1024;
        auto chunk_for = [&](const ssp_mfcc_cfg& cc) {
            int k = (int)std::min<int64_t>(std::max<int64_t>(max_T, 1), 512);
            while (k > 16 && generic_lds_layout(cc, k, 4, &tmp) > half_cap) k = (k * 3) / 4;
            return k;
        };
        nw = 4;
        ch = chunk_for(c);
        if (whole) {
            ch = (int)std::max<int64_t>(max_T, 1);
            if (generic_lds_layout(c, ch, 8, &tmp) <= lds_cap) {
                nw = 8;
            } else if (generic_lds_layout(c, ch, 4, &tmp) > lds_cap) {
                if (c.top_db >= 0.f) {
                    if (c.delta_order != 0 || c.cmvn)
                        SSP_FAIL(SSP_ERR_UNSUPPORTED,
                                 "mfcc: top_db with deltas / cmvn needs a whole utterance per workgroup; %lld frames exceed the 160 KiB LDS",
                                 (long long)max_T);
                    split_topdb = true;
                } else {
                    split_cmvn = true;
                }
                whole = false;
                ch = chunk_for(c);
                if (split_topdb) {  // the chunked first pass keeps no per-chunk log-mel rows: lay the LDS out as if top_db were off
                    ssp_mfcc_cfg c2 = c;
                    c2.top_db = -1.f;
                    ch = chunk_for(c2);
                }
            }
        }
        {
            ssp_mfcc_cfg c2 = c;
            if (split_topdb) c2.top_db = -1.f;
            lds = generic_lds_layout(c2, ch, nw, &p->args);
        }
        if (lds > lds_cap) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: LDS footprint %zu B exceeds 160 KiB", lds);
    }
    std::vector<MfccChunk> chunks;
    chunks.reserve((size_t)fseg->n);
    // wave-stream kernel, machine-filling batches: the waves claim chunks from one counter, so the launch ends with a partial round in
    // which some waves walk one more whole utterance while the others idle (100 000 utterances over 3 072 waves: 32.55 rounds, 1.4 % of
    // the launch).  The utterances that are claimed last are cut in two (multiples of 16 frames), which
    // halves that quantum; the second half starts 16 frames early (pad = 12 on top of the 4-frame halo: mfcc_stream.hip) so that it
    // reproduces the uncut utterance bit for bit.  Not with the in-kernel scaling: it needs an utterance in one chunk.
    int64_t tail_from = fseg->n;
#ifndef SSP_NO_TAIL_SPLIT
    if (variant == 3 && !(c.cmvn != 0 && !split_cmvn) && !getenv("SSP_MFCC_NO_TAIL_SPLIT")) {
        const int64_t waves = (int64_t)p->ctx->num_cu * 12;
        if (fseg->n >= 4 * waves) tail_from = fseg->n - waves;  // (measured: halves beat thirds / quarters, one wave-set beats two)
    }
#endif
    p->cache_chunk_first.assign((size_t)fseg->n + 1, 0);
    for (int64_t u = 0; u < fseg->n; ++u) {
        const int64_t T = fseg->host[u + 1] - fseg->host[u];
        if (T > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: utterance %lld has too many frames", (long long)u);
        p->cache_chunk_first[(size_t)u] = (int32_t)std::min<size_t>(chunks.size(), (size_t)INT32_MAX);
        int64_t chu = ch;
        if (u >= tail_from && T >= 64) chu = std::min<int64_t>(ch, ((T + 1) / 2 + 15) / 16 * 16);
        // wave-stream kernel: a cut chunk starts 16 frames early (pad = 12 on top of the 4-frame halo) and then reproduces the uncut
        // utterance's bits — every cut of a machine-filling batch (the 512-frame cuts of long utterances and the tail split: an utterance's
        // bits do not depend on its place in the batch), and the short latency cuts of small batches as well when the plan is reproducible
        // (they are multiples of 16 frames; 12 more frames per 32-frame chunk is why that is not the default)
        const int pad = (variant == 3 && (!small_batch || repro)) ? 12 : 0;
        for (int64_t t0 = 0; t0 < T; t0 += chu)
            chunks.push_back(MfccChunk{(int32_t)u, (int32_t)t0, (int32_t)std::min<int64_t>(chu, T - t0), t0 > 0 ? pad : 0});
    }
    if (chunks.size() > (size_t)INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: too many chunks");
    p->cache_chunk_first[(size_t)fseg->n] = (int32_t)chunks.size();
    SSP_TRY(upload(p->chunks, chunks, p->ctx->stream));
    SSP_HIP(hipStreamSynchronize(p->ctx->stream));  // `chunks` (host) dies at return
    p->fast_max_samples = sseg->max_len();
    p->cache_n_chunks = (int32_t)chunks.size();
    p->cache_chunk_frames = ch;
    p->cache_lds = lds;
    p->cache_waves = nw;
    p->cache_sseg = sseg->serial;
    p->cache_fseg = fseg->serial;
    p->cache_variant = variant;
    p->cache_split_cmvn = split_cmvn;
    p->cache_split_topdb = split_topdb;
    return SSP_OK;
}

}  // namespace ssp

using namespace ssp;

extern "C" {

int ssp_mfcc_num_frames(const ssp_mfcc_cfg* cfg, int64_t n_samples, int64_t* n_frames) {
    if (!cfg || !n_frames || n_samples < 0) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_num_frames: bad argument");
    if (cfg->hop < 1 || cfg->frame_mode < 0 || cfg->frame_mode > 2) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_num_frames: bad cfg");
    *n_frames = frames_for(*cfg, n_samples);
    return SSP_OK;
}

int ssp_mfcc_out_dim(const ssp_mfcc_cfg* cfg, int32_t* d_out) {
    if (!cfg || !d_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_out_dim: null");
    *d_out = cfg->n_ceps * (1 + cfg->delta_order);
    return SSP_OK;
}

int ssp_mfcc_plan_create(ssp_ctx* ctx, const ssp_mfcc_cfg* cfg, const float* window, const float* fbank,
                         const float* dct, ssp_mfcc_plan** out) {
    ssp::TraceRange trace_("ssp_mfcc_plan_create");
    if (!out) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_plan_create: null out");
    *out = nullptr;
    SSP_TRY(use_ctx(ctx));
    SSP_TRY(validate_cfg(cfg));
    if (!window || !fbank || !dct) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_plan_create: null table");
    ssp_mfcc_plan* p = new (std::nothrow) ssp_mfcc_plan;
    if (!p) SSP_FAIL(SSP_ERR_NOMEM, "mfcc plan: host alloc");
    p->ctx = ctx;
    p->cfg = *cfg;
    p->d_out = cfg->n_ceps * (1 + cfg->delta_order);
    const int n_fft = cfg->n_fft, nb = n_fft / 2 + 1;

    std::vector<float> win(n_fft, 0.f);
    for (int i = 0; i < cfg->win_len; ++i) win[i] = window[i];
    std::vector<float2> tw(n_fft);
    for (int k = 0; k < n_fft; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)n_fft;
        tw[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    // banded form of the filterbank: per filter [lo, lo+len) = first..last non-zero bin
    std::vector<int32_t> lo(cfg->n_filt), len(cfg->n_filt), ofs(cfg->n_filt);
    std::vector<float> w;
    int32_t maxlen = 0;
    for (int j = 0; j < cfg->n_filt; ++j) {
        const float* row = fbank + (size_t)j * nb;
        int first = -1, last = -1;
        for (int k = 0; k < nb; ++k)
            if (row[k] != 0.f) {
                if (first < 0) first = k;
                last = k;
            }
        lo[j] = first < 0 ? 0 : first;
        len[j] = first < 0 ? 0 : last - first + 1;
        ofs[j] = (int32_t)w.size();
        for (int k = 0; k < len[j]; ++k) w.push_back(row[lo[j] + k]);
        maxlen = std::max(maxlen, len[j]);
    }
    p->max_filt_len = maxlen;
    // generic kernel: per group of 64 filters a transposed, zero-padded tap table read 4 taps at a time from the first bin
    // rounded down to a multiple of 4
    const int n_grp = (cfg->n_filt + 63) / 64;  // <= 8 (n_filt <= 512)
    std::vector<int32_t> lo4(cfg->n_filt);
    std::vector<float> wT;
    MfccArgs& ga = p->args;
    const int buf_floats = n_fft + n_fft / 16;  // the wave buffer the spectrum row sits in
    std::vector<int32_t> grp(16, 0);
    for (int g = 0; g < n_grp; ++g) {
        const int j1 = std::min(cfg->n_filt, g * 64 + 64);
        // a filter's reads start at its first bin rounded down to 4 and cover `steps` 16-byte steps (the group's widest filter,
        // rounded up to 2 steps); reads that would leave the wave buffer start earlier instead (leading zero taps)
        int32_t steps = 2, taps = 0;
        for (int pass = 0; pass < 8; ++pass) {
            taps = 0;
            const int lo_cap = std::max(0, (buf_floats - 4 * steps) & ~3);
            for (int j = g * 64; j < j1; ++j) {
                lo4[j] = std::min(lo[j] & ~3, lo_cap);
                taps = std::max(taps, len[j] + (lo[j] - lo4[j]));
            }
            const int32_t need = std::max(2, ((taps + 7) / 8) * 2);
            if (need <= steps) break;
            steps = need;
        }
        const int nl = j1 - g * 64;
        for (int j = g * 64; j < j1; ++j)
            if (lo4[j] + 4 * steps > buf_floats || len[j] + (lo[j] - lo4[j]) > 4 * steps) {
                delete p;
                SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: filter %d (bins %d..%d) next to %d-tap filters does not fit the kernel's spectrum row",
                         j, lo[j], lo[j] + len[j] - 1, taps);
            }
        grp[g] = steps;
        grp[8 + g] = (int32_t)(wT.size() / 4);
        const size_t base = wT.size();
        wT.resize(base + (size_t)steps * nl * 4, 0.f);
        for (int j = g * 64; j < j1; ++j)
            for (int k = 0; k < len[j]; ++k) {
                const int tap = k + (lo[j] - lo4[j]);
                wT[base + ((size_t)(tap / 4) * nl + (j - g * 64)) * 4 + (tap & 3)] = w[(size_t)ofs[j] + k];
            }
    }
    ga.filt_w4_total = (int32_t)(wT.size() / 4);
    ga.dct_identity = cfg->n_ceps == cfg->n_filt;
    for (int q = 0; q < cfg->n_ceps && ga.dct_identity; ++q)
        for (int j = 0; j < cfg->n_filt; ++j)
            if (dct[(size_t)q * cfg->n_filt + j] != (q == j ? 1.f : 0.f)) {
                ga.dct_identity = 0;
                break;
            }
    ga.dct_ncp = 1;
    while (ga.dct_ncp < std::min(cfg->n_ceps, 64)) ga.dct_ncp *= 2;
    std::vector<float> dctT((size_t)cfg->n_ceps * cfg->n_filt);
    for (int q = 0; q < cfg->n_ceps; ++q)
        for (int j = 0; j < cfg->n_filt; ++j) dctT[(size_t)j * cfg->n_ceps + q] = dct[(size_t)q * cfg->n_filt + j];
    std::vector<float> dctv(dct, dct + (size_t)cfg->n_ceps * cfg->n_filt);
    std::vector<float> dense(fbank, fbank + (size_t)cfg->n_filt * nb);
    hipStream_t s = ctx->stream;
    int rc = upload(p->window, win, s);
    if (rc == SSP_OK) rc = upload(p->twiddle, tw, s);
    if (rc == SSP_OK) rc = upload(p->filt_lo4, lo4, s);
    if (rc == SSP_OK) rc = upload(p->filt_grp, grp, s);
    if (rc == SSP_OK) rc = upload(p->filt_wT, wT, s);
    if (rc == SSP_OK) rc = upload(p->dct, dctv, s);
    if (rc == SSP_OK) rc = upload(p->dctT, dctT, s);
    if (rc == SSP_OK) rc = upload(p->fbank_dense, dense, s);
    if (rc == SSP_OK && hipStreamSynchronize(s) != hipSuccess) {
        set_error("mfcc plan: table upload failed");
        rc = SSP_ERR_HIP;
    }
    if (rc == SSP_OK) rc = build_s2k_tables(p);  // (n_fft == 2048 without deltas; leaves s2k_ready false otherwise)
    if (rc == SSP_OK && mfcc_fast_supported(*cfg)) {
        const int frc = build_fast_tables(p);  // a filterbank the fused kernel cannot lay out only disables that kernel
        if (frc != SSP_OK && frc != SSP_ERR_UNSUPPORTED) rc = frc;
        if (frc == SSP_OK && mfcc_stream_supported(p)) rc = build_stream_tables(p);
    }
    if (rc != SSP_OK) {
        delete p;
        return rc;
    }
    MfccArgs& a = p->args;
    a.window = p->window.as<float>();
    a.twiddle = p->twiddle.as<float2>();
    a.filt_lo4 = p->filt_lo4.as<int32_t>();
    a.filt_grp = p->filt_grp.as<int32_t>();
    a.filt_wT = p->filt_wT.as<float>();
    a.dct = p->dct.as<float>();
    a.dctT = p->dctT.as<float>();
    a.win_len = cfg->win_len;
    a.hop = cfg->hop;
    a.n_fft = cfg->n_fft;
    a.n_filt = cfg->n_filt;
    a.n_ceps = cfg->n_ceps;
    a.d_out = p->d_out;
    a.frame_mode = cfg->frame_mode;
    a.preemph_mode = cfg->preemph_mode;
    a.spec_power = cfg->spec_power;
    a.log_mode = cfg->log_mode;
    a.floor_mode = cfg->floor_mode;
    a.delta_order = cfg->delta_order;
    a.delta_N = cfg->delta_order > 0 ? cfg->delta_N : 0;
    a.cmvn = cfg->cmvn;
    a.preemph = cfg->preemph;
    a.spec_scale = cfg->spec_scale;
    a.eps = cfg->eps;
    a.top_db = cfg->top_db;
    int den = 0;
    for (int i = 1; i <= a.delta_N; ++i) den += 2 * i * i;  // GMM_UBM.py:61
    a.delta_inv_denom = den > 0 ? 1.0f / (float)den : 0.f;
    *out = p;
    return SSP_OK;
}

int ssp_mfcc_plan_destroy(ssp_mfcc_plan* plan) {
    if (!plan) return SSP_OK;
    ssp::quiesce_ctx(plan->ctx);  // (the ctx may already be gone: common.hpp)
    delete plan;
    return SSP_OK;
}

int ssp_mfcc_plan_set_flags(ssp_mfcc_plan* plan, uint32_t flags) {
    if (!plan) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_plan_set_flags: null");
    if (flags & ~(uint32_t)SSP_MFCC_REPRODUCIBLE) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_plan_set_flags: unknown flag 0x%x", flags);
    const bool r = (flags & SSP_MFCC_REPRODUCIBLE) != 0;
    if (r != plan->reproducible) {
        plan->reproducible = r;
        plan->cache_sseg = plan->cache_fseg = 0;  // the cached work table was laid out under the other rule
        plan->cache_variant = plan->cache_request = -1;
    }
    return SSP_OK;
}

int ssp_mfcc_frame_segments(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, ssp_segments** frame_seg_out) {
    if (!plan || !sample_seg || !frame_seg_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_frame_segments: null");
    std::vector<int64_t> fo((size_t)sample_seg->n + 1);
    fo[0] = 0;
    for (int64_t u = 0; u < sample_seg->n; ++u) {
        const int64_t ns = sample_seg->host[u + 1] - sample_seg->host[u];
        if (plan->cfg.frame_mode == 2 && ns > 0 && ns <= plan->cfg.n_fft / 2)
            SSP_FAIL(SSP_ERR_INVALID, "mfcc: utterance %lld has %lld samples; reflect padding needs > n_fft/2 = %d",
                     (long long)u, (long long)ns, plan->cfg.n_fft / 2);
        fo[u + 1] = fo[u] + frames_for(plan->cfg, ns);
    }
    return segments_make(plan->ctx, fo.data(), sample_seg->n, frame_seg_out);
}

// ---- the run: checks + work table (run_prepare), launches over a range of utterances (run_launch), and the three ways data gets there
}  // extern "C"

namespace ssp {

// int16 PCM (what utils/tools.py:45-47 / scipy.io.wavfile hand the reference's extractors) -> the float32 the kernels read: the value
// itself, no scaling (sidekit's mfcc takes the integers as they are).  8 samples per thread: one 16-byte read, two 16-byte writes.
__global__ __launch_bounds__(256) void widen_i16_kernel(const int16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n && ((reinterpret_cast<uintptr_t>(in + i) & 15) == 0)) {
        const int4 v = *reinterpret_cast<const int4*>(in + i);
        const int w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = (float)(int16_t)(w[k] & 0xffff);
            f[2 * k + 1] = (float)(int16_t)(w[k] >> 16);
        }
        *reinterpret_cast<float4*>(out + i) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(out + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        for (int64_t k = i; k < n && k < i + 8; ++k) out[k] = (float)in[k];
    }
}

static int launch_widen_i16(const int16_t* in, float* out, int64_t n, hipStream_t s) {
    if (n <= 0) return SSP_OK;
    const int64_t blocks = (n + 2047) / 2048;
    if (blocks > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: too many int16 samples for one widening launch");
    hipLaunchKernelGGL(widen_i16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

static int run_prepare(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg, int variant) {
    if (plan->checked_sseg != sample_seg->serial || plan->checked_fseg != frame_seg->serial) {
        // frame segments must follow the plan's framing rule (checked once per segment pair: segments are immutable, and the loop
        // is 0.2 ms of host time at 100k utterances)
        for (int64_t u = 0; u < frame_seg->n; ++u) {
            const int64_t T = frame_seg->host[u + 1] - frame_seg->host[u];
            if (T != frames_for(plan->cfg, sample_seg->host[u + 1] - sample_seg->host[u]))
                SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_run: frame segment %lld does not match the framing rule", (long long)u);
        }
        plan->checked_sseg = sample_seg->serial;
        plan->checked_fseg = frame_seg->serial;
    }
    int v = variant;
    if (v == 0) {
        v = (mfcc_fast_supported(plan->cfg) && plan->fast_ready) ? 2 : 1;
        // wide / dense filterbanks (e.g. the Bark rows of the PLP front end) fall to the fused kernel's banded LDS sweep, which the
        // generic kernel's register-accumulated 4-tap reads beat (62 vs 75 ms at 21 x 257 taps)
        if (v == 2 && plan->fast.melv == 0 && plan->max_filt_len > 64) v = 1;
    }
    if (v == 2 && !(mfcc_fast_supported(plan->cfg) && plan->fast_ready))
        SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_mfcc_run: the fused fast kernel does not cover this cfg");
    const bool stream_ok = plan->stream_ready && mfcc_stream_supported(plan);
    if (v == 3 && !stream_ok) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_mfcc_run: the wave-stream kernel does not cover this cfg");
    if (variant == 0 && stream_ok && (v == 2 || mfcc_stream_dense(plan))) v = 3;
    if (v == 4 && !mfcc_s2k_supported(plan)) SSP_FAIL(SSP_ERR_UNSUPPORTED, "ssp_mfcc_run: the 2048-point wave-stream kernel does not cover this cfg");
    if (variant == 0 && v == 1 && mfcc_s2k_supported(plan)) v = 4;
    // the work table is cached per (segment pair, REQUESTED variant): an auto request that fell back to another kernel is remembered
    // as such instead of being rebuilt (chunk table upload + stream sync) on every call
    if (plan->cache_sseg != sample_seg->serial || plan->cache_fseg != frame_seg->serial || plan->cache_request != variant) {
        int brc = build_work(plan, sample_seg, frame_seg, v);
        if (brc == SSP_ERR_UNSUPPORTED && variant == 0 && v == 3) {  // auto: a batch the stream kernel cannot lay out goes to the workgroup kernel
            v = 2;
            brc = build_work(plan, sample_seg, frame_seg, v);
        }
        if (brc == SSP_ERR_UNSUPPORTED && variant == 0 && v == 4) {  // auto: ... the 2048-point stream kernel cannot lay out goes to the generic one
            v = 1;
            brc = build_work(plan, sample_seg, frame_seg, v);
        }
        if (brc == SSP_ERR_UNSUPPORTED && variant == 0 && v == 2) {  // auto: a batch the fused kernel cannot lay out goes to the generic one
            v = 1;
            brc = build_work(plan, sample_seg, frame_seg, v);
        }
        SSP_TRY(brc);
        plan->cache_request = variant;
    }
    return SSP_OK;
}

// the cached work table's launches for utterances [u0, u1) (the whole batch: 0, n).  `d_samples` / `d_out` are addressed with the
// batch's ABSOLUTE offsets (sample_off[u], frame_off[u] . d_out): a caller that holds only a slice passes pointers biased by the
// slice's first offsets — the kernels never touch an address outside [u0, u1)'s samples and rows.
static int run_launch(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg, const float* d_samples, float* d_out,
                      int64_t u0, int64_t u1, hipStream_t s) {
    const int v = plan->cache_variant;
    const bool all = u0 == 0 && u1 == frame_seg->n;
    const int c0 = plan->cache_chunk_first[(size_t)u0], n_chunks = plan->cache_chunk_first[(size_t)u1] - c0;
    if (n_chunks <= 0) return SSP_OK;
    const int64_t total_frames = frame_seg->total();
    MfccArgs a = plan->args;
    if (plan->cache_split_cmvn) a.cmvn = 0;  // utterances longer than one workgroup's chunk: normalised by the CMVN kernel below
    a.lm_out = nullptr;
    if (plan->cache_split_topdb && (v == 1 || v == 4)) {  // two-pass top_db: log-mel rows to a scratch, clamp + DCT in a second kernel
        if (!all) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc: the two-pass top_db path runs on whole batches");
        SSP_TRY(plan->lm_scratch.reserve((size_t)total_frames * plan->cfg.n_filt * sizeof(float)));
        a.lm_out = plan->lm_scratch.as<float>();
        SSP_TRY(plan->umax_scratch.reserve((size_t)frame_seg->n * sizeof(float)));
        a.utt_max = plan->umax_scratch.as<float>();
        SSP_HIP(hipMemsetD32Async((hipDeviceptr_t)a.utt_max, (int)0xff800000u, (size_t)frame_seg->n, s));  // -inf
        a.top_db = -1.f;
    }
    a.samples = d_samples;
    a.sample_off = sample_seg->dev.as<int64_t>();
    a.frame_off = frame_seg->dev.as<int64_t>();
    a.out = d_out;
    a.chunks = plan->chunks.as<MfccChunk>() + c0;
    if (v == 4)
        SSP_TRY(launch_mfcc_s2k(a, plan, n_chunks, s));
    else if (v == 3)
        SSP_TRY(launch_mfcc_stream(a, plan, n_chunks, s));
    else if (v == 2)
        SSP_TRY(launch_mfcc_fast(a, plan, n_chunks, plan->cache_chunk_frames, s));
    else
        SSP_TRY(launch_mfcc_generic(a, n_chunks, plan->cache_lds, plan->cache_waves, plan->ctx->num_cu, s));
    if (a.lm_out && !(v == 4 && plan->cache_s2k_fused))
        SSP_TRY(launch_topdb_dct(a.lm_out, frame_seg->dev.as<int64_t>(), frame_seg->n, a.chunks, n_chunks, a.utt_max,
                                 plan->cfg.n_filt, plan->cfg.n_ceps, plan->dct.as<float>(),
                                 plan->cfg.top_db >= 0.f ? plan->cfg.top_db : INFINITY /* no clamp: max - inf */, d_out, s));
    if (plan->cache_split_cmvn)  // (offsets stay absolute: the table pointer moves to utterance u0, the data pointer does not)
        SSP_TRY(launch_cmvn(d_out, d_out, frame_seg->dev.as<int64_t>() + u0, u1 - u0, plan->d_out, frame_seg->max_len(), s));
    return SSP_OK;
}

// Large host-fed batches (and int16 input from either side): the batch goes through the ctx's ring of slice-sized slots — slice i + 1
// copies in while slice i computes and slice i - 1's features copy back (HostPipe, staging.hpp).  Slices are runs of whole utterances of
// about `slice_bytes` of fp32 samples (an utterance longer than that is a slice of its own).  host_in / host_out: where the operand
// lives (int16 device input with device output only widens through the ring; nothing is copied).
static int run_sliced(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg, const void* samples, int stype,
                      float* feats_out, bool host, size_t slice_bytes, float* kernel_ms) {
    ssp_ctx* ctx = plan->ctx;
    hipStream_t cs = ctx->stream;
    if (!ctx->pipe) {
        ctx->pipe = new (std::nothrow) HostPipe;
        if (!ctx->pipe) SSP_FAIL(SSP_ERR_NOMEM, "mfcc: host alloc (pipeline)");
    }
    HostPipe& hp = *ctx->pipe;
    SSP_TRY(hp.init());
    const int64_t n = frame_seg->n, D = plan->d_out;
    const std::vector<int64_t>&so = sample_seg->host, &fo = frame_seg->host;
    // slices: [cut[i], cut[i + 1]) utterances
    std::vector<int64_t> cut{0};
    size_t max_in = 0, max_out = 0;
    {
        const int64_t per = (int64_t)(slice_bytes / sizeof(float));
        int64_t u = 0;
        while (u < n) {
            int64_t e = u + 1;
            while (e < n && so[e + 1] - so[u] <= per) ++e;
            max_in = std::max(max_in, (size_t)(so[e] - so[u]));
            max_out = std::max(max_out, (size_t)(fo[e] - fo[u]) * (size_t)D);
            cut.push_back(e);
            u = e;
        }
    }
    const int n_slices = (int)cut.size() - 1;
    // (a slot that has to grow may still be read by work of an earlier call on the other streams: everything is drained first)
    bool grow = false;
    for (int k = 0; k < HostPipe::RING; ++k)
        grow = grow || hp.in[k].bytes < max_in * 4 + 4096 || (host && hp.out[k].bytes < max_out * 4 + 16) || (stype == 1 && host && hp.raw[k].bytes < max_in * 2 + 64);
    if (grow) {
        SSP_HIP(hipStreamSynchronize(cs));
        SSP_HIP(hipStreamSynchronize(hp.h2d));
        SSP_HIP(hipStreamSynchronize(hp.d2h));
        for (int k = 0; k < HostPipe::RING; ++k) {
            SSP_TRY(hp.in[k].reserve(max_in * 4 + 4096));
            if (host) SSP_TRY(hp.out[k].reserve(max_out * 4 + 16));
            if (stype == 1 && host) SSP_TRY(hp.raw[k].reserve(max_in * 2 + 64));
        }
    }
    if (plan->cache_variant == 3)   // (the launcher's grow-only counter / flag buffer: sized for the whole table now, not by a later, larger slice)
        SSP_TRY(plan->f_counter.reserve(64 + (size_t)std::max(plan->cache_n_chunks, 1) * sizeof(int32_t)));
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, cs));
    // the copy streams start behind whatever the ctx stream holds (the slots' last readers of an earlier call included)
    SSP_HIP(hipEventRecord(hp.computed[0], cs));
    SSP_HIP(hipStreamWaitEvent(hp.h2d, hp.computed[0], 0));
    SSP_HIP(hipStreamWaitEvent(hp.d2h, hp.computed[0], 0));
    const size_t esz = stype == 1 ? sizeof(int16_t) : sizeof(float);
    // SSP_HOST_TRACE=1 (diagnostic): per slice, when its copy-in, its kernels and its copy-back ended (ms from the call's start, stderr)
    const bool trace = getenv("SSP_HOST_TRACE") != nullptr;
    std::vector<hipEvent_t> tev;
    hipEvent_t t_start = nullptr;
    if (trace) {
        tev.resize((size_t)n_slices * 3, nullptr);
        for (hipEvent_t& e : tev) SSP_HIP(hipEventCreate(&e));
        SSP_HIP(hipEventCreate(&t_start));
        SSP_HIP(hipEventRecord(t_start, cs));
    }
    for (int i = 0; i < n_slices; ++i) {
        const int k = i % HostPipe::RING;
        const int64_t u0 = cut[(size_t)i], u1 = cut[(size_t)i + 1];
        const int64_t s0 = so[(size_t)u0], ns = so[(size_t)u1] - s0, f0 = fo[(size_t)u0], nf = fo[(size_t)u1] - f0;
        const char* src = static_cast<const char*>(samples) + (size_t)(s0 - so[0]) * esz;   // (the caller's array starts at the first utterance's first sample)
        float* d_in = hp.in[k].as<float>();
        if (host) {
            // slot k's last reader was slice i - RING.  The HOST waits for it (it runs at most RING slices ahead, and the call is synchronous
            // anyway): a device-side hipStreamWaitEvent on the copy stream made every copy-in start only when the previous slice's
            // copy-back had ended (measured per slice with SSP_HOST_TRACE: 1.70 ms a slice instead of 1.20) — the two directions serialised
            if (i >= HostPipe::RING) SSP_HIP(hipEventSynchronize(hp.computed[k]));
            SSP_HIP(hipMemcpyAsync(stype == 1 ? hp.raw[k].p : (void*)d_in, src, (size_t)ns * esz, hipMemcpyHostToDevice, hp.h2d));
            SSP_HIP(hipEventRecord(hp.in_ready[k], hp.h2d));
            if (trace) SSP_HIP(hipEventRecord(tev[(size_t)i * 3], hp.h2d));
            SSP_HIP(hipStreamWaitEvent(cs, hp.in_ready[k], 0));
            if (i >= HostPipe::RING) SSP_HIP(hipEventSynchronize(hp.out_done[k]));              // ... and its features have left the out slot
            if (stype == 1) SSP_TRY(launch_widen_i16(hp.raw[k].as<int16_t>(), d_in, ns, cs));
        } else {
            SSP_TRY(launch_widen_i16(reinterpret_cast<const int16_t*>(src), d_in, ns, cs));   // (in-order on cs: slot k's last reader is long done)
        }
        float* d_o = host ? hp.out[k].as<float>() - (size_t)(f0 - fo[0]) * (size_t)D : feats_out;
        // pointers biased by the slice's first offsets: the kernels address with the batch's absolute offsets (run_launch)
        SSP_TRY(run_launch(plan, sample_seg, frame_seg, d_in - (size_t)(s0 - so[0]), d_o, u0, u1, cs));
        if (host) {
            SSP_HIP(hipEventRecord(hp.computed[k], cs));
            if (trace) SSP_HIP(hipEventRecord(tev[(size_t)i * 3 + 1], cs));
            SSP_HIP(hipStreamWaitEvent(hp.d2h, hp.computed[k], 0));
            SSP_HIP(hipMemcpyAsync(feats_out + (size_t)(f0 - fo[0]) * (size_t)D, hp.out[k].p, (size_t)nf * (size_t)D * sizeof(float), hipMemcpyDeviceToHost, hp.d2h));
            SSP_HIP(hipEventRecord(hp.out_done[k], hp.d2h));
            if (trace) SSP_HIP(hipEventRecord(tev[(size_t)i * 3 + 2], hp.d2h));
        }
    }
    if (host) {   // the ctx stream ends behind the last copy-back (one stream to wait on, for this call and for whoever comes next)
        SSP_HIP(hipStreamWaitEvent(cs, hp.out_done[(n_slices - 1) % HostPipe::RING], 0));
    }
    SSP_TRY(tm.stop(cs, kernel_ms));
    if (host) SSP_HIP(hipStreamSynchronize(cs));
    if (trace) {
        if (host) {
            fprintf(stderr, "[ssp host pipeline] %d slices of <= %zu MiB (fp32); ms from start: copy-in done | kernels done | copy-back done\n", n_slices, slice_bytes >> 20);
            for (int i = 0; i < n_slices; ++i) {
                float a = 0.f, b = 0.f, c = 0.f;
                (void)hipEventElapsedTime(&a, t_start, tev[(size_t)i * 3]);
                (void)hipEventElapsedTime(&b, t_start, tev[(size_t)i * 3 + 1]);
                (void)hipEventElapsedTime(&c, t_start, tev[(size_t)i * 3 + 2]);
                fprintf(stderr, "[ssp host pipeline] slice %3d: %8.3f %8.3f %8.3f\n", i, a, b, c);
            }
        }
        for (hipEvent_t e : tev) (void)hipEventDestroy(e);
        (void)hipEventDestroy(t_start);
    }
    return SSP_OK;
}


static int mfcc_run_any(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg, const void* samples, int stype,
                        float* feats_out, int where, int variant, float* kernel_ms) {
    if (!plan || !sample_seg || !frame_seg) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_run: null handle");
    SSP_TRY(use_ctx(plan->ctx));
    if (sample_seg->n != frame_seg->n) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_run: sample/frame segment counts differ");
    if (where != SSP_HOST && where != SSP_DEVICE) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_run: where");
    if (variant < 0 || variant > 4) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_run: variant");
    const int64_t total_frames = frame_seg->total();
    const int64_t n_samp_total = sample_seg->host.back();
    if (kernel_ms) *kernel_ms = 0.f;
    if (total_frames == 0) return SSP_OK;
    if (!samples || !feats_out) SSP_FAIL(SSP_ERR_INVALID, "ssp_mfcc_run: null data pointer");
    SSP_TRY(run_prepare(plan, sample_seg, frame_seg, variant));
    const int v = plan->cache_variant;
    hipStream_t s = plan->ctx->stream;
    const size_t out_bytes = (size_t)(frame_seg->host.back()) * plan->d_out * sizeof(float);
    const size_t slice = host_slice_bytes();
    const bool two_pass = plan->cache_split_topdb && (v == 1 || v == 4);   // (needs the whole batch's rows in one scratch)
    const bool big = (size_t)n_samp_total * sizeof(float) >= 2 * slice && frame_seg->n >= 2 && sample_seg->host.front() == 0 && frame_seg->host.front() == 0;
    if (!two_pass && ((where == SSP_HOST && big) || (stype == 1 && where == SSP_DEVICE && sample_seg->host.front() == 0 && frame_seg->host.front() == 0)))
    {
        const int src = run_sliced(plan, sample_seg, frame_seg, samples, stype, feats_out, where == SSP_HOST, slice, kernel_ms);
        if (src != SSP_OK && plan->ctx->pipe) {
            // a call that failed half way must not leave copies in flight that read / write the CALLER's arrays after it has returned:
            // the three streams are drained before the error goes up (the message of the first failure stays)
            HostPipe& hp = *plan->ctx->pipe;
            (void)hipStreamSynchronize(s);
            if (hp.h2d) (void)hipStreamSynchronize(hp.h2d);
            if (hp.d2h) (void)hipStreamSynchronize(hp.d2h);
        }
        return src;
    }

    // one piece: the operands as they are (device pointers), or staged whole through the ctx's pool (host pointers)
    Staged sin, sout, sraw;
    int rc;
    const float* d_samples;
    if (stype == 1) {
        const int16_t* d_raw = (const int16_t*)sraw.in(plan->ctx, samples, (size_t)n_samp_total * sizeof(int16_t), where, &rc);
        SSP_TRY(rc);
        SSP_TRY(sin.get(plan->ctx, (size_t)n_samp_total * sizeof(float) + 16));
        SSP_TRY(launch_widen_i16(d_raw, (float*)sin.p, n_samp_total, s));
        d_samples = (const float*)sin.p;
    } else {
        d_samples = (const float*)sin.in(plan->ctx, samples, (size_t)n_samp_total * sizeof(float), where, &rc);
        SSP_TRY(rc);
    }
    float* d_out = (float*)sout.out(plan->ctx, feats_out, out_bytes, where, &rc);
    SSP_TRY(rc);
    Timer tm;
    SSP_TRY(tm.start(kernel_ms != nullptr, s));
    SSP_TRY(run_launch(plan, sample_seg, frame_seg, d_samples, d_out, 0, frame_seg->n, s));
    SSP_TRY(tm.stop(s, kernel_ms));
    SSP_TRY(sout.back(plan->ctx, feats_out, out_bytes, where));
    if (where == SSP_HOST || stype == 1) SSP_HIP(hipStreamSynchronize(s));   // (the staging slots are given back at return)
    return SSP_OK;
}

}  // namespace ssp

extern "C" {

int ssp_mfcc_run(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg,
                 const float* samples, float* feats_out, int where, int variant, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_mfcc_run");
    return mfcc_run_any(plan, sample_seg, frame_seg, samples, 0, feats_out, where, variant, kernel_ms);
}

int ssp_mfcc_run_i16(ssp_mfcc_plan* plan, const ssp_segments* sample_seg, const ssp_segments* frame_seg,
                     const int16_t* samples, float* feats_out, int where, int variant, float* kernel_ms) {
    ssp::TraceRange trace_("ssp_mfcc_run_i16");
    return mfcc_run_any(plan, sample_seg, frame_seg, samples, 1, feats_out, where, variant, kernel_ms);
}

}  // extern "C"

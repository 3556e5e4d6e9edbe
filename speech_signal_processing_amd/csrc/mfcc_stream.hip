// Host side of the wave-stream MFCC kernel (mfcc_stream_kernel.hpp), its FIRST kernel's instances and the scan kernel behind them; the
// third kernel's instances — the rare walk over chunks that held a non-finite cepstrum — are compiled in mfcc_stream_walk.hip.
#include "mfcc_stream_kernel.hpp"

namespace ssp {

// ------------------------------------------------------------------------------------------------ host side
// dense-band instance: <= 24 filterbank rows that the piece filterbank cannot hold, identity "DCT", no deltas, the sidekit front end
bool mfcc_stream_dense(const ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    return mfcc_fast_supported(c) && p->fast_ready && !getenv("SSP_MFCC_NO_STREAM") && p->fast.melv == 0 && p->args.dct_identity &&
           c.n_filt <= 24 && c.delta_order == 0 && c.win_len > 384 && c.win_len <= 416 && c.spec_power == 2 && c.preemph_mode != 0;
    // (win_len > 384: this instance takes the legacy window product on the LAST 32-sample row only and has no scan / walk kernels behind
    //  it; a shorter window has several padded rows through which a NaN sample behind a frame would reach it — those plans take the fused kernel)
}

bool mfcc_stream_supported(const ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    const FastArgs& f = p->fast;
    if (!(mfcc_fast_supported(c) && p->fast_ready)) return false;
    if (getenv("SSP_MFCC_NO_STREAM")) return false;
    if (mfcc_stream_dense(p)) return true;
    const int ks = (c.n_filt + 3) / 4;
    return f.melv >= 2 && f.melv <= (ks <= 6 ? 4 : 5) && c.n_ceps == 13 && ks <= 10 && (c.delta_order == 0 || c.delta_N == 2);
}

// the instances that scale the features themselves (cmvn): the sidekit call-site family
bool mfcc_stream_fuses_cmvn(const ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    return mfcc_stream_supported(p) && !mfcc_stream_dense(p) && c.win_len <= 416 && c.spec_power == 2 && c.preemph_mode != 0 && (c.n_filt + 3) / 4 <= 6;
}


int build_stream_tables(ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    const int KS = stream_ks(c);
    // A operand of the DCT product: lane (ceps = l & 15, kq = l >> 4), k-step s <-> filter KS kq + s
    std::vector<float> dcth((size_t)c.n_ceps * c.n_filt), dA((size_t)KS * 64, 0.f);
    SSP_HIP(hipMemcpy(dcth.data(), p->dct.p, dcth.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int s = 0; s < KS; ++s)
        for (int l = 0; l < 64; ++l) {
            const int q = l & 15, jf = KS * (l >> 4) + s;
            if (q < c.n_ceps && jf < c.n_filt) dA[(size_t)s * 64 + l] = dcth[(size_t)q * c.n_filt + jf];
        }
    SSP_TRY(p->s_dctA.alloc(dA.size() * sizeof(float)));
    SSP_HIP(hipMemcpy(p->s_dctA.p, dA.data(), dA.size() * sizeof(float), hipMemcpyHostToDevice));
    if (mfcc_stream_dense(p)) {
        // lane (c = l & 15, b = l >> 4): bands 6 b + k on bins 16 c + i (i < 16) and, on chunk 15 only, bin 256; the split step leaves
        // 2 X[k], the scale of the spectrum is folded in (as the piece filterbank does)
        const int nb = 257;
        std::vector<float> fb((size_t)c.n_filt * nb), dw((size_t)64 * 6 * 20, 0.f);
        SSP_HIP(hipMemcpy(fb.data(), p->fbank_dense.p, fb.size() * sizeof(float), hipMemcpyDeviceToHost));
        const float pscale = (c.spec_power == 2 ? 0.25f : 0.5f) * c.spec_scale;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 6; ++k) {
                const int band = 6 * (l >> 4) + k, ch = l & 15;
                if (band >= c.n_filt) continue;
                float* d = dw.data() + ((size_t)l * 6 + k) * 20;
                for (int i = 0; i < 16; ++i) d[i] = pscale * fb[(size_t)band * nb + 16 * ch + i];
                if (ch == 15) d[16] = pscale * fb[(size_t)band * nb + 256];
            }
        SSP_TRY(p->s_dense.alloc(dw.size() * sizeof(float)));
        SSP_HIP(hipMemcpy(p->s_dense.p, dw.data(), dw.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    p->stream_ready = true;
    return SSP_OK;
}

// ------------------------------------------------------------------------------------------------ the scan between the two
// mfcc_stream_scan_kernel — the SECOND kernel of a launch: which chunks did the first kernel get wrong?  It reads the answer off the rows
// the first kernel stored and sets redo_flags[chunk] (+ the "any" word the third kernel asks first).
//
// What the first kernel's arithmetic guarantees (mfcc_stream512_kernel, WALK = 0).  A frame's cepstra are finite or non-finite TOGETHER
// (each is a sum over the same log-mel row; a non-finite term times any weight, zero included, is non-finite; finite rows of O(10) cannot
// overflow), and a matrix product spreads a non-finite operand over every element it contributes to with ANY weight, zero included:
//   * transposed step (interior steps of the non-scaling instances): the delta product contracts over all 24 frames of the step's window
//     — one non-finite frame in reach makes the delta columns of EVERY emitted row of the step non-finite; the cepstra leave straight
//     from the ring (a row is wrong only if its own frame is);
//   * chained step (utterance ends; every step of the scaling instances): cepstra leave straight from the ring (a row is wrong only if
//     its own frame is), delta tile 0 = emitted rows up to rb + 9 contracts over frames rb - 8 .. rb + 11, tile 1 = rows rb + 10,
//     rb + 11 over rb + 8 .. rb + 15, delta-delta over both tiles — so the delta columns of the FIRST and the LAST emitted row of a step
//     tell for every emitted row of it (a step that emits rows of one tile only has both in that tile);
//   * scaling (CM): a non-finite entry makes its column's mean — and with it the column of the whole utterance — NaN.
// A row is wrong in the first kernel's output only if it is non-finite there (finite rows were formed from finite operands by the same
// sums as ever), so: delta_order >= 1 — the first column of the HIGHEST-order block (its product sees everything the lower ones saw: the
// chained delta-delta contracts over both delta tiles) of one emitted row of a transposed step, of the first and the last emitted row
// of a chained one; delta_order 0 — the first cepstrum of every emitted row (no product that would spread a leaked NaN sample's frame).  Rows that are non-finite in the reference too are flagged as well; the
// third kernel reproduces them.  Cost: one 32-byte sector per 16 rows and a few more at the utterance ends.
// One THREAD per (chunk, look): `per_chunk` looks per chunk (enough for the longest chunk of the table; a look past a chunk's last
// step does nothing), each three dependent loads deep — chunk record, frame offset, the row's word — and millions of them in flight.
// (A wave per chunk with lanes over its looks took 0.09 ms at configs[1], this takes 0.02.)  The flags were zeroed by the first launch;
// a look that finds a non-finite word sets its chunk's flag and the "any" word (plain stores of 1: whoever comes last writes the same).
__global__ __launch_bounds__(256) void mfcc_stream_scan_kernel(MfccArgs a, StreamArgs sa, int per_chunk) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cidx = (int)(i / per_chunk), k = (int)(i - (int64_t)cidx * per_chunk);
    if (cidx >= sa.n_chunks) return;
    const MfccChunk ch = a.chunks[cidx];
    const int64_t f0 = a.frame_off[ch.utt];
    const int dord = a.delta_order, Dd = a.d_out, t0 = ch.t0, n = ch.n;
    // the step geometry of the first kernel: ta = first computed frame (no halo without deltas), step b emits rows
    // [ta + 16 b - 4, ta + 16 b + 12) of the chunk's; a step whose window lies strictly inside the utterance takes the transposed form
    const int ta = dord > 0 ? max(t0 - 4 - ch.pad, 0) : t0;
    const int n_steps = (t0 + n - ta + 4 + 15) >> 4;
    const int T = (int)(a.frame_off[ch.utt + 1] - f0);
    int F = -1, col = 0;
    if (dord > 0) {
        // a transposed step pollutes every row it emits: ONE look; a chained step's two delta tiles: its first and its last emitted row
        const int rb = ta + 16 * (k >> 1);
        const bool spread = sa.tstep != 0 && rb - 8 >= 1 && rb + 16 <= T - 2;
        const int lo = max(rb - 4, t0), hi = min(rb + 12, t0 + n);
        if ((k >> 1) < n_steps && lo < hi && !(spread && (k & 1))) F = (k & 1) ? hi - 1 : lo;
        col = 13 * dord;
    } else {
        // without deltas both step forms store the ring's rows as they are: every row is looked at
        const int rb = ta + 16 * (k >> 4), r = k & 15;
        const int lo = max(rb - 4, t0), hi = min(rb + 12, t0 + n);
        const int Fc = rb - 4 + r;
        if ((k >> 4) < n_steps && Fc >= lo && Fc < hi) F = Fc;
    }
    if (F < 0) return;
    const float v = a.out[((size_t)f0 + (size_t)F) * Dd + col];
    if (!(__builtin_fabsf(v) < INFINITY)) {  // NaN or +-inf
        sa.redo_flags[cidx] = 1;
        sa.work_counter[1] = 1;
    }
}

int launch_mfcc_stream_walk(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream);  // mfcc_stream_walk.hip

int launch_mfcc_stream(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream, bool dry_run) {
    SSP_TRY(launch_mfcc_stream_impl<0>(args, p, n_chunks, stream, dry_run));
    if (dry_run || n_chunks <= 0 || mfcc_stream_dense(p)) return SSP_OK;
    // the second kernel: which chunks came out polluted by a non-finite cepstrum (normally none) ...
    StreamArgs sa{};
    sa.n_chunks = n_chunks;
    sa.work_counter = p->f_counter.as<int32_t>();
    sa.redo_flags = sa.work_counter + 16;
    sa.tstep = (args.cmvn == 0 && !(p->fast.melv >= 4 && p->fast.mel_ns > 2)) ? 1 : 0;  // (mfcc_stream512_kernel: TSTEP)
    // looks per chunk: two per step with deltas, sixteen without; steps of the longest chunk the table may hold (+ halo and pad)
    const int max_steps = (p->cache_chunk_frames + 16 + 4 + 15) >> 4;
    const int per_chunk = (args.delta_order > 0 ? 2 : 16) * max_steps;
    const int64_t looks = (int64_t)n_chunks * per_chunk;
    if ((looks + 255) / 256 > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream): too many chunks for the scan kernel's grid");
    hipLaunchKernelGGL(mfcc_stream_scan_kernel, dim3((unsigned)((looks + 255) / 256)), dim3(256), 0, stream, args, sa, per_chunk);
    SSP_HIP(hipGetLastError());
    // ... and the third: those chunks once more, term by term (every workgroup reads the "any" word and leaves when it is zero)
    return launch_mfcc_stream_walk(args, p, n_chunks, stream);
}

}  // namespace ssp

#ifdef SSP_S_CLOCK
extern "C" int ssp_debug_clock(ssp_mfcc_plan* p, unsigned long long* out2) {
    unsigned long long h[8];
    if (hipMemcpy(h, p->f_counter.p, 64, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    out2[0] = h[4];  // (64-bit words behind the launch counters: work_counter + 8, + 10)
    out2[1] = h[5];
    return 0;
}
#endif

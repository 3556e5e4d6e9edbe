// (shared by mfcc_stream.hip — the first kernel of a launch, the host side — and mfcc_stream_walk.hip — the second kernel: two translation
//  units so that their instances compile side by side)
// mfcc_stream512_kernel — the throughput kernel of the fused MFCC pass for n_fft == 512 dialects with 13 cepstra, <= 40 filters and
// N = 2 regression deltas (the reference's sidekit call sites GMM_UBM.py:89 / d_vector.py:91 and its own utils/processing.py:110-144).
//
// Every WAVE is an independent stream: it claims a chunk (a run of consecutive frames of one utterance) from a global counter and
// walks it four frames ("quad") at a time, 16 lanes per frame, with no workgroup barrier, no global scratch and no separate delta
// pass.  Front end per quad = mfcc_fused512_kernel's (mfcc_fast.hip): LDS-DMA sample stage, pre-emphasis + window, radix-16 x
// radix-16 FFT through a swizzled LDS transpose, split step, register-resident piece filterbank, hardware log.
// Back end on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32):
//   * DCT per quad: C[ceps][frame] = DCT[ceps][filter] . LM^T[filter][frame]; the log-mel rows are the B operand straight from
//     the frame images, the DCT matrix is the A operand (KS VGPRs); the cepstra of the quad's frames go to a 24-frame ring in
//     wave-private LDS (64 B per frame)
//   * every 4th quad (16 new frames): delta = T . c and delta-delta = T . delta as banded "time" products, T[t][t'] = regression
//     weight of frame t' in delta[t] INCLUDING the reference's edge padding (GMM_UBM.py:64 pads with the first / last row, so
//     weights that fall outside [0, T) fold onto frame 0 / T-1; the A operands are generated from lane ids).  The accumulator
//     layout of one product (frames = rows in registers, cepstra = columns on lanes) is exactly the B-operand layout of the next
//     one, so c -> delta -> delta-delta chains without any data movement.  Rows [16 b - 4, 16 b + 12) x (c, delta, delta-delta)
//     leave with bounds-checked 4-byte buffer stores (13 store instructions per 16 frames).
// HBM sees every sample once and every feature once; nothing else.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "mfcc.hpp"
#include "cplx.hpp"

namespace ssp {

namespace {
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const volatile v2f* lds_cv2f_t;
typedef __attribute__((address_space(3))) const volatile float* lds_cvf_t;
constexpr int ZROW = 128;          // bytes per 16-complex row of a frame's transpose image (chunks XOR-swizzled, see mfcc_fast.hip)
constexpr int ZFRAME = 16 * ZROW;  // 2048 B per frame
constexpr int LM_OFF = 1792;       // log-mel row of frame g sits at LM_OFF - 64 g inside its image (behind the P row)
constexpr int RING_FRAMES = 24;    // cepstrum ring: the 16 newest frames + 8 of history (delta-delta reaches back 4 + 4)
constexpr int RING_ROW = 64;       // bytes per ring row: 16 cepstral slots (13 used)
constexpr int STREAM_WAVES = 4;
// P row (257 power / magnitude bins) of frame g starts PSH bytes into its image for odd g: the images are 512 dwords apart, so the four
// frames' P rows would sit on the same banks and the 16-lane groups of two frames that share a ds_write_b32 half-wave conflict 2-way on
// every P write (bank = dword mod 32); 16 dwords apart they take the two halves of the bank row.  (SSP_S_PSHIFT=0: the layout of rounds 2-5)
#ifndef SSP_S_PSHIFT
#define SSP_S_PSHIFT 64
#endif
constexpr int PSH = SSP_S_PSHIFT;
// (a piece read starts at a filter's first bin rounded down to 4 and spans at most 4 MELV <= 20 taps: it ends below dword 256 + 24 of the P
//  row — with the shift still far below the lowest log-mel row, LM_OFF - 192, which a read must never reach: last quad's ln 0 = -inf there)
static_assert(4 * (256 + 24) + PSH <= LM_OFF - 192, "shifted P row: piece reads must stay below the log-mel rows");
#define SSP_STR_(x) #x
#define SSP_STR(x) SSP_STR_(x)

__device__ __forceinline__ float stream_log(const FastArgs& f, float v) {
    return __builtin_amdgcn_logf(fmaxf(v + f.log_add, f.log_max)) * f.log_k;
}

// weight of frame tp in delta[t] for the N = 2 regression with edge replication (GMM_UBM.py:53-69): sum over u in [t-2, t+2] with
// clamp(u, 0, T-1) == tp of (u - t) * inv_denom;  half_inv = inv_denom / 2.  All quantities are small integers held in floats.
__device__ __forceinline__ float delta_weight(float t, float tp, float Tm1, float half_inv) {
    const float lo = tp <= 0.f ? -1.0e6f : tp;
    const float hi = tp >= Tm1 ? 1.0e6f : tp;
    const float a = fmaxf(lo, t - 2.f), b = fminf(hi, t + 2.f);
    const float cnt = fmaxf(b - a + 1.f, 0.f);
    const float w = cnt * (a + b - 2.f * t) * half_inv;
    return (tp < 0.f || tp > Tm1) ? 0.f : w;
}
}  // namespace

// NZ / POWER / PRE / MELV as in mfcc_fused512_kernel; KS = 4-filter k-steps of the DCT product (n_filt <= 4 KS); NS = DPP scan
// steps of the piece filterbank (a filter's pieces span <= 2^NS lanes)
// OCC = waves per SIMD the register budget is cut for: 3 (168 VGPRs, 52 KiB of LDS per workgroup) for the hop-160 dialects; 2 (256
// VGPRs, every twiddle resident) where the sample stage of a longer hop or a wider DCT operand does not fit three workgroups per CU
// CM: per-utterance mean / variance scaling (sklearn.preprocessing.scale, GMM_UBM.py:93) inside the kernel, for batches whose utterances
// are all single chunks: the wave sums x and x^2 of every column it stores (float64), then re-reads its own rows and rewrites them
// WALK: the THIRD kernel of a launch (mfcc_stream_walk.hip).  It walks the chunks the scan kernel flagged (a non-finite cepstrum in a time
// step's window: a digitally silent frame, a NaN sample) once more, sequentially, with every step formed term by term as the reference
// forms it, and rewrites every row of them; nothing flagged — the normal case — costs one load per workgroup.
template <int NZ, int POWER, int PRE, int MELV, int KS, int NS, int OCC, int CM, int WALK>
__global__ __launch_bounds__(64 * STREAM_WAVES, OCC) void mfcc_stream512_kernel(MfccArgs a, FastArgs f, StreamArgs sa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (WALK && sa.work_counter[1] == 0) return;  // nothing was flagged
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    constexpr int nc = 13;
    constexpr int mel_ns = NS;

    char* zbuf = smem + wave * sa.wave_bytes;            // 4 frame images
    float* stage = reinterpret_cast<float*>(zbuf + 4 * ZFRAME);
    char* ring = zbuf + 4 * ZFRAME + sa.stage_bytes;     // [RING_FRAMES][16] floats

    // ---- lane-resident tables (as mfcc_fast.hip): window taps, twiddles, piece filterbank
    v2f wreg[NZ];
#pragma unroll
    for (int n1 = 0; n1 < NZ; ++n1) wreg[n1] = *reinterpret_cast<const v2f*>(a.window + 32 * n1 + 2 * j);
#pragma unroll
    for (int n1 = 0; n1 < NZ; ++n1) wreg[n1] = edge_row_taps(wreg[n1], 32 * n1 + 2 * j, a.win_len);  // (see win_rows)
    // Register diet (168 VGPRs = three waves per SIMD, and 52 KiB of LDS per workgroup = three workgroups per CU, leave no room for
    // tables anywhere else): twiddles W_256^(k1 j) are resident for k1 <= NTW and W^(k1 j) = W^((k1 - 8) j) W^(8 j) above; split twiddles
    // W_512^(j + 16 i) resident for i < NWP and times W_8 above; that pays for the DCT matrix as resident MFMA A operand
    // (lane (ceps = j, kq = g), k-step s <-> filter KS g + s)
#ifndef SSP_STREAM_NTW
#define SSP_STREAM_NTW 15  // resident twiddles W_256^(k1 j), k1 <= NTW (all of them since the unit is compiled without the SLP vectorizer — build.py —; 12 before: the largest sets that leave no spill inside the loop; a spill
#endif                     // reload there waits on vmcnt behind the sample DMA and exposes its whole latency every quad)
#ifndef SSP_STREAM_NTW_CM
#define SSP_STREAM_NTW_CM 12  // ... of the instances that also carry the column sums of the scaling (CM); 9 with the SLP vectorizer
#endif
#ifndef SSP_STREAM_NWP
#define SSP_STREAM_NWP 8   // resident split twiddles
#endif
    constexpr int NTW = OCC >= 3 ? (CM ? SSP_STREAM_NTW_CM : ((MELV <= 3 && NS <= 2) ? SSP_STREAM_NTW : ((MELV >= 4 && NS >= 4) ? 8 : 9))) : 15, NWP = OCC >= 3 ? SSP_STREAM_NWP : 8;
    static_assert(NTW >= 8, "rows above NTW take W^((k1 - 8) j) W^(8 j): W^(8 j) = twr[7] must be resident");
    v2f twr[NTW], wpr[NWP];
#pragma unroll
    for (int k1 = 1; k1 <= NTW; ++k1) twr[k1 - 1] = *reinterpret_cast<const v2f*>(&f.tw16[k1 * 16 + j]);
    float dA[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) dA[s] = sa.dctA[s * 64 + lane];
#pragma unroll
    for (int i = 0; i < NWP; ++i) wpr[i] = *reinterpret_cast<const v2f*>(&f.wpost[i * 16 + j]);
    // MELV == 0: DENSE filterbank rows (the Bark bands of the PLP front end: every band has a weight on every bin) with an identity
    // "DCT".  Lane (c = lane & 15, b = lane >> 4) holds the weights of bands 6 b .. 6 b + 5 on bins 16 c .. 16 c + 15 (and on bin 256)
    // in registers; the log band energies of a quad leave as two coalesced stores, no ring, no time steps.
    constexpr bool DENSE = MELV == 0;
    constexpr int MV = DENSE ? 1 : MELV;
    v4f mw[MV];
    int mofs[MV];
    v2f mk01 = v2f{0.f, 0.f}, mk23 = v2f{0.f, 0.f};
    int mfid = -1;
    v4f dw[DENSE ? 6 : 1][4];
    float dw256[DENSE ? 6 : 1];
    if (DENSE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dw[DENSE ? k : 0][i] = *reinterpret_cast<const v4f*>(sa.dense_w + ((size_t)lane * 6 + k) * 20 + 4 * i);
            dw256[DENSE ? k : 0] = sa.dense_w[((size_t)lane * 6 + k) * 20 + 16];
        }
    } else {
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            mw[i] = *reinterpret_cast<const v4f*>(f.pc_w + ((size_t)lane * MV + i) * 4);
            mofs[i] = f.pc_ofs[lane * MV + i];
        }
        mk01 = *reinterpret_cast<const v2f*>(f.pc_mask + lane * 4);
        mk23 = NS > 2 ? *reinterpret_cast<const v2f*>(f.pc_mask + lane * 4 + 2) : v2f{0.f, 0.f};
        mfid = f.pc_fid[lane];
    }

    const int hop = a.hop;
    const float pre = PRE ? a.preemph : 0.f;
    const float npre = -pre;
    const int n_piece = (f.slen + 255) >> 8;
    const bool has_half = (f.slen & 255) != 0 && (f.slen & 255) <= 128;
    const int n_full = has_half ? n_piece - 1 : n_piece;
    const int dord = a.delta_order;
    const int Dd = a.d_out;
    const float half_inv = 0.5f * a.delta_inv_denom;
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)stage);

#ifdef SSP_S_CLOCK  // diagnostic build: shader clock (s_memtime) against the 100 MHz constant clock (s_memrealtime) over the kernel's life
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // second kernel: the waves share the chunks' flags in blocks of 64 (no atomics: one flag per lane, a ballot, the set bits in turn)
    uint64_t wm = 0;
    int wblk = blockIdx.x * STREAM_WAVES + wave, wbase = 0;
    for (;;) {
        int cidx = 0;
        if constexpr (WALK) {
            bool done = false;
            while (wm == 0) {
                if ((int64_t)wblk * 64 >= sa.n_chunks) {
                    done = true;
                    break;
                }
                const int idx = wblk * 64 + lane;
                const int fl = idx < sa.n_chunks ? sa.redo_flags[idx] : 0;
                wm = __builtin_amdgcn_ballot_w64(fl != 0);
                wbase = wblk * 64;
                wblk += gridDim.x * STREAM_WAVES;
            }
            if (done) break;
            cidx = __builtin_amdgcn_readfirstlane(wbase + __builtin_ctzll(wm));
            wm &= wm - 1;
        } else {
            if (lane == 0) cidx = atomicAdd(sa.work_counter, 1);
            cidx = __builtin_amdgcn_readfirstlane(cidx);
            if (cidx >= sa.n_chunks) break;
        }
        const MfccChunk ch = a.chunks[cidx];
        const int64_t s0 = a.sample_off[ch.utt];
        const int64_t N = a.sample_off[ch.utt + 1] - s0;
        const int64_t f0 = a.frame_off[ch.utt];
        const int T = __builtin_amdgcn_readfirstlane((int)(a.frame_off[ch.utt + 1] - f0));
        const int t0 = __builtin_amdgcn_readfirstlane(ch.t0), n = __builtin_amdgcn_readfirstlane(ch.n);
        // halo: 4 frames either side (a multiple of 4 for every delta order: a frame meets the same 4-frame k-groups of the delta product
        // wherever its chunk starts).  The delta-delta product's k-groups are STRIDED over the 16-row step window, so its summation
        // order — the last bits — depends on where the step windows sit in the utterance: chunks cut with the same rule agree bit for
        // bit, a chunk that starts at t0 = 16 m with the plain halo (windows at 8 mod 16) agrees with the uncut utterance (windows at
        // 12 mod 16) only to rounding.  ch.pad = 12 extra frames in front (ta = t0 - 16) puts the windows where the uncut utterance
        // has them: such a chunk reproduces the uncut bits (the work table's tail split uses it).
        const int H = dord > 0 ? 4 : 0;
        const int Hlo = dord > 0 ? H + __builtin_amdgcn_readfirstlane(ch.pad) : 0;
        const int ta = max(t0 - Hlo, 0), tb = min(t0 + n + H, T);
        const int R = tb - ta;                     // frames computed (relative index r = t - ta)
        const int nquads = (R + 3) >> 2;
        const int E = t0 + n - ta;                 // emitted frames end (relative)
        const int n_steps = (E + 4 + 15) >> 4;     // step b emits rows [16 b - 4, 16 b + 12)
        const int Q = DENSE ? nquads : 4 * n_steps;
        const float Tm1 = (float)(T - 1);

        const uint64_t xaddr = reinterpret_cast<uint64_t>(a.samples + s0);
        const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xaddr), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xaddr >> 32));
        const int xbytes = __builtin_amdgcn_readfirstlane((int)(N * 4));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>(((uint64_t)xhi << 32) | xlo), 0, xbytes, 0x00020000);
        const uint64_t oaddr = reinterpret_cast<uint64_t>(a.out + (size_t)f0 * Dd);
        const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)oaddr), ohi = __builtin_amdgcn_readfirstlane((uint32_t)(oaddr >> 32));
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>(((uint64_t)ohi << 32) | olo), 0, __builtin_amdgcn_readfirstlane(T * Dd * 4), 0x00020000);

        // sample DMA of quad q: slen floats from (ta + 4 q) hop, 1-KiB pieces (instruction offsets advance the global and the
        // LDS address together), a trailing half piece on lanes 0..31; outside [0, N) reads as zero
        auto prefetch = [&](int q) {
            const int vo = (ta + 4 * q) * hop * 4 + lane * 16;
            const lds_ptr_t lp = (lds_ptr_t)(uintptr_t)stage_lds;
            if (n_full > 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lp, 16, vo, 0, 0, 0);
            if (n_full > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lp, 16, vo, 0, 1024, 0);
            if (n_full > 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lp, 16, vo, 0, 2048, 0);
            if (n_full > 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lp, 16, vo, 0, 3072, 0);
            if (n_full > 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(stage_lds + 4096), 16, vo + 4096, 0, 0, 0);
            if (has_half && lane < 32)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(stage_lds + n_full * 1024), 16, vo + n_full * 1024, 0, 0, 0);
        };

        // zero the cepstrum ring: frames before the chunk's first one must read as finite values (their weights are zero)
        {
            float zz = 0.f;
            asm volatile("" : "+v"(zz));  // (opaque: a zero quad hoisted out of the chunk loop stays live across it — four registers the scaling instance spilled)
            v4f z4 = v4f{zz, zz, zz, zz};
            *reinterpret_cast<v4f*>(ring + lane * 16) = z4;
            if (lane < (RING_FRAMES * RING_ROW - 1024) / 16) *reinterpret_cast<v4f*>(ring + 1024 + lane * 16) = z4;
        }
        prefetch(0);
        int stores_pending = 0;  // buffer stores issued behind the DMA that is waited for at the top of the next iteration
        // Non-finite cepstra (a digitally silent frame: ln 0 = -inf in the dialects without a floor, GMM_UBM.py:89 / d_vector.py:96-98; a NaN
        // sample): the time products spread one over the whole 16-row step (0 . inf = NaN), where the reference's delta reaches +-2 / +-4
        // frames.  THIS kernel does nothing about it — it stores what it has.  mfcc_stream_scan_kernel (mfcc_stream.hip) reads the pollution off the
        // stored rows afterwards and flags such chunks, and the kernel's WALK = 1 instance walks flagged chunks once more, every step
        // formed term by term as the reference forms it, and rewrites their rows.  Nothing of the rare case may sit in this kernel: a
        // cold block inside the quad loop costs the loop its registers (16 to 50 resident registers spilled and reloaded per quad for a
        // block behind the products; round 4's check in front of the step cost 3 to 6 %), a block behind the loop 2.7 %, and even one
        // sticky instruction per time step + one flag store per chunk put the kernel into its slow state (2.4 % on the headline, 4.8 %
        // on the in-repo 8 kHz instance: DESIGN 4.1a) — the instruction stream without either is the fast one.
        int cm_nf = WALK;  // CM, second kernel: no column sums were kept, and entries may be NaN: the scaling pass counts (nanmean / nanstd)
        // CM: sums of this lane's stored values per block (column = lane & 15), fp32: a lane adds ~T / 4 terms, and the cepstra are
        // summed relative to a pivot — the utterance's first frame — so that var = E[(x - p)^2] - E[x - p]^2 does not cancel when a
        // column's mean is large against its spread (delta / delta-delta columns have no mean to speak of: pivot 0).  The four lane
        // groups meet in float64 at the end.  (float64 sums cost 12 VGPRs and the third wave per SIMD; these cost 7.)
        float cs1[CM ? 3 : 1] = {}, cs2[CM ? 3 : 1] = {};
        float piv = 0.f;

        // ---- the phases of one quad (lambdas: the dense-band instance runs them in sequence, every other instance software-pipelined)
        // (z / pf / pm are declared inside the loop bodies: declared out here they would be loop-carried through the wave-uniform branches
        //  of the pipelined loop and stay live — 32 + 52 registers — across the back phases)
        typedef v2f zarr_t[16];
        typedef v2f pfarr_t[NZ];
        typedef float pmarr_t[PRE ? NZ : 1];  // x[e - 1], the pre-emphasis partner of the pair (x[e], x[e + 1])
        constexpr int NH = (NZ + 1) / 2;
        // the quad's DMA has landed; the stores of a preceding step were issued behind it and may still be in flight
        auto wait_dma = [&]() {
#ifndef SSP_S_NOWAIT  // (ablation, wrong results: what the wait for the sample DMA costs)
            if (stores_pending == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (stores_pending == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (stores_pending == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (stores_pending == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (stores_pending == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
#endif
            stores_pending = 0;
        };
        // the stage comes to registers in two halves (the first is windowed into z while the second is in flight: all of it at once is
        // the register peak of the kernel)
        auto stage_read_a = [&](pfarr_t& pf, pmarr_t& pm) {
            const uint32_t sp = stage_lds + (g * hop + 2 * j) * 4;  // (LDS byte address: a volatile access through a generic pointer would be a flat load)
#pragma unroll
            for (int n1 = 0; n1 < NH; ++n1) {
                // (volatile: one ds_read_b64 + one ds_read_b32 per row, 2 + 2 LDS cycles.  Left to itself the compiler merges the pair
                //  with its pre-emphasis partner into a ds_read2_b64 — 8 LDS cycles — and spends an address VGPR + add per row on it)
                pf[n1] = *(lds_cv2f_t)(uintptr_t)(sp + 128 * n1);
#ifdef SSP_S_PM64   // (experiment: the partner through a ds_read_b64 of (x[e - 2], x[e - 1]) — bank modulus 64, conflict free — instead of the 2-way conflicting b32)
                if (PRE) pm[n1] = (*(lds_cv2f_t)(uintptr_t)(sp + 128 * n1 - 8)).y;
#else
                if (PRE) pm[n1] = *(lds_cvf_t)(uintptr_t)(sp + 128 * n1 - 4);
#endif
            }
        };
        auto stage_read_b = [&](pfarr_t& pf, pmarr_t& pm) {
            const uint32_t sp = stage_lds + (g * hop + 2 * j) * 4;
#pragma unroll
            for (int n1 = NH; n1 < NZ; ++n1) {
                pf[n1] = *(lds_cv2f_t)(uintptr_t)(sp + 128 * n1);
#ifdef SSP_S_PM64
                if (PRE) pm[n1] = (*(lds_cv2f_t)(uintptr_t)(sp + 128 * n1 - 8)).y;
#else
                if (PRE) pm[n1] = *(lds_cvf_t)(uintptr_t)(sp + 128 * n1 - 4);
#endif
            }
        };
        // A row that runs past the window's last tap carries zero weights there, and whatever the samples behind the frame hold — a NaN of a
        // corrupt recording — must not get into this frame (in the reference's arithmetic it does not).  The quad loop takes the plain
        // product all the same: the leak makes the frame's cepstra NaN, the time step's check sees that and flags the chunk, and the
        // second walk (EDGE: the legacy product, 0 . x = 0 whatever x is, on EVERY row — windows with zero padding in more rows than one
        // included) forms the frame as the reference does.  Window taps that are exactly zero INSIDE the window (numpy.hanning's ends) let
        // a NaN sample through in numpy: edge_row_taps gave them the smallest normal number, which the legacy product does not silence.
        auto win_rows = [&](zarr_t& z, pfarr_t& pf, pmarr_t& pm, auto lo_tag, auto hi_tag, auto edge_tag) {
            constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
            constexpr int EDGE = decltype(edge_tag)::value;  // 0 plain | 1 legacy product on every row | 2 on the last row (the dense-band instance,
                                                             // which has no second walk: one 32-sample row of padding is what its windows have)
#pragma unroll
            for (int n1 = LO; n1 < HI; ++n1) {
                if (n1 < NZ) {
                    v2f y = pf[n1 < NZ ? n1 : 0];
                    if (PRE) {
                        const float xm1 = pm[n1 < NZ ? n1 : 0], x0 = y.x, x1 = y.y;
                        y = v2f{__builtin_fmaf(npre, xm1, x0), __builtin_fmaf(npre, x0, x1)};
                    }
                    z[n1] = (EDGE == 1 || (EDGE == 2 && n1 == NZ - 1)) ? wmul_edge(y, wreg[n1 < NZ ? n1 : 0]) : y * wreg[n1 < NZ ? n1 : 0];
                } else {
                    z[n1] = v2f{0.f, 0.f};
                }
            }
        };
        auto window_a = [&](zarr_t& z, pfarr_t& pf, pmarr_t& pm, auto edge_tag) {
            if (PRE) pm[0] = (j == 0) ? pf[0].x : pm[0];     // y[0] = x[0] - a x[0]
            win_rows(z, pf, pm, std::integral_constant<int, 0>{}, std::integral_constant<int, NH>{}, edge_tag);
        };
        auto window_b = [&](zarr_t& z, pfarr_t& pf, pmarr_t& pm, auto edge_tag) {
            win_rows(z, pf, pm, std::integral_constant<int, NH>{}, std::integral_constant<int, 16>{}, edge_tag);
        };
        // FFT16 over n1, twiddle W_256^(n2 k1), transpose through LDS, FFT16 over n2, split step -> P row of the frame's image
        auto fft_front = [&](zarr_t& z) {
            fft16_in<(NZ <= 13)>(z);  // a 400-sample window leaves rows 13..15 of the 16 x 32 sample matrix zero
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) {
                if (k1 <= NTW) {
                    z[k1] = cmul(z[k1], twr[(k1 - 1) % NTW]);
                } else {  // W^(k1 j) = W^((k1 - 8) j) W^(8 j)
                    z[k1] = cmul(cmul(z[k1], twr[(k1 - 9) % NTW]), twr[7]);
                }
            }
            // ---- transpose through LDS (rows of 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7)
            char* zf = zbuf + g * ZFRAME;
            {
                int wb0 = ((j >> 1) << 4) | ((j & 1) << 3);
                asm volatile("" : "+v"(wb0));
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    char* wp = zf + (wb0 ^ (m << 4));
                    *reinterpret_cast<v2f*>(wp + (2 * m) * ZROW) = z[2 * m];
                    *reinterpret_cast<v2f*>(wp + (2 * m + 1) * ZROW) = z[2 * m + 1];
                }
                int rb0 = j * ZROW + (((j >> 1) & 7) << 4);
                asm volatile("" : "+v"(rb0));
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const v4f r = *reinterpret_cast<const v4f*>(zf + (rb0 ^ (c << 4)));
                    z[2 * c] = v2f{r.x, r.y};
                    z[2 * c + 1] = v2f{r.z, r.w};
                }
            }
            // ---- FFT16 over n2: lane j = k1, register = k2
            fft16(z);
            // ---- split step of the real FFT (partners from lane 16 - j), power / magnitude -> P row
            {
                float* P = reinterpret_cast<float*>(zf + (PSH ? (g & 1) * PSH : 0));
                float* Pm = P + 144 - j;
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
                    // two bin pairs (k2 = 2 kp, 2 kp + 1) per trip: each stream's two values leave with one ds_write2_b32
                    v2f e[2], d[2], w[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k2 = 2 * kp + u;
                        // partner Z[256 - k] from lane 16 - j (lane 0: its own register 16 - k2): row_mirror, then row_shr:1 with `old`
                        const float sx = z[15 - k2].x, sy = z[15 - k2].y;
                        const v2f own = z[(16 - k2) & 15];
                        const float ox = own.x, oy = own.y;
                        float mx = __builtin_amdgcn_update_dpp(sx, sx, 0x140 /*row_mirror*/, 0xF, 0xF, true);
                        float my = __builtin_amdgcn_update_dpp(sy, sy, 0x140 /*row_mirror*/, 0xF, 0xF, true);
                        mx = __builtin_amdgcn_update_dpp(ox, mx, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
                        my = __builtin_amdgcn_update_dpp(oy, my, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
                        const v2f zmk = v2f{mx, my};
                        const v2f zk = z[k2];
                        w[u] = wpr[k2 < NWP ? k2 : k2 - 4];
                        if (k2 >= NWP) w[u] = cmulc(w[u], 0.70710678118654752f, -0.70710678118654752f);  // W_512^64 = W_8
                        e[u] = __builtin_elementwise_fma(zmk, v2f{1.f, -1.f}, zk);
                        d[u] = __builtin_elementwise_fma(zmk, v2f{-1.f, 1.f}, zk);
                    }
                    d[0] = cmul_negi(d[0], w[0]);  // o = (-i d) w
                    d[1] = cmul_negi(d[1], w[1]);
                    float pa[2], pb[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const v2f o = d[u];
                        const v2f Rr = __builtin_elementwise_fma(xx(o), v2f{1.f, -1.f}, xx(e[u]));
                        const v2f Ii = __builtin_elementwise_fma(yy(o), v2f{1.f, -1.f}, yy(e[u]));
                        const v2f pw = __builtin_elementwise_fma(Rr, Rr, Ii * Ii);
                        pa[u] = pw.x;
                        pb[u] = pw.y;
                        if (POWER == 1) {  // v_sqrt_f32 (1 ulp), not sqrtf(): the correctly rounded expansion is ~18 instructions per bin
                            pa[u] = __builtin_amdgcn_sqrtf(pa[u]);
                            pb[u] = __builtin_amdgcn_sqrtf(pb[u]);
                        }
                    }
                    const int k2 = 2 * kp;
                    P[j + 16 * k2] = pa[0];
                    P[j + 16 * (k2 + 1)] = pa[1];
                    Pm[16 * (6 - k2)] = pb[1];
                    Pm[16 * (7 - k2)] = pb[0];
                }
                const v2f s8 = z[8] * z[8];
                float p128 = 4.f * (s8.x + s8.y);
                if (POWER == 1) p128 = __builtin_amdgcn_sqrtf(p128);
                if (j == 0) P[128] = p128;
            }
        };
        // ---- piece filterbank + log of the four frames whose P rows are in the images: all 64 lanes on one frame at a time
        //      (see mfcc_fast.hip step 7); the log-mel rows go behind the P rows
        auto mel = [&]() {
#ifndef SSP_S_NOMEL
            if (!CM) {
                float* lm = reinterpret_cast<float*>(zbuf + g * ZFRAME + LM_OFF - 64 * g);
                if (j < f.lm_pad) lm[a.n_filt + j] = 0.f;
            } else if (j < f.lm_pad) {  // (scaling instances: the address from an opaque copy of the lane id — hoisted out of the loop it costs the register that spills)
                int ol = lane;
                asm volatile("" : "+v"(ol));
                reinterpret_cast<float*>(zbuf + (ol >> 4) * (ZFRAME - 64) + LM_OFF)[a.n_filt + (ol & 15)] = 0.f;
            }
            float sfr[4];
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) {
                const char* pr = zbuf + fr * ZFRAME + (fr & 1) * PSH;
                v4f acc = *reinterpret_cast<const v4f*>(pr + mofs[0]) * mw[0];
#pragma unroll
                for (int i = 1; i < MV; ++i) acc = __builtin_elementwise_fma(*reinterpret_cast<const v4f*>(pr + mofs[i]), mw[i], acc);
                const v2f h = v2f{acc.x, acc.y} + v2f{acc.z, acc.w};
                sfr[fr] = h.x + h.y;
            }
            v2f s01 = v2f{sfr[0], sfr[1]}, s23 = v2f{sfr[2], sfr[3]};
#define SSP_SCAN_STEP(CTRL, MK)                                                                     \
            {                                                                                           \
                const float a0 = s01.x, a1 = s01.y, a2 = s23.x, a3 = s23.y;                             \
                const float b0 = __builtin_amdgcn_update_dpp(a0, a0, CTRL, 0xF, 0xF, true);           \
                const float b1 = __builtin_amdgcn_update_dpp(a1, a1, CTRL, 0xF, 0xF, true);           \
                const float b2 = __builtin_amdgcn_update_dpp(a2, a2, CTRL, 0xF, 0xF, true);           \
                const float b3 = __builtin_amdgcn_update_dpp(a3, a3, CTRL, 0xF, 0xF, true);           \
                s01 = __builtin_elementwise_fma(v2f{b0, b1}, MK, s01);                                  \
                s23 = __builtin_elementwise_fma(v2f{b2, b3}, MK, s23);                                  \
            }
            if (mel_ns > 0) SSP_SCAN_STEP(0x101 /*row_shl:1*/, xx(mk01))
            if (mel_ns > 1) SSP_SCAN_STEP(0x102 /*row_shl:2*/, yy(mk01))
            if (mel_ns > 2) SSP_SCAN_STEP(0x104 /*row_shl:4*/, xx(mk23))
            if (mel_ns > 3) SSP_SCAN_STEP(0x108 /*row_shl:8*/, yy(mk23))
#undef SSP_SCAN_STEP
            if (mfid >= 0) {
                float* lmf = reinterpret_cast<float*>(zbuf + LM_OFF) + mfid;
                lmf[0 * (ZFRAME - 64) / 4] = stream_log(f, s01.x);
                lmf[1 * (ZFRAME - 64) / 4] = stream_log(f, s01.y);
                lmf[2 * (ZFRAME - 64) / 4] = stream_log(f, s23.x);
                lmf[3 * (ZFRAME - 64) / 4] = stream_log(f, s23.y);
            }
#endif
        };
        // ---- DCT on the matrix cores: C[ceps][frame] over the four frames whose log-mel rows are in the images (columns 4..15 repeat them)
        auto dct_mfma = [&]() -> v4f {
            v4f cq = v4f{0.f, 0.f, 0.f, 0.f};
#ifndef SSP_S_NODCT
            // (lane-derived addresses are recomputed from an opaque copy of the lane id: hoisted out of the loop they would
            //  sit in registers the FFT phases need)
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
            const char* lmrow = zbuf + (j & 3) * (ZFRAME - 64) + LM_OFF + g * (KS * 4);
            float lb[KS];
#pragma unroll
            for (int s = 0; s < KS; s += 2) {
                const v2f v = *reinterpret_cast<const v2f*>(lmrow + 4 * s);
                lb[s] = v.x;
                lb[s + 1] = v.y;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) cq = __builtin_amdgcn_mfma_f32_16x16x4f32(dA[s], lb[s], cq, 0, 0, 0);
#endif
            return cq;
        };
        // cepstra of quad qq -> ring: lane (g, j): cepstra 4 g .. 4 g + 3 of frame ta + 4 qq + j (j < 4); frames past the chunk's last one
        // and virtual quads behind it store zeros (their ring rows must read as finite values)
        auto ring_put = [&](int qq, v4f cq, bool real) {
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
            if (j < 4) {
                const bool ok = real && ta + 4 * qq + j < tb;
                const v4f cv = ok ? cq : v4f{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<v4f*>(ring + ((((qq + 6) % 6) * 4 + j) * RING_ROW) + g * 16) = cv;
            }
        };
        // ================= time step b: rows [16 b - 4, 16 b + 12) of (c, delta, delta-delta) leave =================
        typedef float cbarr_t[6];
        // B operands of a step: cepstra of ring frames rb - 8 + 4 s + g, column j (lane (g, j)), s = 0..5
        auto ring_window = [&](int b, cbarr_t& cb) {
            const int rb = 16 * b;
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int m = (rb + 16 + 4 * s) % RING_FRAMES;  // (rb - 8 + 4 s) mod 24, a multiple of 4
                cb[s] = *reinterpret_cast<const float*>(ring + (m + g) * RING_ROW + j * 4);
            }
        };
        auto time_step = [&](int b, const cbarr_t& cb) {
            const int rb = 16 * b;
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
            const float tg = (float)(ta + rb);  // utterance frame index of relative frame rb
            // A operands = regression weights of frame (tg + tpr) in delta[tg + tr].  Steps whose 24-frame window lies strictly
            // inside the utterance (all but the first and the last one or two) take them from the lane-constant distance
            // d = tpr - tr: d / denom for |d| <= 2; at the utterance ends the edge-replicated form folds the outside weights
            // onto frame 0 / T - 1.
            const bool interior = ta + rb - 8 >= 1 && ta + rb + 16 <= T - 2;  // wave-uniform
            const float inv = 2.f * half_inv;
            auto W = [&](float tr, float tpr) -> float {
                if (interior) {
                    const float d = tpr - tr;
                    return __builtin_fabsf(d) <= 2.f ? d * inv : 0.f;
                }
                return delta_weight(tg + tr, tg + tpr, Tm1, half_inv);
            };
            const float fj = (float)j, fg = (float)g;
            // rows leave with bounds-checked 4-byte buffer stores; a lane that has nothing to store aims out of bounds.  The
            // instruction count per step is fixed (the wait at the top of the next quad counts them)
            const int Fo = ta + rb - 4 + 4 * g;  // first output frame of this lane group (registers r = 0..3 follow)
            const bool full = ta + rb - 4 >= t0 && ta + rb + 12 <= t0 + n;  // wave-uniform: every row of the window is emitted
            const int lane_off = j < nc ? (Fo * Dd + j) * 4 : 0x7ffffff0;
            auto put = [&](int rrel, int blk, float v, bool lane_on) {
                // row Fo + rrel, block blk (0 cepstra | 1 delta | 2 delta-delta)
                int off;
                if (full) {
                    off = lane_on ? lane_off + (rrel * Dd + blk * nc) * 4 : 0x7ffffff0;
                } else {
                    const int F = Fo + rrel;
                    off = (lane_on && F >= t0 && F < t0 + n) ? lane_off + (rrel * Dd + blk * nc) * 4 : 0x7ffffff0;
                }
#ifdef SSP_S_NOSTORE  // ablation: the products stay live, nothing leaves
                asm volatile("" ::"v"(v), "v"(off));
#else
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off, 0, 0);
#endif
                if (CM) {
                    const float dv = off != 0x7ffffff0 ? (blk == 0 ? v - piv : v) : 0.f;
                    cs1[CM ? blk : 0] += dv;
                    cs2[CM ? blk : 0] = __builtin_fmaf(dv, dv, cs2[CM ? blk : 0]);
                }
            };
            if (CM && b == 0) piv = j < nc ? *reinterpret_cast<const float*>(ring + j * 4) : 0.f;  // cepstra of frame 0 (CM: ta == 0)
            // cepstra of the output rows straight from the ring
            {
                const int m4 = (rb + 20) % RING_FRAMES;  // (rb - 4) mod 24
                int slot = m4 + 4 * g;
                slot = slot >= RING_FRAMES ? slot - RING_FRAMES : slot;
                const float* cr = reinterpret_cast<const float*>(ring + slot * RING_ROW + j * 4);
                const v4f o0 = v4f{cr[0], cr[16], cr[32], cr[48]};
#pragma unroll
                for (int r = 0; r < 4; ++r) put(r, 0, o0[r], true);
            }
            if (dord >= 1) {
                // delta tile 0: rows i = j <-> frame rb - 6 + i; contraction over ring frames rb - 8 + 4 s + g, s = 0..4
                v4f d0 = v4f{0.f, 0.f, 0.f, 0.f}, d1 = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 5; ++s) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W(fj - 6.f, fg + (float)(4 * s - 8)), cb[s], d0, 0, 0, 0);
                // delta tile 1: rows i = 0, 4, 8, 12 <-> frames rb + 10 + i / 4 (the other rows are zero); s = 4, 5
                const float fr1 = (float)(j >> 2) + 10.f;
#pragma unroll
                for (int s = 4; s < 6; ++s) {
                    float w = W(fr1, fg + (float)(4 * s - 8));
                    w = (j & 3) == 0 ? w : 0.f;
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, cb[s], d1, 0, 0, 0);
                }
                // delta rows leave from their own layout: tile 0 register r <-> frame rb - 6 + 4 g + r = row Fo + r - 2 (the first two
                // belong to the previous step's window), tile 1 register 0 <-> frame rb + 10 + g = row Fo + 14 - 3 g (g < 2)
#pragma unroll
                for (int r = 0; r < 4; ++r) put(r - 2, 1, d0[r], r >= 2 || g > 0);
                put(14 - 3 * g, 1, d1[0], g < 2);
                if (dord >= 2) {
                    // delta-delta: rows i = j <-> frame rb - 4 + i; B = delta tile 0 register s (frame rb - 6 + 4 g + s) and
                    // delta tile 1 register 0 (frame rb + 10 + g)
                    v4f dd = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s) dd = __builtin_amdgcn_mfma_f32_16x16x4f32(W(fj - 4.f, 4.f * fg + (float)(s - 6)), d0[s], dd, 0, 0, 0);
                    dd = __builtin_amdgcn_mfma_f32_16x16x4f32(W(fj - 4.f, fg + 10.f), d1[0], dd, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) put(r, 2, dd[r], true);
                }
            }
            stores_pending = dord >= 2 ? 13 : (dord == 1 ? 9 : 4);
        };
        // ================= the same step in the TRANSPOSED orientation, for steps whose 24-frame window lies inside the utterance ==========
        // D^T[ceps][frame] = c^T[ceps][t'] . T^T[t'][frame]: the ring registers are the A operand as they stand (A and B share their lane
        // layout), the weights the B operand, and a lane ends up with FOUR CONSECUTIVE CEPSTRA of ONE frame — 16 contiguous bytes of
        // the output row: c, delta and delta-delta leave with one 16-byte store (cepstra 0..11) + one 4-byte store (cepstrum 12) each,
        // 6 store instructions per 16 frames instead of 13, and a frame that is not emitted is simply a lane that aims out of bounds.
        // The accumulator of one product is no longer the operand of the next (frames sit on lanes now), so delta-delta is ONE product
        // with the auto-convolved weights (reach +-4, N = 2: (-10, -4, 1, 4, 4) / denom^2 at |d| = 0..4 = -10 + |d| (37 - d^2) / 6) —
        // for interior frames the same numbers as delta(delta(c)) up to the rounding of the intermediate delta.
        auto time_step_T = [&](int b, const cbarr_t& cb) {
            const int rb = 16 * b;
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
            const float inv = 2.f * half_inv;
            const float ef = (float)(g - 4 - j);  // input frame (rb - 8 + 4 s + g) minus output frame (rb - 4 + j) = ef + 4 s
            const int F = ta + rb - 4 + j;       // this lane's output frame
            const bool emit = F >= t0 && F < t0 + n;
            const int row16 = (emit && g < 3) ? (F * Dd + 4 * g) * 4 : 0x7ffffff0;
            const int row4 = (emit && g == 3) ? (F * Dd + 12) * 4 : 0x7ffffff0;
            auto store = [&](v4f v, int blk) {
                typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), ro, row16 + blk * (nc * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), ro, row4 + blk * (nc * 4), 0, 0);
            };
            {
                // the cepstra of this lane's frame need no product (until round 5 they took four with a unit matrix): four consecutive
                // cepstra of one frame are 16 contiguous bytes of its ring row.  (A non-finite frame therefore stays in its own row of this
                // block; what the scan kernel looks at is the delta block, or every row when there are no deltas.)
                int slot = (rb + 20) % RING_FRAMES + j;  // (rb - 4 + j) mod 24
                slot = slot >= RING_FRAMES ? slot - RING_FRAMES : slot;
                const v4f c = *reinterpret_cast<const v4f*>(ring + slot * RING_ROW + g * 16);
                store(c, 0);
            }
            if (dord >= 1) {
                v4f d = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const float ds = ef + (float)(4 * s);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(cb[s], __builtin_fabsf(ds) <= 2.f ? ds * inv : 0.f, d, 0, 0, 0);
                }
                store(d, 1);
            }
            if (dord >= 2) {
                v4f q = v4f{0.f, 0.f, 0.f, 0.f};
                const float inv2 = inv * inv;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const float a = __builtin_fabsf(ef + (float)(4 * s));
                    const float w = __builtin_fmaf(a, __builtin_fmaf(a * a, -1.f / 6.f, 37.f / 6.f), -10.f) * inv2;
                    q = __builtin_amdgcn_mfma_f32_16x16x4f32(cb[s], a <= 4.f ? w : 0.f, q, 0, 0, 0);
                }
                store(q, 2);
            }
            stores_pending = 2 * (1 + dord);
        };
        // ================= the steps of a chunk that is walked again because a window held a non-finite cepstrum (rare) =================
        // GMM_UBM.py:53-69 term by term: delta[t] = sum_n n c[clamp(t + n)] / denom including the n = 0 term (0 . inf = NaN, as numpy.dot
        // has it), delta-delta = delta(delta(c)) with the clamp at both levels: exactly the frames within +-2 / +-4 of a non-finite
        // cepstrum come out non-finite, every other one from finite terms only.  Lane layout and stores of the transposed form.
        auto time_step_nf = [&](int b) {
            const int rb = 16 * b;
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
            const float inv = 2.f * half_inv;
            const int F = ta + rb - 4 + j;       // this lane's output frame; it forms cepstra 4 g .. 4 g + 3 of it, ONE AT A TIME in rolled loops:
            const bool emit = F >= t0 && F < t0 + n;  // the block is cold, and what it keeps alive at once the quad loop pays for in registers
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int col = 4 * g + k;
                const int off = (emit && col < nc) ? (F * Dd + col) * 4 : 0x7ffffff0;
                // cepstrum `col` of utterance frame clamp(t): every frame an emitted row needs is in the ring (relative frames rb - 8 .. rb + 15)
                auto cread = [&](int t) -> float {
                    t = min(max(t, 0), T - 1);
                    const unsigned slot = (unsigned)(t - ta + 2 * RING_FRAMES) % (unsigned)RING_FRAMES;
                    return *reinterpret_cast<const float*>(ring + slot * RING_ROW + col * 4);
                };
                auto dl = [&](int u) -> float {
                    u = min(max(u, 0), T - 1);
                    float s = 0.f;
#pragma unroll 1
                    for (int m = -2; m <= 2; ++m) s = __builtin_fmaf((float)m, cread(u + m), s);
                    return s * inv;
                };
                auto store = [&](float v, int blk) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off + blk * (nc * 4), 0, 0); };
                store(cread(F), 0);
                if (dord >= 1) store(dl(F), 1);
                if (dord >= 2) {
                    float q = 0.f;
#pragma unroll 1
                    for (int kk = -2; kk <= 2; ++kk) q = __builtin_fmaf((float)kk, dl(F + kk), q);
                    store(q * inv, 2);
                }
            }
            stores_pending = 0;  // (the next wait is for everything in flight)
            cm_nf = 1;
        };
        // a step whose window (frames rb - 8 .. rb + 15) lies strictly inside the utterance takes the transposed form; utterance ends
        // (edge-replicated weights) and the scaling instances (their column sums live in the row-major layout) the chained one
        auto emit_step = [&](int b) {
            cbarr_t cb;
            ring_window(b, cb);
#ifdef SSP_S_NOTSTEP
            time_step(b, cb);
#else
            const bool interior = ta + 16 * b - 8 >= 1 && ta + 16 * b + 16 <= T - 2;  // wave-uniform (the scan kernel repeats this test)
            // (not the scaling instances — their column sums live in the row-major layout — nor the widest filterbank instance, which has no
            //  register left for the second form)
            constexpr bool TSTEP = !CM && !(MELV >= 4 && NS >= 4);
            if (TSTEP && interior) time_step_T(b, cb);
            else time_step(b, cb);
#endif
        };

        if constexpr (DENSE) {
            static_assert(!(DENSE && WALK), "the dense-band instance has no time steps: nothing to walk again");
            for (int q = 0; q < nquads; ++q) {
                zarr_t z;
                pfarr_t pf;
                pmarr_t pm;
                wait_dma();
                stage_read_a(pf, pm);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                window_a(z, pf, pm, std::integral_constant<int, DENSE ? 2 : 0>{});
                stage_read_b(pf, pm);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is in registers: the next quad may overwrite it
#ifndef SSP_S_NODMA
                prefetch(q + 1);                                     // flies under this whole iteration (past the end: zeros)
#endif
                window_b(z, pf, pm, std::integral_constant<int, DENSE ? 2 : 0>{});
                fft_front(z);
                // ---- dense bands: partial sums of this lane's 17 bins for 6 bands and the quad's 4 frames ...
                int ol = lane;
                asm volatile("" : "+v"(ol));
                const int c = ol & 15;
                float r[4][6];
#pragma unroll
                for (int fr = 0; fr < 4; ++fr) {
                    const char* pr = zbuf + fr * ZFRAME + (fr & 1) * PSH;
                    const v4f p0 = *reinterpret_cast<const v4f*>(pr + 64 * c), p1 = *reinterpret_cast<const v4f*>(pr + 64 * c + 16);
                    const v4f p2 = *reinterpret_cast<const v4f*>(pr + 64 * c + 32), p3 = *reinterpret_cast<const v4f*>(pr + 64 * c + 48);
                    const float p256 = *reinterpret_cast<const float*>(pr + 1024);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        v4f acc = p0 * dw[DENSE ? k : 0][0];
                        acc = __builtin_elementwise_fma(p1, dw[DENSE ? k : 0][1], acc);
                        acc = __builtin_elementwise_fma(p2, dw[DENSE ? k : 0][2], acc);
                        acc = __builtin_elementwise_fma(p3, dw[DENSE ? k : 0][3], acc);
                        const v2f hs = v2f{acc.x, acc.y} + v2f{acc.z, acc.w};
                        r[fr][k] = __builtin_fmaf(p256, dw256[DENSE ? k : 0], hs.x + hs.y);
                    }
                }
                // ... meet through LDS ([frame][band of the group][lane] over the frame images, whose P rows are consumed): output
                // o = frame * n_bands + band sums the 16 bin-chunk lanes of its band group
                float* sc = reinterpret_cast<float*>(zbuf);
#pragma unroll
                for (int fr = 0; fr < 4; ++fr)
#pragma unroll
                    for (int k = 0; k < 6; ++k) sc[(fr * 6 + k) * 64 + ol] = r[fr][k];
                const int nb = a.n_filt;
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int o = rd * 64 + ol;
                    const int fr = (o >= nb) + (o >= 2 * nb) + (o >= 3 * nb);
                    const int band = o - fr * nb;
                    const int bg = (band * 43) >> 8, k = band - 6 * bg;  // band / 6 for band < 24
                    const bool valid = o < 4 * nb;
                    const float* src = sc + ((valid ? fr * 6 + k : 0) * 64 + (valid ? 16 * bg : 0));
                    const v4f s0 = *reinterpret_cast<const v4f*>(src), s1 = *reinterpret_cast<const v4f*>(src + 4);
                    const v4f s2 = *reinterpret_cast<const v4f*>(src + 8), s3 = *reinterpret_cast<const v4f*>(src + 12);
                    const v4f s4 = (s0 + s1) + (s2 + s3);
                    const float val = stream_log(f, (s4.x + s4.y) + (s4.z + s4.w));
                    const bool ok = valid && ta + 4 * q + fr < tb;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, ok ? ((ta + 4 * q) * Dd + o) * 4 : 0x7ffffff0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch is read: the next quad's transposes may overwrite it
                stores_pending = 2;
            }
        } else if constexpr (WALK) {
            // ---- the chunk's quads in sequence (no software pipeline), every step through time_step_nf — its term-by-term sums are the
            // reference's for finite windows as well; the legacy product on every window row (win_rows)
            for (int q = 0; q < Q; ++q) {
                if (q < nquads) {
                    zarr_t z;
                    pfarr_t pf;
                    pmarr_t pm;
                    wait_dma();
                    stage_read_a(pf, pm);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    window_a(z, pf, pm, std::integral_constant<int, 1>{});
                    stage_read_b(pf, pm);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    prefetch(q + 1);
                    window_b(z, pf, pm, std::integral_constant<int, 1>{});
                    fft_front(z);
                    mel();
                    const v4f cq = dct_mfma();
                    ring_put(q, cq, true);
                } else {
                    ring_put(q, v4f{0.f, 0.f, 0.f, 0.f}, false);
                }
                if ((q & 3) == 3) time_step_nf(q >> 2);
            }
        } else {
#ifdef SSP_S_SEQ  // (A/B: the phases of a quad in sequence, as before round 3)
            for (int q = 0; q < Q; ++q) {
                zarr_t z;
                pfarr_t pf;
                pmarr_t pm;
                if (q < nquads) {
                    wait_dma();
                    stage_read_a(pf, pm);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    window_a(z, pf, pm, std::integral_constant<int, DENSE ? 2 : 0>{});
                    stage_read_b(pf, pm);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef SSP_S_NODMA
                    prefetch(q + 1);
#endif
                    window_b(z, pf, pm, std::integral_constant<int, DENSE ? 2 : 0>{});
                    fft_front(z);
                    mel();
                    const v4f cq = dct_mfma();
                    ring_put(q, cq, true);
                } else {
                    ring_put(q, v4f{0.f, 0.f, 0.f, 0.f}, false);
                }
#ifndef SSP_S_NOSTEP
                if ((q & 3) == 3) emit_step(q >> 2);
#endif
            }
#else
            // Software-pipelined quad loop: iteration q runs the FRONT of quad q (stage -> window -> FFT -> split -> P rows) and the BACK of
            // quad q - 1 (filterbank + log -> DCT -> ring -> time step).  The back's LDS round trips ride under the front's: the P reads
            // of the filterbank are issued behind the first half of the stage reads and return with them, the log-mel reads of the DCT
            // behind the second half, and the DCT's dependent MFMA chain runs while the second half is windowed.  LDS operations of a
            // wave execute in order, and the back's reads of the images are all issued before the front's transposes overwrite them.
#ifdef SSP_S_PAD  // (diagnostic: the loop's code address shifted by SSP_S_PAD instructions)
            asm volatile(".rept " SSP_STR(SSP_S_PAD) "\n s_nop 0\n .endr");
#endif
            for (int q = 0; q < nquads; ++q) {
                zarr_t z;
                pfarr_t pf;
                pmarr_t pm;
                wait_dma();
                stage_read_a(pf, pm);
                const v4f cq = dct_mfma();  // quad q - 1 (q = 0: whatever the images hold; masked in ring_put)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                window_a(z, pf, pm, std::integral_constant<int, DENSE ? 2 : 0>{});
                stage_read_b(pf, pm);
                ring_put(q - 1, cq, q > 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is in registers: the next quad may overwrite it
#ifndef SSP_S_NODMA
                prefetch(q + 1);                                     // flies under this whole iteration (past the end: zeros)
#endif
                window_b(z, pf, pm, std::integral_constant<int, DENSE ? 2 : 0>{});
                fft_front(z);
                mel();
#ifndef SSP_S_NOSTEP
                if ((q & 3) == 0 && q > 0) emit_step((q - 1) >> 2);  // (here, where no FFT register is live)
#endif
            }
            // drain: the back of the last quad, then the virtual quads behind the chunk's last frame
            for (int qb = nquads - 1; qb < Q; ++qb) {
                v4f cq = v4f{0.f, 0.f, 0.f, 0.f};
                if (qb < nquads) cq = dct_mfma();
                ring_put(qb, cq, qb < nquads);
#ifndef SSP_S_NOSTEP
                if ((qb & 3) == 3) emit_step(qb >> 2);
#endif
            }
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last prefetch (zeros) and every store have completed
        if (CM) {
            // ---- scaling pass: column statistics over the four lane groups, then the utterance's rows once more through L2
            int ol = lane;
            asm volatile("" : "+v"(ol));
            const int g = ol >> 4, j = ol & 15;
            float* ct = reinterpret_cast<float*>(ring);  // [0, 48): means, [48, 96): 1 / std  (the ring is idle until the next chunk zeroes it)
            if (cm_nf) {
                // the utterance holds non-finite cepstra (rare): sklearn.preprocessing.scale takes the statistics over the entries that
                // are not NaN (nanmean / nanstd, sk:preprocessing/_data.py scale) and leaves the NaN entries as they are.  Lane = column;
                // the rows come back through L2 (this wave wrote them; its stores have completed)
                double n1 = 0.0, a1 = 0.0, a2 = 0.0;
                const int col = ol < Dd ? ol : 0;
                for (int F0 = 0; F0 < T; F0 += 8) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ro, F0 + k < T ? ((F0 + k) * Dd + col) * 4 : 0x7ffffff0, 0, 1));
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool ok = F0 + k < T && v[k] == v[k];
                        const double d = ok ? (double)v[k] : 0.0;
                        n1 += ok ? 1.0 : 0.0;
                        a1 += d;
                        a2 += d * d;
                    }
                }
                const double mean = a1 / n1;  // (no entry at all: NaN, as nanmean has it)
                const double var = a2 / n1 - mean * mean;
                double sd = __builtin_sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
                if (sd < 10.0 * 1.1920929e-07) sd = 1.0;
                if (ol < Dd) {
                    ct[ol] = (float)mean;
                    ct[48 + ol] = (float)(1.0 / sd);
                }
            } else
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                if (blk > dord) break;
                double a1 = (double)cs1[CM ? blk : 0], a2 = (double)cs2[CM ? blk : 0];
                // (the partner lanes through ds_bpermute on the opaque lane id: __shfl_xor's own lane id — v_mbcnt — is loop-invariant, gets
                //  hoisted out of the chunk loop and spilled there)
                auto xor_add = [&](double v, int m) -> double {
                    const v2u b = __builtin_bit_cast(v2u, v);
                    const int addr = (ol ^ m) << 2;
                    const v2u o = v2u{(unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)b.x), (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)b.y)};
                    return v + __builtin_bit_cast(double, o);
                };
                a1 = xor_add(a1, 16);
                a2 = xor_add(a2, 16);
                a1 = xor_add(a1, 32);
                a2 = xor_add(a2, 32);
                const double m0 = a1 / (double)T;                 // mean of (x - pivot)
                const double mean = m0 + (blk == 0 ? (double)piv : 0.0);
                const double var = a2 / (double)T - m0 * m0;
                double sd = __builtin_sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));  // a negative rounding residue is zero; NaN stays NaN
                if (sd < 10.0 * 1.1920929e-07) sd = 1.0;                                 // sk: _handle_zeros_in_scale (as cmvn_kernel)
                if (g == 0 && j < nc) {
                    ct[blk * nc + j] = (float)mean;
                    ct[48 + blk * nc + j] = (float)(1.0 / sd);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the table is written (LDS operations of a wave execute in order)
            // the utterance's rows once more: 16 loads in flight per lane (through L2: glc), 8 bytes per lane when rows are 8-byte
            // aligned (an even number of columns)
            const int tot = T * Dd;
            auto rewrite = [&](auto wtag) {
                constexpr int W = decltype(wtag)::value, B = 16;
                const int n_el = tot / W;
                const int step = (64 * W) % Dd;
                int c = (ol * W) % Dd;
                for (int i0 = 0; i0 < n_el; i0 += 64 * B) {
                    v2f v[B];
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        const int idx = i0 + 64 * k + ol;
                        const int off = idx < n_el ? idx * (4 * W) : 0x7ffffff0;
                        if (W == 2) v[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ro, off, 0, 1));
                        else v[k] = v2f{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ro, off, 0, 1)), 0.f};
                    }
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        const int idx = i0 + 64 * k + ol;
                        const int off = idx < n_el ? idx * (4 * W) : 0x7ffffff0;
                        const int c1 = c + 1 == Dd ? 0 : c + 1;
                        const float m0 = ct[c], is0 = ct[48 + c];
                        if (W == 2) {
                            const float m1 = ct[c1], is1 = ct[48 + c1];
                            const v2f o = v2f{(v[k].x - m0) * is0, (v[k].y - m1) * is1};
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), ro, off, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (v[k].x - m0) * is0), ro, off, 0, 0);
                        }
                        c += step;
                        c = c >= Dd ? c - Dd : c;
                    }
                }
            };
            if ((Dd & 1) == 0) rewrite(std::integral_constant<int, 2>{});
            else rewrite(std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the next chunk zeroes the ring)
        }
    }
#ifdef SSP_S_CLOCK
    if (tid == 0) {
        const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(reinterpret_cast<unsigned long long*>(sa.work_counter + 8), ck1 - ck0);
        atomicAdd(reinterpret_cast<unsigned long long*>(sa.work_counter + 10), rt1 - rt0);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ launch (first / third kernel)
// k-steps of the DCT product the instances are built for: 6 (<= 24 filters: the sidekit dialects) or 10 (<= 40: the in-repo MFCC)
static inline int stream_ks(const ssp_mfcc_cfg& c) { return (c.n_filt + 3) / 4 <= 6 ? 6 : 10; }

// WALK = 0: the first kernel of a launch (mfcc_stream.hip instantiates these); WALK = 1: the second (mfcc_stream_walk.hip), same
// instance, same grid, same arguments.  dry_run: only answers whether an instance of the kernel exists for (cfg, in-kernel scaling) — the
// work-table builder asks before it commits a batch to this kernel, so that auto mode falls back to the workgroup kernel instead of
// failing at launch
template <int WALK>
int launch_mfcc_stream_impl(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream, bool dry_run) {
    if (n_chunks <= 0 && !dry_run) return SSP_OK;
    FastArgs f = p->fast;
    const ssp_mfcc_cfg& c = p->cfg;
    StreamArgs sa{};
    const int KS = stream_ks(c);
    // padded filter slots of the log-mel rows (up to 4 KS) must read as finite zeros
    f.lm_pad = 4 * KS - c.n_filt;
    if (f.lm_pad > 16 && f.melv != 0) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream): %d filters leave more than 16 padded slots", c.n_filt);
    sa.dctA = p->s_dctA.as<float>();
    sa.dense_w = p->s_dense.as<float>();
    // the trailing half piece of the sample stage only writes 512 B
    const int n_piece = (f.slen + 255) >> 8;
    const bool has_half = (f.slen & 255) != 0 && (f.slen & 255) <= 128;
    sa.stage_bytes = has_half ? (n_piece - 1) * 1024 + 512 : n_piece * 1024;
    sa.wave_bytes = 4 * ZFRAME + sa.stage_bytes + RING_FRAMES * RING_ROW;
    sa.table_bytes = 0;
    sa.n_chunks = n_chunks;
    if (!dry_run) {
        // one buffer: a 64-byte head ([0] next chunk to claim, [1] a chunk was flagged) + one flag per chunk, all zeroed by the first
        // launch; the scan kernel sets what it finds
        SSP_TRY(p->f_counter.reserve(64 + (size_t)std::max(n_chunks, 1) * sizeof(int32_t)));
        sa.work_counter = p->f_counter.as<int32_t>();
        sa.redo_flags = sa.work_counter + 16;
    }
    size_t lds = (size_t)sa.table_bytes + (size_t)STREAM_WAVES * sa.wave_bytes;
    if (const char* e = getenv("SSP_MFCC_LDS_PAD")) lds = std::min<size_t>(160 * 1024, lds + (size_t)atoi(e));  // diagnostic: caps the workgroups per CU
    if (lds > 160 * 1024) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream): LDS footprint %zu B exceeds 160 KiB", lds);
    if ((int64_t)p->fast_max_samples * 4 > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream): utterance too long for 32-bit offsets");
    const int nz = c.win_len <= 416 ? 13 : 16, pw = c.spec_power, pr = c.preemph_mode ? 1 : 0;
    // three workgroups per CU (52 KiB each, 168 VGPRs) when the stage and the operands allow it, two otherwise
    // (measured on the in-repo dialect, 59 KiB per workgroup: 2 x 4 waves per CU with every twiddle resident 8.2 ms; 168-VGPR instances
    //  in 1- / 2- / 3-wave workgroups, 11 / 10 / 9 waves per CU, 9.1 - 9.4 ms)
    const int cm = args.cmvn != 0 ? 1 : 0;  // (the plan only leaves cmvn set when mfcc_stream_fuses_cmvn and every utterance is one chunk)
    // (the 102 weights per lane of the dense-band instance need the registers of two waves per SIMD)
    // (scaling instances keep three waves per SIMD only where the seven extra registers of the column sums fit without a spill)
    const int occ = (nz == 13 && KS == 6 && lds <= 53248 && f.melv != 0 && (!cm || (f.melv <= 3 && f.mel_ns <= 2))) ? 3 : 2;
    const int wg_waves = STREAM_WAVES;
    bool launched = false;
#define SSP_STREAM_CASE(NZ_, PW_, PR_, MV_, KS_, OCC_) SSP_STREAM_CASE_CM(NZ_, PW_, PR_, MV_, KS_, OCC_, 0)
#define SSP_STREAM_CASE_CM(NZ_, PW_, PR_, MV_, KS_, OCC_, CM_)                                                        \
    if (!launched && dry_run && nz == NZ_ && pw == PW_ && pr == PR_ && f.melv == MV_ && KS == KS_ && occ == OCC_ && cm == CM_) \
        launched = true;                                                                                                \
    if (!launched && nz == NZ_ && pw == PW_ && pr == PR_ && f.melv == MV_ && KS == KS_ && occ == OCC_ && cm == CM_) {   \
        auto* kfn = f.mel_ns <= 2 ? mfcc_stream512_kernel<NZ_, PW_, PR_, MV_, KS_, 2, OCC_, CM_, WALK>                    \
                                  : mfcc_stream512_kernel<NZ_, PW_, PR_, MV_, KS_, 4, OCC_, CM_, WALK>;                   \
        if (lds > 64 * 1024)                                                                                            \
            SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        int per_cu = 0;                                                                                                 \
        SSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * wg_waves, lds));                        \
        const int grid = std::min((n_chunks + wg_waves - 1) / wg_waves, std::max(1, per_cu) * p->ctx->num_cu);          \
        if (!WALK) SSP_HIP(hipMemsetAsync(sa.work_counter, 0, 64 + (size_t)n_chunks * 4, stream)); /* counters + flags */    \
        if (getenv("SSP_DEBUG")) fprintf(stderr, "[ssp] mfcc stream%s: grid %d (%d per CU), lds %zu\n", WALK ? " (second kernel)" : "", grid, per_cu, lds); \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * wg_waves), lds, stream, args, f, sa);                             \
        launched = true;                                                                                                \
    }
    // (scaling at three waves per SIMD exists for two scan steps only: occ above)
#define SSP_STREAM_CASE_CM3(MV_)                                                                                          \
    if (!launched && dry_run && nz == 13 && pw == 2 && pr == 1 && f.melv == MV_ && KS == 6 && occ == 3 && cm == 1) launched = true; \
    if (!launched && nz == 13 && pw == 2 && pr == 1 && f.melv == MV_ && KS == 6 && occ == 3 && cm == 1) {                   \
        auto* kfn = mfcc_stream512_kernel<13, 2, 1, MV_, 6, 2, 3, 1, WALK>;                                               \
        int per_cu = 0;                                                                                                   \
        SSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * wg_waves, lds));                          \
        const int grid = std::min((n_chunks + wg_waves - 1) / wg_waves, std::max(1, per_cu) * p->ctx->num_cu);            \
        if (!WALK) SSP_HIP(hipMemsetAsync(sa.work_counter, 0, 64 + (size_t)n_chunks * 4, stream));                            \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * wg_waves), lds, stream, args, f, sa);                               \
        launched = true;                                                                                                  \
    }
    SSP_STREAM_CASE_CM3(2) SSP_STREAM_CASE_CM3(3)
#undef SSP_STREAM_CASE_CM3
    SSP_STREAM_CASE_CM(13, 2, 1, 2, 6, 2, 1) SSP_STREAM_CASE_CM(13, 2, 1, 3, 6, 2, 1) SSP_STREAM_CASE_CM(13, 2, 1, 4, 6, 2, 1)
    if constexpr (!WALK) {  // dense bands (the PLP front end): no time steps, no second kernel
        SSP_STREAM_CASE(13, 2, 1, 0, 6, 2)
    }
#ifdef SSP_FAST_MINIMAL
    SSP_STREAM_CASE(13, 2, 1, 3, 6, 3)
    SSP_STREAM_CASE(16, 1, 0, 3, 10, 2)
    SSP_STREAM_CASE(16, 1, 0, 5, 10, 2)
#else
#define SSP_STREAM_MV(NZ_, PW_, PR_, KS_, OCC_)                                                                         \
    SSP_STREAM_CASE(NZ_, PW_, PR_, 2, KS_, OCC_) SSP_STREAM_CASE(NZ_, PW_, PR_, 3, KS_, OCC_) SSP_STREAM_CASE(NZ_, PW_, PR_, 4, KS_, OCC_)
    SSP_STREAM_MV(13, 2, 1, 6, 3)
    SSP_STREAM_MV(13, 2, 0, 6, 3)
    SSP_STREAM_MV(13, 1, 0, 6, 3)
    SSP_STREAM_MV(16, 2, 1, 6, 2)
    SSP_STREAM_MV(16, 2, 0, 6, 2)
    SSP_STREAM_MV(16, 1, 0, 6, 2)
    SSP_STREAM_MV(16, 2, 1, 10, 2)
    SSP_STREAM_MV(16, 2, 0, 10, 2)
    SSP_STREAM_MV(16, 1, 0, 10, 2)
    SSP_STREAM_MV(13, 2, 1, 10, 2)
    // the folded 40-filter bank of the in-repo MFCC at 8 kHz needs five 16-byte reads per lane
    SSP_STREAM_CASE(16, 2, 1, 5, 10, 2) SSP_STREAM_CASE(16, 2, 0, 5, 10, 2) SSP_STREAM_CASE(16, 1, 0, 5, 10, 2)
    SSP_STREAM_CASE(13, 2, 1, 5, 10, 2)
#undef SSP_STREAM_MV
#endif
#undef SSP_STREAM_CASE
#undef SSP_STREAM_CASE_CM
    if (!launched) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream): no kernel instance for this cfg");
    if (dry_run) return SSP_OK;
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

}  // namespace ssp

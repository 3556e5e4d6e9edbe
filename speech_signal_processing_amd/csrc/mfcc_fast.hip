// mfcc_fused512_kernel — the throughput kernel of the fused MFCC pass for n_fft == 512 (sidekit and in-repo dialects).
//
// Work decomposition (CDNA4, 64-wide waves):
//   workgroup (256 threads = 4 waves)  = PERSISTENT (3 per CU = 3 waves per SIMD, 168 VGPRs): claims chunks of frames of one
//                                        utterance (normally the whole utterance) from a global counter
//   wave                               = 4 frames at a time ("quad"), 16 lanes per frame; quads q = wave, wave + 4, ...
//   lane                               = 16 complex points of the 256-point complex FFT that carries the 512-point real FFT
//
// Per quad, per wave (no workgroup barrier inside the loop; every LDS region below is wave-private):
//   1. the quad's 3 hop + 32 NZ samples arrive by LDS-DMA one quad ahead (coalesced 16-byte buffer loads straight into the
//      wave's stage, bounds-checked: anything outside the utterance is the zero padding the dialects need); lanes read
//      (x[e], x[e+1]) and (x[e-2], x[e-1]), e = g*hop + 32 n1 + 2 n2, from the stage with aligned 8-byte reads
//   2. pre-emphasis + window in registers: z[n1] = (y[32 n1 + 2 n2], y[32 n1 + 2 n2 + 1]) * w     (n1 = 0..NZ-1, rest zero)
//   3. radix-16 FFT over n1 in registers, twiddle W_256^(n2 k1)
//   4. 16x16 transpose through LDS (unpadded 128-B rows, 16-byte chunks XOR-swizzled: conflict-free ds_write_b64 / ds_read_b128)
//   5. radix-16 FFT over n2 in registers -> Z[k1 + 16 k2]
//   6. split step of the real FFT: lane k1 owns the bin pairs k = k1 + 16 k2 <-> 256 - k (k2 < 8); the partners come from
//      lane 16 - k1 by two DPP row permutes; power / magnitude go to a per-frame P row in natural bin order
//   7. filterbank + log: register-resident pieces (MELV > 0: lane = up to 4 MELV taps of one filter, masked DPP scan over the
//      pieces of a filter) or the banded sweep from LDS tables (MELV = 0)
//   8. DCT rows: lane = cepstral index; the quad's cepstra go to the workgroup's slot of a global scratch (L2 resident)
// After the loop one barrier, then delta / delta-delta: without CMVN straight from the scratch to the output (4 frames per
// thread, buffer stores); with CMVN through an LDS copy of the cepstra, two-pass statistics and a coalesced block write.
// HBM sees every sample once and every output feature once (+ the scratch round trip at the L2's fabric side).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "mfcc.hpp"
#include "cplx.hpp"

namespace ssp {

constexpr int ZROW = 128;             // bytes per 16-complex row of the transpose / Z image: no padding, the 16-byte chunks of
                                      // row r are XOR-swizzled by (r >> 1) & 7 instead (conflict-free writes and 16-byte reads)
constexpr int ZFRAME = 16 * ZROW;     // 2048 B per frame
constexpr int MAX_PASS = MFCC_FAST_MAX_PASS;
constexpr int FAST_WAVES_DEFAULT = 4;  // waves per workgroup; two ~60 KiB workgroups per CU = 2 waves per SIMD, 256-VGPR budget
constexpr int LM_OFF = 1792;          // byte offset of frame 0's log-mel row (<= 48 floats) near the END of its Z image; frame g
                                      // sits 64 g bytes lower so the 4 frames' broadcast reads use different banks
constexpr int PSWEEP = (LM_OFF - 192) / 4;    // filterbank sweeps may run past the P row into stale (finite) Z data, never into log-mel

// storage index of spectrum bin b in a P row (natural order)
__host__ __device__ constexpr int p_sigma(int b) { return b; }

#ifdef SSP_STAMP
// Diagnostic build only (never shipped): per-phase cycle accounting of the quad loop with s_memtime stamps.
__device__ unsigned long long g_stamps[16];
#define STAMP(ph)                                                                                   \
    {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        st_acc[ph] += t_ - st_last;                                                                 \
        st_last = t_;                                                                               \
    }
#else
#define STAMP(ph)
#endif

struct __attribute__((packed, aligned(4))) f2u {
    float x, y;
};

// log of a filterbank output, branch free: floor_mode 1 adds eps, 2 clamps at eps (FastArgs carries 0 / -inf for the unused
// one); v_log_f32 (log2, ~1 ulp on the normal range) scaled to ln / log10 / 10 log10
__device__ __forceinline__ float fast_log(const FastArgs& f, float v) {
    return __builtin_amdgcn_logf(fmaxf(v + f.log_add, f.log_max)) * f.log_k;
}


// NZ: non-zero 32-sample rows of the window (13 for win <= 416, else 16); POWER: 1 magnitude | 2 power spectrum;
// PRE: per-frame pre-emphasis on/off; FAST_WAVES: waves per workgroup
// MELV: 0 = banded filterbank sweep from LDS tables | 2..5 = register-resident piece filterbank with MELV 16-byte reads per lane
// TUNED (0 | 8 | 12): the 13-cepstra / <= 32- or <= 48-filter / delta_N == 2 shapes of the reference's dialects as compile-time constants
//        (n_ceps 13, one DCT pass of 8 four-filter steps, <= 2 scan steps, N = 2 regression): no loop or branch overhead in
//        the filterbank / DCT stages, and a delta tail that emits 4 consecutive frames per thread
template <int NZ, int POWER, int PRE, int FAST_WAVES, int MELV, int TUNED>
#ifndef SSP_FAST_OCC
#define SSP_FAST_OCC 3  // waves per SIMD the register budget is cut for (168 VGPRs)
#endif
__global__ __launch_bounds__(64 * FAST_WAVES, SSP_FAST_OCC) void mfcc_fused512_kernel(MfccArgs a, FastArgs f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 64 * FAST_WAVES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-level control flow stays on the SALU
    const int g = lane >> 4, j = lane & 15;
    const int nc = TUNED ? 13 : a.n_ceps;
    const int q_pass = TUNED ? 1 : f.q_pass, n_filt4 = TUNED ? TUNED : f.n_filt4, mel_ns = TUNED ? 2 : f.mel_ns;

    float* s_melw = reinterpret_cast<float*>(smem + f.off_melw);
    int* s_melpk = reinterpret_cast<int*>(smem + f.off_mello);  // storage start | (filter id + 1) << 16
    float* s_dct = reinterpret_cast<float*>(smem + f.off_dct);
    float* s_stats = reinterpret_cast<float*>(smem + f.off_stats);
    // wave-private LDS: ONE region of 4 frame images (2048 B each) that is, in program order, the transpose image, the
    // Z image, the P rows and the log-mel rows of the quad
    char* zbuf = smem + f.off_wave + wave * f.wave_bytes;

    // ---- shared tables -> LDS
    // this lane's window taps stay in registers for the whole kernel: w[32 n1 + 2 j], w[32 n1 + 2 j + 1]
    v2f wreg[NZ];
#pragma unroll
    for (int n1 = 0; n1 < NZ; ++n1) wreg[n1] = *reinterpret_cast<const v2f*>(a.window + 32 * n1 + 2 * j);
#pragma unroll
    for (int n1 = 0; n1 < NZ; ++n1) wreg[n1] = edge_row_taps(wreg[n1], 32 * n1 + 2 * j, a.win_len);
    // this lane's twiddles stay in registers: W_256^(k1 j) for the step between the two radix-16 passes and
    // W_512^(8j+1+i) for the split step (LDS is the busiest unit of this kernel; registers are not)
    v2f twr[15], wpr[8];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) twr[k1 - 1] = *reinterpret_cast<const v2f*>(&f.tw16[k1 * 16 + j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) wpr[i] = *reinterpret_cast<const v2f*>(&f.wpost[i * 16 + j]);  // W_512^(j + 16 i)
    // piece filterbank: this lane's taps, read offsets, scan masks and (first lane of a run) filter id live in registers
    constexpr int MV = MELV > 0 ? MELV : 1;
    v4f mw[MV];
    int mofs[MV];
    v2f mk01 = v2f{0.f, 0.f}, mk23 = v2f{0.f, 0.f};
    int mfid = -1;
    if (MELV > 0) {
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            mw[i] = *reinterpret_cast<const v4f*>(f.pc_w + ((size_t)lane * MV + i) * 4);
            mofs[i] = f.pc_ofs[lane * MV + i];
        }
        mk01 = *reinterpret_cast<const v2f*>(f.pc_mask + lane * 4);
        mk23 = *reinterpret_cast<const v2f*>(f.pc_mask + lane * 4 + 2);
        mfid = f.pc_fid[lane];
    } else {
        for (int i = tid; i < f.total_steps * 64; i += NT) s_melw[i] = f.melw[i];
        for (int i = tid; i < f.n_pass * 16; i += NT) s_melpk[i] = f.mel_lo[i] | ((f.mel_id[i] + 1) << 16);
    }
    for (int i = tid; i < n_filt4 * 4 * q_pass * 16; i += NT) s_dct[i] = f.dctT[i];
    __syncthreads();

#ifdef SSP_SETUP_ONLY  // ablation: cost of workgroup launch + table setup alone
    if (mw[0].x == 12345.f) a.out[tid] = wreg[0].x + twr[3].y + wpr[2].x + mk01.x + mk23.y + (float)mofs[0] + (float)mfid + s_dct[tid];
    return;
#endif
    // ---- persistent workgroup: chunks are claimed from a global counter (ragged batches balance themselves); the tables
    //      above are loaded once per workgroup.  Cepstra of the chunk in flight go to this workgroup's slot of a small global
    //      scratch (L2 / Infinity-Cache resident: grid x 16 KiB) instead of LDS, which is what lets 3 workgroups share a CU.
    __shared__ int s_next;
    float* __restrict__ scr = f.ceps_scratch + (size_t)blockIdx.x * f.ceps_stride;
#ifdef SSP_STAMP
    unsigned long long st_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
#endif
  for (;;) {
    if (tid == 0) s_next = atomicAdd(f.work_counter, 1);
    __syncthreads();
    const int cidx = __builtin_amdgcn_readfirstlane(s_next);
    if (cidx >= f.n_chunks) break;
    const MfccChunk ch = a.chunks[cidx];
    const int64_t s0 = a.sample_off[ch.utt];
    const int64_t N = a.sample_off[ch.utt + 1] - s0;
    const int64_t f0 = a.frame_off[ch.utt];
    const int T = (int)(a.frame_off[ch.utt + 1] - f0);
    const float* __restrict__ x = a.samples + s0;
    const int t0 = ch.t0, n = ch.n;
    const int H = a.delta_order * a.delta_N;
    const int ta = max(t0 - H, 0), tb = min(t0 + n + H, T);
    const int hop = a.hop;
    const float pre = PRE ? a.preemph : 0.f;
    const int nquads = (tb - ta + 3) >> 2;

    // ---- sample path: the wave's 3*hop + 32*NZ samples of a quad stream HBM -> LDS by LDS-DMA (coalesced 16-byte
    //      buffer loads that write the wave-private stage directly: no VGPRs, 4 instructions per quad, every sample
    //      crosses L2 ~1.4x instead of the 2.1x of per-lane gathers), one quad ahead.  Bounds-checked: anything
    //      outside the utterance [0, N) lands as 0 = the zero padding the dialects need.
    // (the descriptor is assembled from readfirstlane'd words: the compiler must see it as wave-uniform, or every DMA
    //  instruction is wrapped in a waterfall loop of readfirstlane / compare / branch)
    const uint64_t xaddr = reinterpret_cast<uint64_t>(x);
    const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xaddr), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xaddr >> 32));
    const int xbytes = __builtin_amdgcn_readfirstlane((int)(N * 4));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((uint64_t)xhi << 32) | xlo), 0, xbytes, 0x00020000);
    float* stage = reinterpret_cast<float*>(zbuf + 4 * ZFRAME);
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)stage);
    const bool dma16 = ((xlo & 15) == 0) && ((hop & 3) == 0);  // wave-uniform
    const int n_piece = (f.slen + 255) >> 8;  // 1 KiB pieces (<= 5: slen <= 3 * 256 + 512 floats)
    const bool last_half = (f.slen & 255) != 0 && (f.slen & 255) <= 128;
    auto prefetch = [&](int q) {
        const int sq4 = (ta + 4 * q) * hop * 4;  // byte offset of the quad's first sample inside the utterance
        if (dma16) {
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (c < n_piece && (c + 1 < n_piece || !last_half || lane < 32))  // a trailing half piece: lanes 0..31 only
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(stage_lds + c * 1024), 16, sq4 + c * 1024 + lane * 16, 0, 0, 0);
        } else {  // ragged batches whose utterances do not start on 16-byte boundaries: 4-byte DMA pieces
            for (int c = 0; c < 4 * n_piece; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(stage_lds + c * 256), 4, sq4 + c * 256 + lane * 4, 0, 0, 0);
        }
    };
    v2f pf[NZ];   // (x[e], x[e+1]),  e = g*hop + 32 n1 + 2 j inside the staged quad
#ifdef SSP_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    prefetch(wave);
    float cdef = 0.f;   // the cepstrum of the previous quad: stored one iteration late so that the store (like the DMA) has a whole
    int cdef_off = -1;  // iteration to complete before the next s_waitcnt vmcnt(0)
    const float npre = -pre;
    for (int q = wave; q < nquads; q += FAST_WAVES) {
        const int t = ta + 4 * q + g;  // this lane group's frame
        // ---- 1+2. per-frame pre-emphasis (y[0] = x[0] - a x[0], y[n] = x[n] - a x[n-1]) and window, in registers.
        //      Both operand pairs (x[e], x[e+1]) and (x[e-1], x[e]) come from the LDS stage (the VALU, not the LDS, binds).
        v2f z[16];
#ifndef SSP_NO_DMAWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this quad's DMA has landed (issued one iteration ago)
#endif
        STAMP(10)  // wait for the DMA
        float pm[PRE ? NZ : 1];  // x[e-1]: the pre-emphasis partner comes from the stage too
        {
            // (volatile LDS-address-space loads, one ds_read_b64 + one ds_read_b32 per row: the compiler otherwise merges the pair with its
            //  partner into a ds_read2_b64 — 8 LDS cycles against 2 + 2 — as it did in mfcc_stream.hip before round 3)
            typedef __attribute__((address_space(3))) const volatile v2f* lds_cv2f_t;
            typedef __attribute__((address_space(3))) const volatile float* lds_cvf_t;
            const uint32_t sp = (uint32_t)(uintptr_t)(lds_ptr_t)stage + (g * hop + 2 * j) * 4;
#pragma unroll
            for (int n1 = 0; n1 < NZ; ++n1) {
                pf[n1] = *(lds_cv2f_t)(uintptr_t)(sp + 128 * n1);
                if (PRE) pm[n1] = *(lds_cvf_t)(uintptr_t)(sp + 128 * n1 - 4);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is in registers: the next quad may overwrite it
        STAMP(11)  // stage reads
#ifndef SSP_NO_DMA
        prefetch(q + FAST_WAVES);  // flies under this whole iteration (past the end it stages zeros)
#endif
        if (cdef_off >= 0) scr[cdef_off] = cdef;
        STAMP(12)  // DMA issue
        // y[n] = x[n] - a x[n-1]; the first sample of a frame pairs with itself (y[0] = x[0] - a x[0])
        if (PRE) pm[0] = (j == 0) ? pf[0].x : pm[0];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < NZ) {
                v2f y = pf[n1 < NZ ? n1 : 0];
                if (PRE) {
                    const float xm1 = pm[n1 < NZ ? n1 : 0], x0 = y.x, x1 = y.y;
                    y = v2f{__builtin_fmaf(npre, xm1, x0), __builtin_fmaf(npre, x0, x1)};
                }
                // (a zero weight of the PADDING silences whatever the sample holds — rows run past the window's last tap into the samples behind
                //  the frame, in windows shorter than 32 (NZ - 1) + 1 taps more rows than one: the legacy product, cplx.hpp, on every row of
                //  this fallback kernel: one instruction per row more than the packed product.  The wave-stream kernel takes the plain
                //  product; its scan kernel finds the chunks a leak shows up in and its third kernel redoes them.)
                z[n1] = wmul_edge(y, wreg[n1 < NZ ? n1 : 0]);
            } else {
                z[n1] = v2f{0.f, 0.f};
            }
        }
        STAMP(0)  // DMA landed + stage reads + next DMA issue + pre-emphasis + window
        // ---- 3. FFT16 over n1, twiddle W_256^(n2 k1)
#ifndef SSP_NO_FFT1
        fft16(z);
#endif
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], twr[k1 - 1]);
        STAMP(1)  // FFT1 + twiddle
        // ---- 4. transpose through LDS
        char* zf = zbuf + g * ZFRAME;
#ifndef SSP_NO_T2
        {
            // lane j (= n2) stores z[k1] into row k1, 8-byte slot j of the row: 16-byte chunk (j >> 1) ^ m, m = (k1 >> 1) & 7
            int wb0 = ((j >> 1) << 4) | ((j & 1) << 3);
            asm volatile("" : "+v"(wb0));  // keeps the 7 swizzled bases out of loop-invariant registers (one v_xor each instead)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                char* wp = zf + (wb0 ^ (m << 4));
                *reinterpret_cast<v2f*>(wp + (2 * m) * ZROW) = z[2 * m];
                *reinterpret_cast<v2f*>(wp + (2 * m + 1) * ZROW) = z[2 * m + 1];
            }
            // lane j (= k1) reads row j: logical chunk c (columns 2c, 2c+1) sits at chunk c ^ ((j >> 1) & 7)
            int rb0 = j * ZROW + (((j >> 1) & 7) << 4);
            asm volatile("" : "+v"(rb0));
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const v4f r = *reinterpret_cast<const v4f*>(zf + (rb0 ^ (c << 4)));
                z[2 * c] = v2f{r.x, r.y};
                z[2 * c + 1] = v2f{r.z, r.w};
            }
        }
        STAMP(2)  // transpose write + read
        // ---- 5. FFT16 over n2: lane j = k1, register = k2
        fft16(z);
#endif
        STAMP(3)  // FFT2
#ifdef SSP_NO_SPLIT
        { float* P = reinterpret_cast<float*>(zf);
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) P[j + 16 * k2] = z[k2].x * z[k2].x + z[k2].y * z[k2].y; }
#else
        // ---- 6. split step.  Lane j (= k1) owns the bin pairs k = j + 16 k2 <-> 256 - k for k2 = 0..7.  Z[256 - k] lives
        //         in lane 16 - j, register 15 - k2 (lane 0: its own register 16 - k2): the partners are fetched with two
        //         DPP row permutes per value (LDS is the binding unit of this kernel, the VALU is not), and every pair
        //         is formed exactly once.
        {
            // partner exchange inside the 16-lane row, no LDS: dst[j] = src[(16 - j) & 15] = row_shr:1(row_mirror(src)) for
            // j >= 1; lane 0 has no source in the shift and keeps `old` = its own partner Z[256 - 16 k2] (register 16 - k2)
            v2f zm[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const float sx = z[15 - k2].x, sy = z[15 - k2].y;
                const v2f own = z[(16 - k2) & 15];
                const float ox = own.x, oy = own.y;
                float mx = __builtin_amdgcn_update_dpp(sx, sx, 0x140 /*row_mirror*/, 0xF, 0xF, true);
                float my = __builtin_amdgcn_update_dpp(sy, sy, 0x140 /*row_mirror*/, 0xF, 0xF, true);
                mx = __builtin_amdgcn_update_dpp(ox, mx, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
                my = __builtin_amdgcn_update_dpp(oy, my, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
                zm[k2] = v2f{mx, my};
            }
            float* P = reinterpret_cast<float*>(zf);
            float pa[8], pb[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const v2f zk = z[k2];
                const v2f w = wpr[k2];                                                 // W_512^(j + 16 k2)
                const v2f e = __builtin_elementwise_fma(zm[k2], v2f{1.f, -1.f}, zk);   // 2E = Z[k] + conj Z[256-k]
                const v2f d = __builtin_elementwise_fma(zm[k2], v2f{-1.f, 1.f}, zk);   // 2D = Z[k] - conj Z[256-k]
                const v2f o = cmul_negi(d, w);                                         // 2 (-i D) W^k
                // 2 X[k] = e + o and conj(2 X[256-k]) = e - o in transposed form: R = (re, re'), I = (im, im'), so that both powers
                // come out of one packed multiply + one packed FMA (the broadcasts are operand selects)
                const v2f R = __builtin_elementwise_fma(xx(o), v2f{1.f, -1.f}, xx(e));
                const v2f I = __builtin_elementwise_fma(yy(o), v2f{1.f, -1.f}, yy(e));
                const v2f pw = __builtin_elementwise_fma(R, R, I * I);
                pa[k2] = pw.x;
                pb[k2] = pw.y;
                if (POWER == 1) {
                    pa[k2] = __builtin_amdgcn_sqrtf(pa[k2]);
                    pb[k2] = __builtin_amdgcn_sqrtf(pb[k2]);
                }
            }
            // bin 128 pairs with itself: 2 X[128] = 2 conj Z[128] (lane 0, register 8)
            const v2f s8 = z[8] * z[8];
            float p128 = 4.f * (s8.x + s8.y);
            if (POWER == 1) p128 = __builtin_amdgcn_sqrtf(p128);
            // the 1/4 (power) or 1/2 (magnitude) and spec_scale live in the filterbank weights
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) P[j + 16 * k2] = pa[k2];      // grouped by base so the stores pair into ds_write2_b32
            float* Pm = P + 144 - j;                                       // P[256 - j - 16 k2] = Pm[16 (7 - k2)]
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) Pm[16 * (7 - k2)] = pb[k2];
            if (j == 0) P[128] = p128;
        }
#endif
        STAMP(4)  // partner exchange + split step + P row
        // ---- 7. banded filterbank + log: lane = filter slot, 4 taps per step (16-byte LDS reads), swept in blocks of
        //         4 fully unrolled steps (weights zero padded to whole blocks), so a pass is a
        //         few rounds of independent loads instead of a long chain of dependent round trips.
#ifndef SSP_NO_MEL
        if (MELV > 0) {
            // piece filterbank: all 64 lanes work on ONE frame at a time.  Lane = up to 4*MELV consecutive taps of one
            // filter (weights in registers, MELV 16-byte reads of the frame's P row); the pieces of a filter sit in
            // consecutive lanes of a 16-lane row and are added by a masked DPP suffix scan, two frames per packed FMA;
            // the first lane of every run takes the log and stores the frame's log-mel entry.
            float* lm = reinterpret_cast<float*>(zf + LM_OFF - 64 * g);
            if (j < f.lm_pad) lm[a.n_filt + j] = 0.f;  // padded filter slots must read as finite zeros
            float sfr[4];
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) {
                const char* pr = zbuf + fr * ZFRAME;
                v4f acc = *reinterpret_cast<const v4f*>(pr + mofs[0]) * mw[0];
#pragma unroll
                for (int i = 1; i < MV; ++i) acc = __builtin_elementwise_fma(*reinterpret_cast<const v4f*>(pr + mofs[i]), mw[i], acc);
                const v2f h = v2f{acc.x, acc.y} + v2f{acc.z, acc.w};
                sfr[fr] = h.x + h.y;
            }
            v2f s01 = v2f{sfr[0], sfr[1]}, s23 = v2f{sfr[2], sfr[3]};
#define SSP_SCAN_STEP(CTRL, MK)                                                                             \
            {                                                                                                   \
                const float a0 = s01.x, a1 = s01.y, a2 = s23.x, a3 = s23.y;                                     \
                const float b0 = __builtin_amdgcn_update_dpp(a0, a0, CTRL, 0xF, 0xF, true);                   \
                const float b1 = __builtin_amdgcn_update_dpp(a1, a1, CTRL, 0xF, 0xF, true);                   \
                const float b2 = __builtin_amdgcn_update_dpp(a2, a2, CTRL, 0xF, 0xF, true);                   \
                const float b3 = __builtin_amdgcn_update_dpp(a3, a3, CTRL, 0xF, 0xF, true);                   \
                s01 = __builtin_elementwise_fma(v2f{b0, b1}, MK, s01);                                          \
                s23 = __builtin_elementwise_fma(v2f{b2, b3}, MK, s23);                                          \
            }
            if (mel_ns > 0) SSP_SCAN_STEP(0x101 /*row_shl:1*/, xx(mk01))
            if (mel_ns > 1) SSP_SCAN_STEP(0x102 /*row_shl:2*/, yy(mk01))
            if (mel_ns > 2) SSP_SCAN_STEP(0x104 /*row_shl:4*/, xx(mk23))
            if (mel_ns > 3) SSP_SCAN_STEP(0x108 /*row_shl:8*/, yy(mk23))
#undef SSP_SCAN_STEP
            if (mfid >= 0) {
                float* lmf = reinterpret_cast<float*>(zbuf + LM_OFF) + mfid;
                lmf[0 * (ZFRAME - 64) / 4] = fast_log(f, s01.x);
                lmf[1 * (ZFRAME - 64) / 4] = fast_log(f, s01.y);
                lmf[2 * (ZFRAME - 64) / 4] = fast_log(f, s23.x);
                lmf[3 * (ZFRAME - 64) / 4] = fast_log(f, s23.y);
            }
        } else {
            const float* P = reinterpret_cast<const float*>(zf);
            float* lm = reinterpret_cast<float*>(zf + LM_OFF - 64 * g);
            if (j < f.lm_pad) lm[a.n_filt + j] = 0.f;  // padded filter slots must read as finite zeros
            int wofs = 0;
            for (int ps = 0; ps < f.n_pass; ++ps) {
                const int nblk = f.mel_blocks[ps];  // 4-step blocks of this pass (weights zero padded to whole blocks)
                const int pk = s_melpk[ps * 16 + j];
                const v4f* pp = reinterpret_cast<const v4f*>(P + (pk & 0xffff));
                const v4f* ww = reinterpret_cast<const v4f*>(s_melw + (size_t)wofs * 64) + j;  // [step][lane][4]
                v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
                for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int st = blk * 4 + u;
                        acc = __builtin_elementwise_fma(pp[st], ww[st * 16], acc);
                    }
                }
                const int id = (pk >> 16) - 1;
                if (id >= 0) lm[id] = fast_log(f, (acc.x + acc.y) + (acc.z + acc.w));
                wofs += nblk * 4;
            }
        }
        STAMP(5)  // filterbank + log
        // ---- 8. DCT rows: lane = cepstral index, 4 filters per step, blocks of 4 fully unrolled steps
        {
            const v4f* lm4 = reinterpret_cast<const v4f*>(zf + LM_OFF - 64 * g);
#pragma unroll
            for (int qp = 0; qp < q_pass; ++qp) {
                const int qq = qp * 16 + j;
                const v4f* dd = reinterpret_cast<const v4f*>(s_dct) + qq;
                v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int blk = 0; blk < n_filt4 / 4; ++blk) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int st = blk * 4 + u;
                        acc = __builtin_elementwise_fma(lm4[st], dd[st * q_pass * 16], acc);
                    }
                }
                const float cv = (acc.x + acc.y) + (acc.z + acc.w);
                const int coff = (qq < nc && t < tb) ? (t - ta) * nc + qq : -1;
                if (qp == 0) {
                    cdef = cv;
                    cdef_off = coff;
                } else if (coff >= 0) {
                    scr[coff] = cv;
                }
            }
        }
#else
        cdef = reinterpret_cast<const float*>(zf)[j + 7];  // ablation: no filterbank / DCT
        cdef_off = (j < nc && t < tb) ? (t - ta) * nc + j : -1;
#endif
        STAMP(6)  // DCT + cepstra
    }
    STAMP(7)  // loop exit
    if (cdef_off >= 0) scr[cdef_off] = cdef;
    // no LDS-DMA may still be landing when the wave regions are reused below, and every cepstrum store must have completed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(8)  // barrier wait
#ifdef SSP_NO_TAIL
    if (tid == 0) a.out[(size_t)(f0 + t0) * a.d_out] = scr[0];
    __syncthreads();
    continue;
#endif
    if (TUNED && !a.cmvn && a.delta_order > 0) {
        // ---- direct tail (no CMVN): thread -> (4 consecutive frames, cepstral index); 12 cepstra straight from the scratch
        //      (L2 resident, written by this workgroup before the barrier), 4 x (c, delta, delta-delta) straight to the output
        //      with 4-byte buffer stores (the workgroup fills contiguous 4 x d_out blocks; L2 merges the lines).  No LDS
        //      image, no further barrier: the next chunk's claim barrier separates these reads from the next cepstrum stores.
        const int Dd = a.d_out;
        const float invd = a.delta_inv_denom;
        const uint64_t oaddr = reinterpret_cast<uint64_t>(a.out + (size_t)f0 * Dd);
        // (uint32_t temporaries: readfirstlane returns int, and a low word with bit 31 set would sign-extend into the high word)
        const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)oaddr), ohi = __builtin_amdgcn_readfirstlane((uint32_t)(oaddr >> 32));
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>(((uint64_t)ohi << 32) | olo), 0, __builtin_amdgcn_readfirstlane(T * Dd * 4), 0x00020000);
        auto put = [&](int idx, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, idx * 4, 0, 0); };
        auto cepg = [&](int u, int qq) -> float { return scr[(u - ta) * 13 + qq]; };
        auto dlg = [&](int u, int qq) -> float {  // (+ the n = 0 term of numpy.dot, GMM_UBM.py:68: 0 . NaN = NaN)
            return __builtin_fmaf(0.f, cepg(u, qq), ((cepg(min(u + 1, T - 1), qq) - cepg(max(u - 1, 0), qq)) + 2.f * (cepg(min(u + 2, T - 1), qq) - cepg(max(u - 2, 0), qq))) * invd);
        };
        const int ngrp = (n + 3) >> 2;
        for (int idx = tid; idx < ngrp * 13; idx += NT) {
            const int rg = idx / 13, qq = idx - rg * 13;
            const int u0 = t0 + 4 * rg;  // first frame of the group, utterance coordinates
            if (4 * rg + 3 < n && u0 >= 4 && u0 + 7 <= T - 1) {
                const float* cp = scr + (u0 - ta) * 13 + qq;
                float v[12];
                // (delta only: the chunk's halo is 2 frames, not 4 — rows u0 - 4, u0 - 3, u0 + 6, u0 + 7 feed the delta-delta alone and
                //  lie outside this workgroup's scratch slot: for the first / last slot outside the allocation)
                const bool dd2 = a.delta_order >= 2;
#pragma unroll
                for (int k = 0; k < 12; ++k) v[k] = (dd2 || (k >= 2 && k < 10)) ? cp[(k - 4) * 13] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c1 = __builtin_fmaf(0.f, v[4 + i], ((v[5 + i] - v[3 + i]) + 2.f * (v[6 + i] - v[2 + i])) * invd);
                    const float c2 = f.ddw[0] * (v[i] + v[8 + i]) + f.ddw[1] * (v[1 + i] + v[7 + i]) + f.ddw[2] * (v[2 + i] + v[6 + i]) +
                                     f.ddw[3] * (v[3 + i] + v[5 + i]) + f.ddw[4] * v[4 + i];
                    const int oi = (u0 + i) * Dd + qq;
                    put(oi, v[4 + i]);
                    put(oi + 13, c1);
                    if (a.delta_order >= 2) put(oi + 26, c2);
                }
            } else {
                for (int i = 0; i < 4; ++i)
                    if (4 * rg + i < n) {  // utterance / chunk edges: the nested edge-padded form
                        const int u = u0 + i, oi = u * Dd + qq;
                        put(oi, cepg(u, qq));
                        put(oi + 13, dlg(u, qq));
                        if (a.delta_order >= 2)
                            put(oi + 26, __builtin_fmaf(0.f, dlg(u, qq), ((dlg(min(u + 1, T - 1), qq) - dlg(max(u - 1, 0), qq)) +
                                                                          2.f * (dlg(min(u + 2, T - 1), qq) - dlg(max(u - 2, 0), qq))) * invd));
                    }
            }
        }
        STAMP(9)
        continue;
    }
    // ---- the chunk's cepstra come back from the scratch into the (now idle) wave regions: coalesced 16-byte copies
    float* s_ceps = reinterpret_cast<float*>(smem + f.off_wave);
    const int ceps_floats = ((tb - ta) * nc + 3) & ~3;
    for (int i = tid; i < ceps_floats / 4; i += NT) reinterpret_cast<v4f*>(s_ceps)[i] = reinterpret_cast<const v4f*>(scr)[i];
    __syncthreads();
    // ---- delta / delta-delta from the cepstra in LDS (edge padding at utterance ends, GMM_UBM.py:64)
    const int Nd = TUNED ? 2 : a.delta_N;
    const float inv = a.delta_inv_denom;
    const int D = a.d_out;
    auto cep = [&](int u, int qq) -> float { return s_ceps[(size_t)(u - ta) * nc + qq]; };
    auto dl = [&](int u, int qq) -> float {
        float acc = 0.f * cep(u, qq);  // (the n = 0 term of numpy.dot, GMM_UBM.py:68: 0 . NaN = NaN)
        for (int m = 1; m <= Nd; ++m) acc += (float)m * (cep(min(u + m, T - 1), qq) - cep(max(u - m, 0), qq));
        return acc * inv;
    };
    auto ddl = [&](int u, int qq) -> float {
        float acc = 0.f * dl(u, qq);
        for (int m = 1; m <= Nd; ++m) acc += (float)m * (dl(min(u + m, T - 1), qq) - dl(max(u - m, 0), qq));
        return acc * inv;
    };
    // c, delta, delta-delta of one (frame, cepstral index): interior frames take ONE sweep over the 4N+1 neighbours
    // (delta-delta = the cepstra convolved with the auto-convolution of the regression weights), edge frames the
    // nested edge-padded form
    auto emit = [&](int u, int qq, float& c0, float& c1, float& c2) {
        c0 = cep(u, qq);
        c1 = 0.f;
        c2 = 0.f;
        if (a.delta_order == 0) return;
        if (Nd == 2 && u >= 4 && u + 4 <= T - 1) {
            // the reference's N = 2 (GMM_UBM.py:53): 9 neighbours, no predicates
            const float* cp = s_ceps + (size_t)(u - ta) * nc + qq;
            const float m4 = cp[-4 * nc], m3 = cp[-3 * nc], m2 = cp[-2 * nc], m1 = cp[-nc], p1 = cp[nc], p2 = cp[2 * nc],
                        p3 = cp[3 * nc], p4 = cp[4 * nc];
            c1 = __builtin_fmaf(0.f, c0, ((p1 - m1) + 2.f * (p2 - m2)) * inv);  // (+ the n = 0 term: 0 . NaN = NaN)
            c2 = f.ddw[0] * (m4 + p4) + f.ddw[1] * (m3 + p3) + f.ddw[2] * (m2 + p2) + f.ddw[3] * (m1 + p1) + f.ddw[4] * c0;
        } else if (u - 2 * Nd >= 0 && u + 2 * Nd <= T - 1) {
            // all 4N+1 neighbours are fetched first (one LDS round trip), then reduced from registers (delta_N <= 4)
            float cv[17];
#pragma unroll
            for (int k = 0; k < 17; ++k) cv[k] = (k >= 8 - 2 * Nd && k <= 8 + 2 * Nd) ? cep(u + k - 8, qq) : 0.f;
#pragma unroll
            for (int k = 0; k < 17; ++k) {
                c2 = fmaf(f.ddw[min(max(k - 8 + 2 * Nd, 0), 16)], cv[k], c2);  // cv[k] == 0 outside the 4N+1 window
                if (k >= 8 - Nd && k <= 8 + Nd) c1 = fmaf((float)(k - 8) * inv, cv[k], c1);
            }
        } else {
            c1 = dl(u, qq);
            if (a.delta_order >= 2) c2 = ddl(u, qq);
        }
    };
    float* __restrict__ out = a.out + (size_t)(f0 + t0) * D;
    // ---- output: blocks of rows are assembled as a contiguous (rows x D) image in the waves' transpose LDS (idle by
    //      now) and leave with 16-byte coalesced stores; thread -> (row, cepstral index) emits c, delta, delta-delta.
    float* obuf = s_ceps + ceps_floats;
    const int rows_blk = max(4, ((FAST_WAVES * f.wave_bytes - ceps_floats * 4 - 16) / (D * 4)) & ~3);
    const int qsub = tid & 15, rsub = tid >> 4;
    if (a.cmvn) {
        // per-utterance CMVN: (x - mean) / std per output dimension, ddof = 0, std < 10 eps -> 1 (sklearn scale)
        auto value = [&](int r, int d) -> float {
            const int blk = d / nc, qq = d - blk * nc;
            float c0, c1, c2;
            emit(t0 + r, qq, c0, c1, c2);
            return blk == 0 ? c0 : (blk == 1 ? c1 : c2);
        };
        for (int d = wave; d < D; d += FAST_WAVES) {
            // (statistics over the entries that are not NaN, which stay NaN: sklearn's nanmean / nanstd — see cmvn_kernel)
            float s = 0.f, cn = 0.f;
            for (int r = lane; r < n; r += 64) {
                const float x = value(r, d);
                const bool ok = x == x;
                s += ok ? x : 0.f;
                cn += ok ? 1.f : 0.f;
            }
            for (int o = 32; o > 0; o >>= 1) {
                s += __shfl_xor(s, o);
                cn += __shfl_xor(cn, o);
            }
            const float mean = s / cn;
            float v = 0.f;
            for (int r = lane; r < n; r += 64) {
                const float x = value(r, d);
                const float e = x == x ? x - mean : 0.f;
                v = fmaf(e, e, v);
            }
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            float sd = sqrtf(v / cn);
            if (sd < 10.0f * 1.1920929e-07f) sd = 1.0f;
            if (lane == 0) {
                s_stats[d] = mean;
                s_stats[D + d] = 1.0f / sd;
            }
        }
        __syncthreads();
    }
    for (int r0 = 0; r0 < n; r0 += rows_blk) {
        const int nr = min(rows_blk, n - r0);
        float* dst = out + (size_t)r0 * D;
        // the LDS image starts at the destination's offset inside its 16-byte line, so image and destination share their
        // 16-byte phase: aligned 16-byte LDS reads feed aligned 16-byte global stores
        float* img = obuf + ((reinterpret_cast<uintptr_t>(dst) & 15) >> 2);
        auto put = [&](int r, int qq, float c0, float c1, float c2) {
            float* o = img + (size_t)r * D + qq;
            if (a.cmvn) {
                c0 = (c0 - s_stats[qq]) * s_stats[D + qq];
                if (a.delta_order >= 1) c1 = (c1 - s_stats[nc + qq]) * s_stats[D + nc + qq];
                if (a.delta_order >= 2) c2 = (c2 - s_stats[2 * nc + qq]) * s_stats[D + 2 * nc + qq];
            }
            o[0] = c0;
            if (a.delta_order >= 1) o[nc] = c1;
            if (a.delta_order >= 2) o[2 * nc] = c2;
        };
        if (TUNED) {
            // thread -> (group of 4 consecutive frames, cepstral index): 12 cepstra feed 4 x (c, delta, delta-delta)
            const int ngrp = (nr + 3) >> 2;
            for (int idx = tid; idx < ngrp * 13; idx += NT) {
                const int rg = idx / 13, qq = idx - rg * 13;
                const int u0 = t0 + r0 + 4 * rg;  // first frame of the group, utterance coordinates
                if (a.delta_order > 0 && 4 * rg + 3 < nr && u0 >= 4 && u0 + 7 <= T - 1) {
                    const float* cp = s_ceps + (size_t)(u0 - ta) * 13 + qq;
                    float v[12];
                    const bool dd2 = a.delta_order >= 2;  // (delta only: a 2-frame halo, see the direct tail)
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[k] = (dd2 || (k >= 2 && k < 10)) ? cp[(k - 4) * 13] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float c1 = __builtin_fmaf(0.f, v[4 + i], ((v[5 + i] - v[3 + i]) + 2.f * (v[6 + i] - v[2 + i])) * inv);
                        const float c2 = f.ddw[0] * (v[i] + v[8 + i]) + f.ddw[1] * (v[1 + i] + v[7 + i]) + f.ddw[2] * (v[2 + i] + v[6 + i]) +
                                         f.ddw[3] * (v[3 + i] + v[5 + i]) + f.ddw[4] * v[4 + i];
                        put(4 * rg + i, qq, v[4 + i], c1, c2);
                    }
                } else {
                    for (int i = 0; i < 4; ++i)
                        if (4 * rg + i < nr) {
                            float c0, c1, c2;
                            emit(u0 + i, qq, c0, c1, c2);
                            put(4 * rg + i, qq, c0, c1, c2);
                        }
                }
            }
        } else {
            for (int qq = qsub; qq < nc; qq += 16)
                for (int r = rsub; r < nr; r += NT / 16) {
                    float c0, c1, c2;
                    emit(t0 + r0 + r, qq, c0, c1, c2);
                    put(r, qq, c0, c1, c2);
                }
        }
        __syncthreads();
        // coalesced copy of nr * D floats: scalar head up to a 16-byte boundary, 16-byte body, scalar tail
        const int tot = nr * D;
        const int head = min(tot, (int)(((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15) >> 2));
        if (tid < head) dst[tid] = img[tid];
        const int nvec = (tot - head) >> 2;
        for (int i = tid; i < nvec; i += NT)
            *reinterpret_cast<v4f*>(dst + head + 4 * i) = *reinterpret_cast<const v4f*>(img + head + 4 * i);
        const int done = head + 4 * nvec;
        if (tid < tot - done) dst[done + tid] = img[done + tid];
        __syncthreads();
    }
    STAMP(9)  // delta / CMVN / output tail
  }  // persistent chunk loop
#ifdef SSP_STAMP
    if (lane == 0)
        for (int i = 0; i < 13; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
    if (tid == 0) atomicAdd(&g_stamps[15], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------------ host side
bool mfcc_fast_supported(const ssp_mfcc_cfg& c) {
    // floor_mode 2 (max(eps, .), numpy.maximum) stays with the generic kernel: fast_log / stream_log floor through fmaxf, which turns a NaN
    // mel sum (a NaN sample in the frame, or behind it in the stream kernel's padded rows) into log(eps) — finite and wrong, invisible
    // to the scan kernel; numpy.maximum keeps the NaN.  (No shipped 512-point preset uses that floor; librosa's is the 2048-point kernel's.)
    return c.n_fft == 512 && c.floor_mode != 2 && c.hop >= 2 && c.hop <= 256 && (c.hop & 1) == 0 && c.n_filt <= 64 && c.n_ceps <= 64 &&
           c.frame_mode != 2 && c.top_db < 0.f && (c.delta_order == 0 || c.delta_N <= 4);
}

int build_fast_tables(ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    const int nb = 257;
    std::vector<float2> tw16(256), wpost(144);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double ang = -2.0 * M_PI * (double)(k1 * n2) / 256.0;
            tw16[k1 * 16 + n2] = make_float2((float)cos(ang), (float)sin(ang));
        }
    for (int i = 0; i < 9; ++i)
        for (int pl = 0; pl < 16; ++pl) {
            const int k = i < 8 ? pl + 16 * i : 0;  // lane pl owns the pairs k = pl + 16 i, i = 0..7
            const double ang = -2.0 * M_PI * (double)k / 512.0;
            wpost[i * 16 + pl] = make_float2((float)cos(ang), (float)sin(ang));
        }
    // banded filters sorted by band length (descending) into slots of 16 lanes; bands are expressed in P-row STORAGE
    // coordinates (p_sigma) and cut into 4-tap steps that start on a 16-byte boundary
    std::vector<int32_t> lo(c.n_filt), len(c.n_filt), order(c.n_filt);
    std::vector<float> dense((size_t)c.n_filt * nb);
    SSP_HIP(hipMemcpy(dense.data(), p->fbank_dense.p, dense.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int jf = 0; jf < c.n_filt; ++jf) {
        int first = -1, last = -1;
        for (int k = 0; k < nb; ++k)
            if (dense[(size_t)jf * nb + k] != 0.f) {
                if (first < 0) first = k;
                last = k;
            }
        lo[jf] = first < 0 ? 0 : first;
        len[jf] = first < 0 ? 0 : last - first + 1;
        order[jf] = jf;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len[x] > len[y]; });
    const int n_pass = (c.n_filt + 15) / 16;
    FastArgs& f = p->fast;
    const int n_pass2 = (n_pass + 1) & ~1;  // passes are swept in pairs
    std::vector<int32_t> mel_lo(n_pass2 * 16, 0), mel_id(n_pass2 * 16, -1);
    std::vector<std::vector<float>> wpass(n_pass2);  // [pass][step][lane][4]
    std::vector<int> psteps(n_pass2, 0);
    const float pscale = (c.spec_power == 2 ? 0.25f : 0.5f) * c.spec_scale;  // the split step works on 2 X[k]
    for (int ps = 0; ps < n_pass; ++ps) {
        // Bank-conflict-free sweep: the 16 lanes of a pass read P with 16-byte loads that advance in lock step, so the
        // reads never conflict when the 16 band starts fall into 16 different 16-byte bank groups (start/4 mod 16).
        // Greedy, longest band first: move a band start DOWN by whole 4-tap steps (zero weights in front) until its
        // bank group is free; short bands absorb the shift without lengthening the pass.
        // (bipartite matching band -> bank group, smallest shift bound that admits a perfect matching)
        int start4[16], sl4v[16], shv[16], nf = 0, fl[16];
        for (int l = 0; l < 16; ++l) {
            const int s = ps * 16 + l;
            start4[l] = 0;
            if (s >= c.n_filt || len[order[s]] == 0) continue;
            const int jf = order[s];
            sl4v[l] = p_sigma(lo[jf]) & ~3;
            shv[l] = p_sigma(lo[jf] + len[jf] - 1);
            start4[l] = sl4v[l];
            fl[nf++] = l;
        }
        int smin = 1;
        for (int fi = 0; fi < nf; ++fi) smin = std::max(smin, (shv[fl[fi]] - sl4v[fl[fi]]) / 4 + 1);
        for (int S = smin; S <= smin + 16; ++S) {  // smallest pass length that admits a perfect matching
            int owner[16];  // bank group -> index into fl
            for (int r = 0; r < 16; ++r) owner[r] = -1;
            auto shift_for = [&](int fi, int r) -> int {  // smallest shift (in 4-tap steps) that puts band fi on group r
                const int l = fl[fi];
                for (int k = 0; sl4v[l] - 4 * k >= 0 && (shv[l] - (sl4v[l] - 4 * k)) / 4 + 1 <= S; ++k)
                    if ((((sl4v[l] - 4 * k) / 4) & 15) == r) return k;
                return -1;
            };
            bool seen[16];
            struct Rec {
                static bool go(int fi, int* owner, bool* seen, const decltype(shift_for)& sf) {
                    for (int r = 0; r < 16; ++r) {
                        if (seen[r] || sf(fi, r) < 0) continue;
                        seen[r] = true;
                        if (owner[r] < 0 || go(owner[r], owner, seen, sf)) {
                            owner[r] = fi;
                            return true;
                        }
                    }
                    return false;
                }
            };
            int matched = 0;
            for (int fi = 0; fi < nf; ++fi) {
                for (int r = 0; r < 16; ++r) seen[r] = false;
                if (Rec::go(fi, owner, seen, shift_for)) ++matched;
            }
            if (matched == nf) {
                for (int r = 0; r < 16; ++r)
                    if (owner[r] >= 0) start4[fl[owner[r]]] = sl4v[fl[owner[r]]] - 4 * shift_for(owner[r], r);
                break;
            }
        }
        int steps4 = 0;
        for (int fi = 0; fi < nf; ++fi) steps4 = std::max(steps4, (shv[fl[fi]] - start4[fl[fi]]) / 4 + 1);
        f.mel_steps[ps] = steps4;
        psteps[ps] = steps4;
        wpass[ps].assign((size_t)steps4 * 64, 0.f);
        for (int l = 0; l < 16; ++l) {
            const int s = ps * 16 + l;
            if (s >= c.n_filt) continue;
            const int jf = order[s];
            mel_id[s] = jf;
            const int sl4 = start4[l];
            mel_lo[s] = sl4;
            for (int k = 0; k < len[jf]; ++k) {
                const int pos = p_sigma(lo[jf] + k) - sl4;
                if (pos / 4 >= steps4) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): filterbank band longer than the pass sweep");
                wpass[ps][((size_t)(pos / 4) * 16 + l) * 4 + (pos & 3)] = pscale * dense[(size_t)jf * nb + lo[jf] + k];
            }
        }
    }
    // pack the passes: [step][lane][4], whole blocks of 4 steps, zero padded; every sweep must stay inside the P row +
    // the stale-but-finite Z data behind it (never reach the log-mel rows)
    std::vector<float> melw;
    int total = 0;
    for (int ps = 0; ps < n_pass; ++ps) {
        const int nblk = (psteps[ps] + 3) / 4;
        f.mel_blocks[ps] = nblk;
        melw.resize((size_t)(total + nblk * 4) * 64, 0.f);
        std::copy(wpass[ps].begin(), wpass[ps].end(), melw.begin() + (size_t)total * 64);
        for (int l = 0; l < 16; ++l)
            if (mel_lo[ps * 16 + l] + 16 * nblk > PSWEEP)
                SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): filterbank sweep does not fit the P row");
        total += nblk * 4;
    }
    for (int ps = n_pass; ps < MAX_PASS; ++ps) f.mel_blocks[ps] = 0;
    for (int ps = n_pass; ps < MAX_PASS; ++ps) f.mel_steps[ps] = 0;
    const int q_pass = (c.n_ceps + 15) / 16;
    const int n_filt4 = (((c.n_filt + 3) / 4) + 3) & ~3;  // 4-filter steps, padded to whole blocks of 4 steps
    // dctT[s][q][e] = dct[q][4 s + e] (zero padded): one 16-byte read per lane per 4 filters
    std::vector<float> dctT((size_t)n_filt4 * q_pass * 16 * 4, 0.f), dcth((size_t)c.n_ceps * c.n_filt);
    SSP_HIP(hipMemcpy(dcth.data(), p->dct.p, dcth.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int q = 0; q < c.n_ceps; ++q)
        for (int jf = 0; jf < c.n_filt; ++jf)
            dctT[((size_t)(jf / 4) * q_pass * 16 + q) * 4 + (jf & 3)] = dcth[(size_t)q * c.n_filt + jf];
    // delta-delta as ONE convolution for interior frames: auto-convolution of the regression weights n / denom
    {
        const int Nd = c.delta_order > 0 ? c.delta_N : 0;
        double den = 0;
        for (int i = 1; i <= Nd; ++i) den += 2.0 * i * i;
        for (int k = 0; k < 17; ++k) f.ddw[k] = 0.f;
        for (int m = -Nd; m <= Nd; ++m)
            for (int nn = -Nd; nn <= Nd; ++nn) f.ddw[m + nn + 2 * Nd] += (float)((double)m * nn / (den * den));
    }
    // ---- register-resident piece filterbank (kernel template MELV > 0).  A piece = up to PW = 4*MELV consecutive taps of
    //      one filter starting on a 16-byte boundary of the P row; a filter's pieces occupy consecutive lanes of one
    //      16-lane row (first-fit decreasing over the 4 rows).  The smallest MELV in 2..5 that fits 64 lanes wins;
    //      filterbanks that fit none (e.g. 40 folded talkbox filters at 8 kHz) keep the banded sweep (MELV = 0).
    f.melv = 0;
    f.mel_ns = 0;
    std::vector<float> pc_w;
    std::vector<int32_t> pc_ofs, pc_fid(64, -1);
    std::vector<float> pc_mask(64 * 4, 0.f);
    if (!getenv("SSP_MFCC_NO_PIECES")) {
        for (int mv = 2; mv <= 5 && f.melv == 0; ++mv) {
            const int PW = 4 * mv;
            std::vector<int> cnt(c.n_filt, 0), ord;
            int maxc = 0;
            for (int jf = 0; jf < c.n_filt; ++jf) {
                if (len[jf] == 0) continue;
                const int s4 = lo[jf] & ~3;
                cnt[jf] = (lo[jf] + len[jf] - s4 + PW - 1) / PW;
                maxc = std::max(maxc, cnt[jf]);
                ord.push_back(jf);
            }
            if (maxc > 16) continue;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return cnt[x] > cnt[y]; });
            int used[4] = {0, 0, 0, 0};
            std::vector<int> lane_of(c.n_filt, -1);
            bool ok = true;
            for (int jf : ord) {
                int r = 0;
                while (r < 4 && used[r] + cnt[jf] > 16) ++r;
                if (r == 4) { ok = false; break; }
                lane_of[jf] = r * 16 + used[r];
                used[r] += cnt[jf];
            }
            if (!ok) continue;
            // a P read may run past bin 256 into the stale (finite) Z data behind the row, never into the log-mel rows
            for (int jf : ord)
                if ((lo[jf] & ~3) + cnt[jf] * PW > PSWEEP) ok = false;
            if (!ok) continue;
            f.melv = mv;
            int ns = 0;
            while ((1 << ns) < maxc) ++ns;
            f.mel_ns = ns;
            pc_w.assign((size_t)64 * mv * 4, 0.f);
            pc_ofs.assign((size_t)64 * mv, 0);
            std::vector<int> pstart(64, 0), pfilt(64, -1);
            for (int jf : ord)
                for (int pcs = 0; pcs < cnt[jf]; ++pcs) {
                    const int l = lane_of[jf] + pcs;
                    pstart[l] = (lo[jf] & ~3) + pcs * PW;
                    pfilt[l] = jf;
                    if (pcs == 0) pc_fid[l] = jf;
                }
            for (int l = 0; l < 64; ++l)
                for (int st = 0; st < 4; ++st)
                    pc_mask[l * 4 + st] = (pfilt[l] >= 0 && (l & 15) + (1 << st) < 16 && pfilt[l + (1 << st)] == pfilt[l]) ? 1.f : 0.f;
            // read order: lane l reads its 16-byte slots in the rotated order (i + rot[l]) % mv, chosen greedily so that the
            // lanes of one ds_read_b128 conflict group hit different 16-byte bank slots of the 256-byte LDS row
            static const int grp_of_lane32[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
            auto grp = [&](int l) { return (l >> 5) * 2 + grp_of_lane32[l & 31]; };
            std::vector<int> rot(64, 0);
            // LDS cycles of the mv reads of one frame: per conflict group and read, the most distinct addresses on one slot
            auto cycles = [&]() {
                int tot = 0;
                for (int gq = 0; gq < 4; ++gq)
                    for (int i = 0; i < mv; ++i) {
                        int cntslot[16] = {0}, seen_addr[16][16], worst = 0;
                        for (int l = 0; l < 64; ++l) {
                            if (grp(l) != gq) continue;
                            const int a = pfilt[l] < 0 ? 0 : pstart[l] + 4 * ((i + rot[l]) % mv);
                            const int sl = (a / 4) & 15;
                            bool dup = false;
                            for (int k = 0; k < cntslot[sl]; ++k) dup = dup || seen_addr[sl][k] == a;
                            if (!dup) seen_addr[sl][cntslot[sl]++] = a;
                            worst = std::max(worst, cntslot[sl]);
                        }
                        tot += worst;
                    }
                return tot;
            };
            int best = cycles();
            uint32_t rng = 12345u;
            for (int it = 0; it < 20000 && best > 4 * mv; ++it) {
                rng = rng * 1664525u + 1013904223u;
                const int l = (rng >> 8) & 63;
                if (pfilt[l] < 0) continue;
                rng = rng * 1664525u + 1013904223u;
                const int old_rot = rot[l];
                rot[l] = (int)((rng >> 8) % (uint32_t)mv);
                const int cst = cycles();
                if (cst <= best) best = cst;
                else rot[l] = old_rot;
            }
            if (getenv("SSP_DEBUG")) fprintf(stderr, "[ssp] mfcc fast: piece filterbank melv %d, %d LDS cycles per frame (ideal %d)\n", mv, best, 4 * mv);
            for (int l = 0; l < 64; ++l) {
                for (int i = 0; i < mv; ++i) {
                    const int slot = (i + rot[l]) % mv;
                    pc_ofs[(size_t)l * mv + i] = (pstart[l] + 4 * slot) * 4;
                    for (int e = 0; e < 4; ++e) {
                        const int k = pstart[l] + 4 * slot + e;
                        const int jf = pfilt[l];
                        if (jf >= 0 && k >= lo[jf] && k < lo[jf] + len[jf] && k < nb)
                            pc_w[((size_t)l * mv + i) * 4 + e] = pscale * dense[(size_t)jf * nb + k];
                    }
                }
            }
        }
    }
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        SSP_TRY(b.alloc(bytes));
        if (bytes) SSP_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        return SSP_OK;
    };
    SSP_TRY(up(p->f_tw16, tw16.data(), tw16.size() * sizeof(float2)));
    SSP_TRY(up(p->f_wpost, wpost.data(), wpost.size() * sizeof(float2)));
    SSP_TRY(up(p->f_melw, melw.data(), melw.size() * sizeof(float)));
    SSP_TRY(up(p->f_mello, mel_lo.data(), mel_lo.size() * sizeof(int32_t)));
    SSP_TRY(up(p->f_melid, mel_id.data(), mel_id.size() * sizeof(int32_t)));
    SSP_TRY(up(p->f_dct, dctT.data(), dctT.size() * sizeof(float)));
    if (f.melv > 0) {
        SSP_TRY(up(p->f_pcw, pc_w.data(), pc_w.size() * sizeof(float)));
        SSP_TRY(up(p->f_pcofs, pc_ofs.data(), pc_ofs.size() * sizeof(int32_t)));
        SSP_TRY(up(p->f_pcmask, pc_mask.data(), pc_mask.size() * sizeof(float)));
        SSP_TRY(up(p->f_pcfid, pc_fid.data(), pc_fid.size() * sizeof(int32_t)));
        f.pc_w = p->f_pcw.as<float>();
        f.pc_ofs = p->f_pcofs.as<int32_t>();
        f.pc_mask = p->f_pcmask.as<float>();
        f.pc_fid = p->f_pcfid.as<int32_t>();
    }
    f.tw16 = p->f_tw16.as<float2>();
    f.wpost = p->f_wpost.as<float2>();
    f.melw = p->f_melw.as<float>();
    f.mel_lo = p->f_mello.as<int32_t>();
    f.mel_id = p->f_melid.as<int32_t>();
    f.dctT = p->f_dct.as<float>();
    f.n_pass = n_pass;
    f.lm_pad = n_filt4 * 4 - c.n_filt;
    f.q_pass = q_pass;
    f.n_filt4 = n_filt4;
    f.total_steps = total;
    const int NZ = c.win_len <= 416 ? 13 : 16;
    f.slen = 3 * c.hop + 32 * NZ;
    f.stage_floats = (f.slen + 255) & ~255;  // whole 256-float chunks are staged
    f.pscale = (c.spec_power == 2 ? 0.25f : 0.5f) * c.spec_scale;
    f.one_minus_a = 1.0f - c.preemph;
    f.log_add = c.floor_mode == 1 ? c.eps : 0.f;
    f.log_max = c.floor_mode == 2 ? c.eps : -INFINITY;
    f.log_k = c.log_mode == 0 ? 0.6931471805599453f : (c.log_mode == 1 ? 0.30102999566398120f : 3.0102999566398120f);
    p->fast_ready = true;
    return SSP_OK;
}

static size_t al16(size_t x) { return (x + 15) & ~size_t(15); }

// LDS carve for chunks of `ch` frames; returns total bytes
int mfcc_fast_waves() {
    const char* e = getenv("SSP_MFCC_WAVES");
    const int w = e ? atoi(e) : FAST_WAVES_DEFAULT;
#ifdef SSP_FAST_WAVES8
    return (w == 4 || w == 8) ? w : FAST_WAVES_DEFAULT;
#else
    (void)w;
    return FAST_WAVES_DEFAULT;
#endif
}

// LDS carve; returns total bytes.  The chunk's cepstra live in a global scratch, so the footprint does not depend on the
// chunk length: tables + FAST_WAVES wave regions (4 frame images + the sample stage each).
size_t mfcc_fast_lds(const ssp_mfcc_cfg& c, FastArgs& f, int ch) {
    const int H = c.delta_order * c.delta_N;
    size_t off = 0;
    f.off_win = f.off_tw16 = f.off_wpost = 0;
    f.off_melw = (int32_t)off;   off = al16(off + (f.melv > 0 ? 0 : (size_t)f.total_steps * 64 * 4));  // piece filterbank: weights in registers
    f.off_mello = (int32_t)off;  off = al16(off + (f.melv > 0 ? 0 : (size_t)f.n_pass * 16 * 4));
    f.off_melid = f.off_mello;
    f.off_dct = (int32_t)off;    off = al16(off + (size_t)f.n_filt4 * 4 * f.q_pass * 16 * 4);
    f.ceps_rows = ch + 2 * H;
    f.off_ceps = 0;
    f.off_stats = (int32_t)off;  off = al16(off + (size_t)2 * c.n_ceps * (1 + c.delta_order) * 4);
    off = (off + 255) & ~size_t(255);
    f.off_wave = (int32_t)off;
    f.wave_bytes = 4 * ZFRAME + (((f.slen + 255) >> 8) << 10);  // transpose images + the LDS-DMA sample stage
    return off + mfcc_fast_waves() * (size_t)f.wave_bytes + 256;  // + pad: lane 0 reads (and discards) one row past the last image
}

// the delta tail holds the chunk's cepstra AND at least 16 output rows in the wave regions
int mfcc_fast_max_chunk(const ssp_mfcc_cfg& c, const FastArgs& f) {
    const int H = c.delta_order * c.delta_N;
    const int d_out = c.n_ceps * (1 + c.delta_order);
    const int wave_bytes = 4 * ZFRAME + (((f.slen + 255) >> 8) << 10);
    const long avail = (long)mfcc_fast_waves() * wave_bytes - 16 - 16L * d_out * 4 - 16;
    return (int)std::max<long>(0, avail / (c.n_ceps * 4) - 2 * H);
}

int launch_mfcc_fast(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, int chunk_frames, hipStream_t stream) {
    if (n_chunks <= 0) return SSP_OK;
    FastArgs f = p->fast;
    size_t lds = mfcc_fast_lds(p->cfg, f, chunk_frames);
    if (const char* e = getenv("SSP_MFCC_LDS_PAD")) lds = std::min<size_t>(160 * 1024, lds + (size_t)atoi(e));  // diagnostic: caps the workgroups per CU
    if (lds > 160 * 1024) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): LDS footprint %zu B exceeds 160 KiB", lds);
    if (chunk_frames > mfcc_fast_max_chunk(p->cfg, f)) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): chunk of %d frames exceeds the delta tail's LDS", chunk_frames);
    f.ceps_stride = (f.ceps_rows * p->cfg.n_ceps + 3) & ~3;
    f.n_chunks = n_chunks;
    const int nz = p->cfg.win_len <= 416 ? 13 : 16, pw = p->cfg.spec_power, pr = p->cfg.preemph_mode ? 1 : 0;
    const int nw = mfcc_fast_waves();
    if ((int64_t)p->fast_max_samples * 4 > INT32_MAX) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): utterance too long for 32-bit offsets");
    bool launched = false;
    const ssp_mfcc_cfg& c = p->cfg;
    const int tuned = (f.melv > 0 && f.mel_ns <= 2 && c.n_ceps == 13 && (f.n_filt4 == 8 || f.n_filt4 == 12) && f.q_pass == 1 &&
                       (c.delta_order == 0 || c.delta_N == 2) && !getenv("SSP_MFCC_NO_TUNED")) ? f.n_filt4 : 0;
#define SSP_FAST_CASE(NZ_, PW_, PR_, NW_, MV_, TU_)                                                                    \
    if (!launched && nz == NZ_ && pw == PW_ && pr == PR_ && nw == NW_ && f.melv == MV_ && tuned == TU_) {             \
        auto* kfn = mfcc_fused512_kernel<NZ_, PW_, PR_, NW_, MV_, TU_>;                                                \
        if (lds > 64 * 1024)                                                                                           \
            SSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        int per_cu = 0;                                                                                                \
        SSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * NW_, lds));                            \
        const int grid = std::min(n_chunks, std::max(1, per_cu) * p->ctx->num_cu);                                       \
        SSP_TRY(p->f_scratch.reserve((size_t)grid * f.ceps_stride * sizeof(float)));                                   \
        SSP_TRY(p->f_counter.reserve(sizeof(int32_t)));                                                                \
        f.ceps_scratch = p->f_scratch.as<float>();                                                                     \
        f.work_counter = p->f_counter.as<int32_t>();                                                                   \
        SSP_HIP(hipMemsetAsync(f.work_counter, 0, sizeof(int32_t), stream));                                           \
        if (getenv("SSP_DEBUG")) fprintf(stderr, "[ssp] mfcc fast: grid %d (%d per CU), lds %zu\n", grid, per_cu, lds); \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * NW_), lds, stream, args, f);                                     \
        launched = true;                                                                                               \
    }
#define SSP_FAST_MV(NZ_, PW_, PR_, NW_)                                                                                \
    SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 0, 0) SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 2, 0) SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 3, 0)            \
    SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 4, 0) SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 3, 8) SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 4, 8)            \
    SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 3, 12) SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 4, 12) SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 5, 0)           \
    SSP_FAST_CASE(NZ_, PW_, PR_, NW_, 5, 12)
#ifdef SSP_FAST_WAVES8  // experiment: 8-wave workgroups (one per CU)
#define SSP_FAST_NW(NZ_, PW_, PR_) SSP_FAST_MV(NZ_, PW_, PR_, 4) SSP_FAST_MV(NZ_, PW_, PR_, 8)
#else
#define SSP_FAST_NW(NZ_, PW_, PR_) SSP_FAST_MV(NZ_, PW_, PR_, 4)
#endif
#ifdef SSP_FAST_MINIMAL  // diagnostic builds: only the benchmark instance
    SSP_FAST_CASE(13, 2, 1, 4, 3, 8)
#else
    SSP_FAST_NW(13, 2, 1)
    SSP_FAST_NW(13, 2, 0)
    SSP_FAST_NW(13, 1, 1)
    SSP_FAST_NW(13, 1, 0)
    SSP_FAST_NW(16, 2, 1)
    SSP_FAST_NW(16, 2, 0)
    SSP_FAST_NW(16, 1, 1)
    SSP_FAST_NW(16, 1, 0)
#endif
#undef SSP_FAST_NW
#undef SSP_FAST_MV
#undef SSP_FAST_CASE
    if (!launched) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(fast): no kernel instance for this cfg");
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

}  // namespace ssp

#ifdef SSP_STAMP
extern "C" int ssp_debug_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ssp::g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ssp::g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// mfcc_stream2048_kernel — first pass of the fused MFCC for n_fft == 2048 dialects without deltas (the librosa call site
// MFCC_DTW.py:28-31: 2048-sample periodic Hann window, hop 512, centred frames with reflect padding, power spectrum, 128 Slaney mel
// filters, 10 log10(max(1e-10, .)), clamp at the utterance maximum - 80 dB, DCT-II; and utils/processing.py:110-144 with frameSize 2048).
//
// Every WAVE is an independent stream (as in mfcc_stream.hip): it claims a chunk of consecutive frames of one utterance from a global
// counter and walks it one frame at a time, the whole wave on one frame, 16 complex points of the 1024-point complex FFT (which carries
// the 2048-point real FFT) per lane; no workgroup barrier after the tables are staged.  Per frame:
//   * the samples of the NEXT frame are loaded into registers (16 eight-byte loads per lane, coalesced; frames that touch the utterance
//     ends take a per-sample path with the reflect / zero rule) while this frame is transformed;
//   * 1024 = 16 x 16 x 4: radix-16 over the registers, twiddle, transpose through wave-private LDS, radix-16, twiddle, a second transpose
//     that leaves the four inputs of every radix-4 butterfly in one lane, radix-4;
//   * the spectrum goes to LDS in natural order, the split step of the real FFT pairs bins k and 1024 - k, power / magnitude row P[0..1024];
//   * filterbank as 4-tap pieces spread evenly over the lanes (a filter's pieces sit in different steps, so the LDS float adds that
//     collect a filter's sum do not collide), log, and the log filterbank row leaves for the second pass (the utterance-wide top_db clamp
//     needs every frame's maximum first; dialects without a clamp take the same second pass with the clamp off): launch_topdb_dct.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "mfcc.hpp"
#include "cplx.hpp"

namespace ssp {

namespace {
#ifndef SSP_2K_WAVES
#define SSP_2K_WAVES 12  // one 12-wave workgroup per CU = three waves per SIMD (170 VGPRs); the kernel has no barrier after the table staging
#endif
constexpr int S2K_WAVES = SSP_2K_WAVES;
constexpr int ROW1 = 68;                                   // complex slots per row of the first transpose image (64 + 4 pad)
constexpr int S2K_BUF_BYTES = 16 * ROW1 * 8;               // 8.5 KiB: the first transpose image; the natural-order spectrum image (1084 slots) fits it
constexpr int S2K_WAVE_BYTES = S2K_BUF_BYTES;
constexpr int TWB_ROW = 17;                                // second-pass twiddle rows [lb][ka], 17 slots apart: the four rows a ds_read_b64 touches sit in different banks
constexpr int S2K_TWB_BYTES = 4 * TWB_ROW * 8;
constexpr int S2K_WIN_BYTES = 2048 * 4, S2K_TWA_BYTES = 16 * 64 * 8;  // window and first-pass twiddles: workgroup-shared too (62 registers)
constexpr int S2K_TWS_BYTES = 9 * 64 * 8;                  // split twiddles (workgroup-shared: 18 registers per lane otherwise, and a spill
                                                           // reload inside the frame loop waits behind the next frame's sample loads)
__host__ __device__ constexpr int xpad(int k) { return k + ((k >> 6) << 2); }  // 64-point blocks 4 slots apart

// 8-byte LDS read that stays ONE ds_read_b64 (2 LDS cycles).  Left to itself the compiler pairs neighbouring reads into ds_read2_b64 /
// ds_read2st64_b64, which take 8 LDS cycles for the same 16 bytes (MI355X_MICROARCH.md, LDS table) — this kernel is LDS bound and has
// ~76 such reads per frame.  (LDS address space + volatile: a volatile access through a generic pointer would become a flat load.)
typedef __attribute__((address_space(3))) const volatile v2f* lds_cv2f_t;
__device__ __forceinline__ v2f lds_read_v2f(const void* generic_lds_ptr) {
    return *(lds_cv2f_t)(uintptr_t)(uint32_t)(uintptr_t)(lds_ptr_t)generic_lds_ptr;
}

struct __attribute__((packed, aligned(4))) f2u {
    float x, y;
};

// LDS operations of one wave execute in issue order: between the phases that exchange data through the wave buffer only the COMPILER must
// be kept from reordering.  (A wavefront-scope fence also makes it wait for every outstanding global load — the next frame's samples,
// issued a moment earlier.)
__device__ __forceinline__ void wave_sync2k() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// bounds-checked buffer atomics without a return value (the LLVM intrinsics: this clang has no builtin for them)
extern "C" __device__ int buf_atomic_smax(int, __amdgpu_buffer_rsrc_t, int, int, int) __asm("llvm.amdgcn.raw.ptr.buffer.atomic.smax.i32");
extern "C" __device__ int buf_atomic_umin(int, __amdgpu_buffer_rsrc_t, int, int, int) __asm("llvm.amdgcn.raw.ptr.buffer.atomic.umin.i32");
extern "C" __device__ int buf_atomic_swap(int, __amdgpu_buffer_rsrc_t, int, int, int) __asm("llvm.amdgcn.raw.ptr.buffer.atomic.swap.i32");

__device__ __forceinline__ float log2k(const MfccArgs& a, float v) {
    if (a.floor_mode == 1) v += a.eps;
    else if (a.floor_mode == 2) v = fmaxf(v, a.eps);  // (a NaN is dropped here: frames whose spectrum is not finite never get this far, see row_bad)
    const float l2 = __builtin_amdgcn_logf(v);
    return l2 * (a.log_mode == 0 ? 0.6931471805599453f : (a.log_mode == 1 ? 0.30102999566398120f : 3.0102999566398120f));
}
}  // namespace

struct S2kArgs {
    const float2* twA;   // [16][64]  W_1024^(l k1)
    const float2* twB;   // [4][TWB_ROW]   W_64^(lb ka)
    const float2* twS;   // [9][64]   W_2048^k, k = l + 64 i (k <= 512)
    const char* mel;     // [steps0 + steps1][64] 4-tap weight steps: the group of the 64 shortest filters, then the 64 longest
    const int32_t* minfo;  // [2][64][2] per group and lane: byte offset of the first step in the P row, filter id (-1: none)
    int32_t* work_counter;
    int32_t steps0, steps1, n_chunks, table_bytes;
    int32_t fuse;        // every utterance of the batch is one chunk: the wave that walked it also clamps its rows and takes the DCT
    float top_db;        // (< 0: no clamp)
};

// POWER: 2 = power spectrum, 1 = magnitude.  A template parameter, not a.spec_power: as a run-time (wave-uniform) condition inside the
// unrolled split loop the compiler if-converted it — BOTH sides evaluated, the magnitude side being a correctly rounded square root of
// ~18 instructions per bin — which cost the power dialects 290 of their 845 vector instructions per frame.
// SH: rows of 128 samples by which consecutive frames advance when the hop is a whole number of rows (hop = 128 SH: 4 for the librosa
// dialect's 512, 8 for 1024; 0: any other hop).  A lane's 16 sample pairs of the next frame are then its pairs SH..15 of this frame
// (in padded-signal coordinates, so at the utterance ends too) plus SH new ones: 16 - SH of the 16 eight-byte loads per frame — every
// sample was fetched 2048 / hop times through L2 — become register moves.
template <int POWER, int SH>
__global__ __launch_bounds__(64 * S2K_WAVES, 1) void mfcc_stream2048_kernel(MfccArgs a, S2kArgs s) {
#ifdef SSP_2K_REGTAB
    static_assert(S2K_WAVES <= 8, "register-resident tables need the 256-VGPR budget of two waves per SIMD");
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup-shared tables: twB, split twiddles, window, first-pass twiddles, filterbank weight steps
    {
        constexpr int NB = S2K_TWB_BYTES / 16, NS = S2K_TWS_BYTES / 16, NW = S2K_WIN_BYTES / 16, NA = S2K_TWA_BYTES / 16;
        const int n16 = s.table_bytes >> 4;
        const v4f* src_b = reinterpret_cast<const v4f*>(s.twB);
        const v4f* src_s = reinterpret_cast<const v4f*>(s.twS);
        const v4f* src_w = reinterpret_cast<const v4f*>(a.window);
        const v4f* src_a = reinterpret_cast<const v4f*>(s.twA);
        const v4f* src_m = reinterpret_cast<const v4f*>(s.mel);
        v4f* dst = reinterpret_cast<v4f*>(smem);
        for (int i = tid; i < n16; i += (int)blockDim.x) {
            v4f v;
            if (i < NB) v = src_b[i];
            else if (i < NB + NS) v = src_s[i - NB];
            else if (i < NB + NS + NW) v = src_w[i - NB - NS];
            else if (i < NB + NS + NW + NA) v = src_a[i - NB - NS - NW];
            else v = src_m[i - NB - NS - NW - NA];
            dst[i] = v;
        }
    }
    // (in-wave finish only) the DCT matrix, [16][128] zero padded, behind the tables
    float* dctl = reinterpret_cast<float*>(smem + s.table_bytes);
    if (s.fuse)
        for (int i = tid; i < 16 * 128; i += (int)blockDim.x) {
            const int q = i >> 7, j = i & 127;
            dctl[i] = (q < a.n_ceps && j < a.n_filt) ? a.dct[q * a.n_filt + j] : 0.f;
        }
    __syncthreads();
    const v2f* twB = reinterpret_cast<const v2f*>(smem);
    const v2f* twS = reinterpret_cast<const v2f*>(smem + S2K_TWB_BYTES);
    const v2f* winl = reinterpret_cast<const v2f*>(smem + S2K_TWB_BYTES + S2K_TWS_BYTES) + lane;                 // pairs (w[2m], w[2m+1]), m = 64 r + l
    const v2f* twAl = reinterpret_cast<const v2f*>(smem + S2K_TWB_BYTES + S2K_TWS_BYTES + S2K_WIN_BYTES) + lane;  // [k1][l]
    const char* melt = smem + S2K_TWB_BYTES + S2K_TWS_BYTES + S2K_WIN_BYTES + S2K_TWA_BYTES;
    char* wbase = smem + s.table_bytes + (s.fuse ? 16 * 128 * 4 : 0) + wave * S2K_WAVE_BYTES;
    v2f* buf = reinterpret_cast<v2f*>(wbase);

    // per group and lane (byte offset of the first step in the P row, filter id): read from the LDS copy behind the weight steps when a
    // frame's filterbank starts — as four registers held for the kernel's life they were what the allocator spilled at 168
    const int2* minfol = reinterpret_cast<const int2*>(melt + (s.steps0 + s.steps1) * 1024) + lane;
    const int hop = a.hop;
    const int M = 1024;
    const bool centre = a.frame_mode == 2;
#ifdef SSP_2K_REGTAB  // (experiment: window, first-pass and split twiddles resident in registers at two waves per SIMD)
    v2f wreg[16], tAreg[15], tSreg[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) wreg[r] = winl[64 * r];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tAreg[k1 - 1] = twAl[64 * k1];
#pragma unroll
    for (int i = 0; i < 8; ++i) tSreg[i] = twS[i * 64 + lane];
#endif

    for (;;) {
        int cidx = 0;
        if (lane == 0) cidx = atomicAdd(s.work_counter, 1);
        cidx = __builtin_amdgcn_readfirstlane(cidx);
        if (cidx >= s.n_chunks) break;
        const MfccChunk ch = a.chunks[cidx];
        const int64_t s0 = a.sample_off[ch.utt];
        const int64_t N = a.sample_off[ch.utt + 1] - s0;
        const int64_t f0 = a.frame_off[ch.utt];
        const float* __restrict__ x = a.samples + s0;
        const int t0 = __builtin_amdgcn_readfirstlane(ch.t0), n = __builtin_amdgcn_readfirstlane(ch.n);
        float wave_max = -INFINITY;
        int wave_nan = 0;  // a frame of this chunk had a non-finite spectrum (wave-uniform)
        // rows R0..15 of frame t
        auto load_frame = [&](int t, v2f (&v)[16], auto r0tag) {
            constexpr int R0 = decltype(r0tag)::value;
            const int64_t g0 = (int64_t)t * hop - (centre ? M : 0);
            if (g0 + 128 * R0 >= 0 && g0 + 2 * M <= N) {
                const float* __restrict__ xp = x + g0 + 2 * lane;
#pragma unroll
                for (int r = R0; r < 16; ++r) {
                    const f2u t2 = *reinterpret_cast<const f2u*>(xp + 128 * r);
                    v[r] = v2f{t2.x, t2.y};
                }
            } else {
                // frames that touch the utterance ends: every index is formed first and every load is unconditional, so the 32 loads
                // of a lane are in flight together (a conditional load per sample serialises them: one memory latency EACH, which made
                // the four edge frames of a 47-frame utterance cost more than the other 43)
                const int g0i = (int)g0, Ni = (int)N;
                // (the lane id from an opaque instruction: with the kernel's `lane` the 16 offsets 128 r + 2 lane are loop invariants,
                //  hoisted to the kernel's top, spilled at 168 registers and reloaded here one by one — a scratch round trip in front of
                //  every load of the edge path)
                int lane_e;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
#pragma unroll
                for (int r = R0; r < 16; ++r) {
                    float e[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        int g = g0i + 128 * r + 2 * lane_e + c;
                        // (uniform base + unsigned 32-bit byte offset: the load takes its base from SGPRs and one index register, not a
                        //  64-bit per-lane address — 32 of those at once were the kernel's register peak)
                        const char* xb = reinterpret_cast<const char*>(x);
                        if (centre) {  // numpy.pad(mode='reflect'), as frame_sample() of the generic kernel
                            g = g < 0 ? -g : g;
                            g = g >= Ni ? 2 * (Ni - 1) - g : g;
                            g = max(0, min(g, Ni - 1));
                            e[c] = *reinterpret_cast<const float*>(xb + (unsigned)(g * 4));
                        } else {
                            const float xv = *reinterpret_cast<const float*>(xb + (unsigned)(min(g, Ni - 1) * 4));
                            e[c] = g < Ni ? xv : 0.f;
                        }
                    }
                    v[r] = v2f{e[0], e[1]};
                }
            }
        };

        v2f nx[16];
        load_frame(t0, nx, std::integral_constant<int, 0>{});
        for (int t = t0; t < t0 + n; ++t) {
            v2f z[16];
#pragma unroll
#ifdef SSP_2K_REGTAB
            for (int r = 0; r < 16; ++r) z[r] = nx[r] * wreg[r];
#else
            for (int r = 0; r < 16; ++r) z[r] = nx[r] * lds_read_v2f(&winl[64 * r]);
#endif
#ifndef SSP_2K_NOPREFETCH  // (ablation, wrong results: every frame transforms the chunk's first one)
            if (t + 1 < t0 + n) {
                if constexpr (SH > 0) {
#pragma unroll
                    for (int r = 0; r < 16 - SH; ++r) nx[r] = nx[r + SH];
                }
                load_frame(t + 1, nx, std::integral_constant<int, (SH > 0 ? 16 - SH : 0)>{});
            }
#endif
            // ---- pass 1: DFT16 over r (points 64 r + l), twiddle W_1024^(l k1)
            fft16(z);
#pragma unroll
#ifdef SSP_2K_REGTAB
            for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], tAreg[k1 - 1]);
#else
            for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], lds_read_v2f(&twAl[64 * k1]));
#endif
            // ---- transpose 1: lane (k1, lb) <- points l = 4 la + lb of row k1
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) buf[k1 * ROW1 + lane] = z[k1];
            wave_sync2k();
            const int k1p = lane >> 2, lb = lane & 3;
#pragma unroll
            for (int la = 0; la < 16; ++la) z[la] = lds_read_v2f(&buf[k1p * ROW1 + 4 * la + lb]);
            wave_sync2k();
            // ---- pass 2: DFT16 over la, twiddle W_64^(lb ka)
            fft16(z);
#pragma unroll
            for (int ka = 1; ka < 16; ++ka) z[ka] = cmul(z[ka], lds_read_v2f(&twB[lb * TWB_ROW + ka]));
            // ---- transpose 2: the four lb of a (k1, ka) pair side by side (32-byte unit, units of a row XOR-swizzled by k1).  The
            // reading lane (k1, kah = ka >> 2) takes a unit as two 16-byte halves; lanes kah and kah + 2 of a ds_read_b128 lane set
            // would meet on one 16-byte slot, so the units of kah >= 2 (ka >= 8) hold their halves swapped and those lanes read the
            // upper half first: 16 lanes, 16 slots
#pragma unroll
            for (int ka = 0; ka < 16; ++ka) buf[(k1p * 16 + (ka ^ k1p)) * 4 + (lb ^ ((ka >> 3) << 1))] = z[ka];
            wave_sync2k();
            const int k1q = lane >> 2, kah = lane & 3, swp = (kah >> 1) << 1;
            v2f y[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int unit = k1q * 16 + ((4 * kah + i) ^ k1q);
                const v4f q0 = *reinterpret_cast<const v4f*>(&buf[unit * 4 + swp]), q1 = *reinterpret_cast<const v4f*>(&buf[unit * 4 + 2 - swp]);
                y[i][0] = v2f{q0.x, q0.y};
                y[i][1] = v2f{q0.z, q0.w};
                y[i][2] = v2f{q1.x, q1.y};
                y[i][3] = v2f{q1.z, q1.w};
            }
            wave_sync2k();
            // ---- pass 3: DFT4 over lb; Z[k1 + 16 ka + 256 kb] into the natural-order image
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dft4(y[i][0], y[i][1], y[i][2], y[i][3]);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) buf[xpad(k1q + 16 * (4 * kah + i) + 256 * kb)] = y[i][kb];
            }
            wave_sync2k();
            // ---- split step of the real FFT (bins k and M - k from the pair Z[k], Z[M - k]), power / magnitude
            // (k = lane + 64 i covers bins 0..511 and their partners 1024..513; bin 512 pairs with itself: X[512] = conj Z[512])
            float pa[8], pb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane + 64 * i;
                const v2f zk = lds_read_v2f(&buf[xpad(k)]);
                const v2f zm = lds_read_v2f(&buf[xpad((M - k) & (M - 1))]);
                // (E' = Z[k] + conj Z[M - k] = 2 E, D' = Z[k] - conj Z[M - k] = 2 D: the halvings and the spectrum's scale are folded into
                //  the filterbank weights — two packed additions instead of a multiplication and two FMAs per bin pair)
                const v2f zc = v2f{zm.x, -zm.y};
                const v2f e = zk + zc;
                const v2f d = zk - zc;
#ifdef SSP_2K_REGTAB
                const v2f o = cmul_negi(d, tSreg[i]);
#else
                const v2f o = cmul_negi(d, lds_read_v2f(&twS[i * 64 + lane]));
#endif
                const v2f xa = e + o, xb = e - o;
                float p0 = xa.x * xa.x + xa.y * xa.y, p1 = xb.x * xb.x + xb.y * xb.y;
                if (POWER == 1) {  // v_sqrt_f32 (1 ulp): the magnitude feeds a filter sum and a logarithm
                    p0 = __builtin_amdgcn_sqrtf(p0);
                    p1 = __builtin_amdgcn_sqrtf(p1);
                }
                pa[i] = p0;
                pb[i] = p1;
            }
            const int2 mi0 = minfol[0], mi1 = minfol[64];
            const int mstart0 = mi0.x, mfid0 = mi0.y, mstart1 = mi1.x, mfid1 = mi1.y;
            const v2f z512 = lds_read_v2f(&buf[xpad(M / 2)]);  // (one address for the wave: a broadcast)
            // (X[512] = conj Z[512]: against the other bins' 2 X[k] this one carries a factor 4 (power) / 2 (magnitude))
            float p512 = 4.f * (z512.x * z512.x + z512.y * z512.y);
            if (POWER == 1) p512 = __builtin_amdgcn_sqrtf(p512);
            // a NaN / inf sample anywhere in the frame makes EVERY bin of its spectrum non-finite (each is a sum over all samples), so one
            // bin tells: such a frame's log filterbank row is NaN (numpy.maximum keeps the NaN the floor would drop) and so is the
            // utterance maximum of the top_db clamp (ndarray.max()) — one compare per frame, the rest wave-uniform and cold
            const bool row_bad = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_classf(p512, 0x203 /* NaN, +inf */)) != 0;
            wave_sync2k();
            float* P = reinterpret_cast<float*>(buf);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane + 64 * i;
                P[k] = pa[i];
                P[M - k] = pb[i];
            }
            if (lane == 0) P[M / 2] = p512;
            if (lane < 4) P[M + 1 + lane] = 0.f;  // (16-byte reads: the taps behind bin 1024 carry zero weights)
            wave_sync2k();
            // ---- filterbank + log: a lane per filter, the 64 shortest filters then the 64 longest (each group sweeps as many 4-tap
            // steps as its longest filter needs; shorter ones carry zero weights, and start early enough to stay inside the P row);
            // fixed summation order per filter, no cross-lane traffic; the row leaves for the second pass
#ifndef SSP_2K_NOMEL
            const __amdgpu_buffer_rsrc_t ro =
                __builtin_amdgcn_make_buffer_rsrc(a.lm_out + (size_t)(f0 + t) * (size_t)a.n_filt, 0, a.n_filt * 4, 0x00020000);
            if (row_bad) {  // (cold)
                const float qn = __builtin_nanf("");
                if (mfid0 >= 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, qn), ro, mfid0 * 4, 0, 0);
                if (mfid1 >= 0 && s.steps1 != 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, qn), ro, mfid1 * 4, 0, 0);
                wave_nan = s.top_db >= 0.f ? 1 : 0;  // (the clamp's ndarray.max() becomes NaN; dialects without a clamp keep their NaN rows to themselves)
            } else
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int nst = g == 0 ? s.steps0 : s.steps1;
                if (nst == 0) continue;
#ifdef SSP_2K_MELW_GLOBAL  // (experiment: the weight steps straight from global memory / L1 instead of the LDS copy)
                const v4f* wt = reinterpret_cast<const v4f*>(s.mel) + (size_t)(g == 0 ? 0 : s.steps0) * 64 + lane;
#else
                const v4f* wt = reinterpret_cast<const v4f*>(melt) + (size_t)(g == 0 ? 0 : s.steps0) * 64 + lane;
#endif
                const char* pp = reinterpret_cast<const char*>(P) + (g == 0 ? mstart0 : mstart1);
                v4f acc0 = v4f{0.f, 0.f, 0.f, 0.f}, acc1 = v4f{0.f, 0.f, 0.f, 0.f};
                // four steps per trip while they last (eight 16-byte reads in flight behind one wait: a trip is an LDS round trip with
                // four packed FMAs' worth of work), then two; even steps feed acc0, odd steps acc1, in step order
                int st = 0;
                for (; st + 4 <= nst; st += 4) {
                    v4f w[4], q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        w[u] = wt[(st + u) * 64];
                        q[u] = *reinterpret_cast<const v4f*>(pp + 16 * (st + u));
                    }
                    acc0 = __builtin_elementwise_fma(q[0], w[0], acc0);
                    acc1 = __builtin_elementwise_fma(q[1], w[1], acc1);
                    acc0 = __builtin_elementwise_fma(q[2], w[2], acc0);
                    acc1 = __builtin_elementwise_fma(q[3], w[3], acc1);
                }
                for (; st < nst; st += 2) {  // (step counts are even)
                    const v4f w0 = wt[st * 64], w1 = wt[(st + 1) * 64];
                    const v4f p0 = *reinterpret_cast<const v4f*>(pp + 16 * st), p1 = *reinterpret_cast<const v4f*>(pp + 16 * st + 16);
                    acc0 = __builtin_elementwise_fma(p0, w0, acc0);
                    acc1 = __builtin_elementwise_fma(p1, w1, acc1);
                }
                const v4f acc = acc0 + acc1;
                const int fid = g == 0 ? mfid0 : mfid1;
                if (fid >= 0) {
                    const float v = log2k(a, (acc.x + acc.y) + (acc.z + acc.w));
                    // (a buffer store: the row base travels in SGPRs, the lane adds its filter's 4-byte column.  As a flat 64-bit per-lane
                    //  address this was a value the register allocator spilled — reloaded from scratch before each of a frame's two
                    //  stores, with a vmcnt(0) behind it that also waited for the next frame's 16 sample loads.  Issuing the stores one
                    //  frame late, ahead of the next sample loads, so that the window multiply's vmcnt(0) does not cover them, was
                    //  measured: slower)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, fid * 4, 0, 0);
                    wave_max = fmaxf(wave_max, v);
                }
            }
#endif
            wave_sync2k();
        }
        // utterance maximum for the second pass (float order through the integer trick, as the generic kernel)
        for (int o = 32; o > 0; o >>= 1) wave_max = fmaxf(wave_max, __shfl_xor(wave_max, o));
        if (wave_nan) wave_max = __builtin_nanf("");
        if (s.fuse) {
            // ---- the whole utterance was this wave's: clamp at its maximum - top_db (librosa power_to_db) and DCT-II, rows re-read
            // through L2.  Lane (q = lane & 15, part = lane >> 4): coefficient q over filters 32 part .. 32 part + 31
            const float thr = s.top_db >= 0.f ? wave_max - s.top_db : -INFINITY;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows are written
            const int rl = lane >> 2, part = lane & 3, nc = a.n_ceps, nf = a.n_filt;
            // (the halves go through UNSIGNED variables: readfirstlane returns int, and an int low half with bit 31 set would
            //  sign-extend over the high half when the address is put together)
            const uint64_t raddr = reinterpret_cast<uint64_t>(a.lm_out + (size_t)(f0 + t0) * nf);
            const uint32_t rlo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)raddr);
            const uint32_t rhi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(raddr >> 32));
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<float*>(((uint64_t)rhi << 32) | (uint64_t)rlo), 0, __builtin_amdgcn_readfirstlane(n * nf * 4), 0x00020000);
            // 16 rows per trip: lane (row = lane >> 2, part = lane & 3) holds filters 32 part .. 32 part + 31 of its row; every coefficient
            // is a product with the DCT row's same quarter (LDS, broadcast over the rows) summed over the four parts of a lane quad
            for (int r0 = 0; r0 < n; r0 += 16) {
                const int r = r0 + rl;
                v4f l[8];
#pragma unroll
                for (int i4 = 0; i4 < 8; ++i4) {
                    const int j = 32 * part + 4 * i4;
                    l[i4] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rr, (r < n && j < nf) ? (r * nf + j) * 4 : 0x7ffffff0, 0, 1 /*glc*/));
                    l[i4] = v4f{nanmax(l[i4].x, thr), nanmax(l[i4].y, thr), nanmax(l[i4].z, thr), nanmax(l[i4].w, thr)};  // (numpy.maximum)
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (q >= nc) break;
                    const v4f* dq = reinterpret_cast<const v4f*>(dctl + q * 128 + 32 * part);
                    v4f acc = l[0] * dq[0];
#pragma unroll
                    for (int i4 = 1; i4 < 8; ++i4) acc = __builtin_elementwise_fma(l[i4], dq[i4], acc);
                    float v = (acc.x + acc.y) + (acc.z + acc.w);
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    if (part == 0 && r < n) a.out[(size_t)(f0 + t0 + r) * nc + q] = v;
                }
            }
        } else {
            // ---- the chunk's maximum joins its utterance's (float order through the integer trick, as the generic kernel; a NaN sticks:
            // mfcc.hip).  ONE lane's atomic, but NO lane-0 block: a block of lane 0 at the bottom of this loop sits, across the back edge,
            // right in front of the lane-0 block of the next claim; the compiler threads the other lanes around both, the chunk loop
            // becomes a loop over lane masks whose lanes leave separately, and with lane 0 gone nobody claims — readfirstlane returns
            // the initial 0 and the wave walks chunk 0 for ever (round 4's hang on ragged batches, the only ones that get here;
            // tools/microbench/lane0_loop.hip).  So every lane issues the three bounds-checked buffer atomics, and all but
            // the one that applies aim out of range.
            const uint64_t ua = reinterpret_cast<uint64_t>(a.utt_max + ch.utt);
            const uint32_t ulo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ua), uhi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ua >> 32));
            const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)uhi << 32) | (uint64_t)ulo), 0, 4, 0x00020000);
            const bool one = lane == 0;
            const int bits = __float_as_int(wave_max);
            buf_atomic_swap(0x7fc00000, ru, (one && wave_nan) ? 0 : 0x7ffffff0, 0, 0);
            buf_atomic_smax(bits, ru, (one && !wave_nan && wave_max >= 0.f) ? 0 : 0x7ffffff0, 0, 0);
            buf_atomic_umin(bits, ru, (one && !wave_nan && wave_max < 0.f && wave_max > -INFINITY) ? 0 : 0x7ffffff0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool mfcc_s2k_supported(const ssp_mfcc_plan* p) { return p->s2k_ready && !getenv("SSP_MFCC_NO_STREAM2K"); }

static bool s2k_cfg_ok(const ssp_mfcc_cfg& c) {
    return c.n_fft == 2048 && c.win_len == 2048 && c.n_filt <= 128 && c.preemph_mode == 0 && c.delta_order == 0 && c.cmvn == 0 &&
           c.hop >= 1 && (c.spec_power == 1 || c.spec_power == 2);
}

int build_s2k_tables(ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    p->s2k_ready = false;
    if (!s2k_cfg_ok(c)) return SSP_OK;
    const int nb = 1025;
    std::vector<float2> twA(16 * 64), twB(4 * TWB_ROW), twS(9 * 64);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int l = 0; l < 64; ++l) {
            const double ang = -2.0 * M_PI * (double)(l * k1) / 1024.0;
            twA[k1 * 64 + l] = make_float2((float)cos(ang), (float)sin(ang));
        }
    for (int lb = 0; lb < 4; ++lb)
        for (int ka = 0; ka < 16; ++ka) {
            const double ang = -2.0 * M_PI * (double)(lb * ka) / 64.0;
            twB[lb * TWB_ROW + ka] = make_float2((float)cos(ang), (float)sin(ang));
        }
    for (int i = 0; i < 9; ++i)
        for (int l = 0; l < 64; ++l) {
            const int k = std::min(l + 64 * i, 512);
            const double ang = -2.0 * M_PI * (double)k / 2048.0;
            twS[i * 64 + l] = make_float2((float)cos(ang), (float)sin(ang));
        }
    // filters sorted by their number of 4-tap steps: the 64 shortest form group 0, the rest group 1; a lane sweeps its filter from the
    // first bin rounded down to 4, moved down (leading zero taps) where the group's step count would leave the P row
    std::vector<float> fb((size_t)c.n_filt * nb);
    SSP_HIP(hipMemcpy(fb.data(), p->fbank_dense.p, fb.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<int> lo(c.n_filt, 0), len(c.n_filt, 0), order(c.n_filt);
    for (int j = 0; j < c.n_filt; ++j) {
        int first = -1, last = -1;
        for (int k = 0; k < nb; ++k)
            if (fb[(size_t)j * nb + k] != 0.f) {
                if (first < 0) first = k;
                last = k;
            }
        lo[j] = first < 0 ? 0 : first;
        len[j] = first < 0 ? 0 : last - first + 1;
        order[j] = j;
    }
    auto nsteps = [&](int j) { return len[j] == 0 ? 0 : (lo[j] + len[j] - 1) / 4 - lo[j] / 4 + 1; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nsteps(x) < nsteps(y); });
    const int P_END = 1028;  // floats of the P row that hold finite values (bins 0..1024 + 3 zeros)
    int gsteps[2] = {0, 0};
    for (int i = 0; i < c.n_filt; ++i) gsteps[i >> 6] = std::max(gsteps[i >> 6], nsteps(order[i]));
    for (int g = 0; g < 2; ++g) gsteps[g] = (gsteps[g] + 1) & ~1;  // even (the sweep takes two steps per trip)
    if (c.n_filt <= 64) gsteps[1] = 0;
    std::vector<int32_t> minfo(2 * 64 * 2, 0);
    for (int i = 0; i < 128; ++i) minfo[i * 2 + 1] = -1;
    std::vector<char> mel((size_t)(gsteps[0] + gsteps[1]) * 64 * 16, 0);
    // Lane assignment inside a group.  Every step is one ds_read_b128 of the P row at a per-lane address; the LDS serves that
    // instruction in four fixed 16-lane sets, one cycle per set when the 16 lanes sit in 16 different 16-byte slots of a 256-byte
    // bank row (MI355X_MICROARCH.md, LDS) — and all lanes advance by the same 16 bytes per step, so a set that is conflict-free
    // at the first step stays so.  Filters are therefore dealt to the sets so that the slots of a set differ, a filter's start
    // moving down by whole 4-tap steps (leading zero weights) where the group's sweep leaves it the room.
    static const int kSet[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    // the kernel's split step leaves 2 X[k] (its two halvings are not executed) and does not apply the spectrum's scale: both ride in the
    // weights — (scale / 4) w on a power spectrum, (scale / 2) w on a magnitude spectrum (a filter sum is linear in its weights)
    const float fold = (c.spec_power == 2 ? 0.25f : 0.5f) * c.spec_scale;
    int slot_clashes = 0;
    for (int g = 0; g < 2; ++g) {
        const int i0 = g * 64, i1 = std::min(c.n_filt, i0 + 64);
        if (i1 <= i0) break;
        struct Cand { int j, start_hi, n_opt; };
        std::vector<Cand> cand;
        for (int i = i0; i < i1; ++i) {
            const int j = order[i];
            int start = lo[j] & ~3;
            start = std::min(start, (P_END - 4 * gsteps[g]) & ~3);
            if (start < 0 || start + 4 * gsteps[g] < lo[j] + len[j]) return SSP_OK;  // (a filter wider than the sweep: the generic kernel keeps the plan)
            int n_opt = 1;  // starts start, start - 4, ... that still cover the filter
            while (start - 4 * n_opt >= 0 && start - 4 * n_opt + 4 * gsteps[g] >= lo[j] + len[j] && n_opt < 16) ++n_opt;
            cand.push_back({j, start, len[j] == 0 ? std::min(16, start / 4 + 1) : n_opt});  // (a filter with no tap may sit anywhere at or below its start — not in front of the P row)
        }
        std::stable_sort(cand.begin(), cand.end(), [](const Cand& x, const Cand& y) { return x.n_opt < y.n_opt; });
        bool used[4][16] = {}, lane_taken[64] = {};
        int fill[4] = {0, 0, 0, 0};
        float* wt = reinterpret_cast<float*>(mel.data()) + (size_t)(g == 0 ? 0 : gsteps[0]) * 64 * 4;
        for (const Cand& cd : cand) {
            int best_set = -1, best_m = 0;
            for (int m = 0; m < cd.n_opt && best_set < 0; ++m) {
                const int slot = ((cd.start_hi - 4 * m) / 4) & 15;
                int pick = -1;
                for (int q = 0; q < 4; ++q)  // the emptiest set whose slot is free
                    if (fill[q] < 16 && !used[q][slot] && (pick < 0 || fill[q] < fill[pick])) pick = q;
                if (pick >= 0) { best_set = pick; best_m = m; }
            }
            if (best_set < 0) {  // no free slot anywhere: the emptiest set, at the natural start (one extra LDS cycle per step)
                for (int q = 0; q < 4; ++q)
                    if (fill[q] < 16 && (best_set < 0 || fill[q] < fill[best_set])) best_set = q;
                ++slot_clashes;
            }
            const int start = cd.start_hi - 4 * best_m;
            used[best_set][(start / 4) & 15] = true;
            const int l = kSet[best_set][fill[best_set]++];
            lane_taken[l] = true;
            minfo[(g * 64 + l) * 2] = start * 4;
            // (filters with no taps at all still produce log(0 + floor): they are listed with their id and zero weights)
            minfo[(g * 64 + l) * 2 + 1] = cd.j;
            for (int k = 0; k < len[cd.j]; ++k) {
                const int tap = lo[cd.j] + k - start;
                wt[((size_t)(tap / 4) * 64 + l) * 4 + (tap & 3)] = fold * fb[(size_t)cd.j * nb + lo[cd.j] + k];
            }
        }
        // lanes without a filter read where another lane of their set reads (same address: a broadcast, no bank of its own)
        for (int q = 0; q < 4; ++q)
            for (int e = fill[q]; e < 16; ++e)
                if (fill[q] > 0) minfo[(g * 64 + kSet[q][e]) * 2] = minfo[(g * 64 + kSet[q][0]) * 2];
        (void)lane_taken;
    }
    if (getenv("SSP_DEBUG")) fprintf(stderr, "[ssp] mfcc stream2048 tables: %d + %d filterbank steps, %d filters without a conflict-free slot\n", gsteps[0], gsteps[1], slot_clashes);
    {  // the lane table rides behind the weight steps (one LDS staging loop)
        const size_t w = mel.size();
        mel.resize(w + minfo.size() * sizeof(int32_t));
        memcpy(mel.data() + w, minfo.data(), minfo.size() * sizeof(int32_t));
    }
    const size_t table_bytes = S2K_TWB_BYTES + S2K_TWS_BYTES + S2K_WIN_BYTES + S2K_TWA_BYTES + mel.size();
    if (table_bytes + 16 * 128 * 4 + (size_t)4 * S2K_WAVE_BYTES > 160 * 1024) return SSP_OK;  // (filterbanks whose weight steps do not fit keep the generic kernel)
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        SSP_TRY(b.alloc(bytes));
        SSP_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        return SSP_OK;
    };
    SSP_TRY(up(p->s2k_twA, twA.data(), twA.size() * sizeof(float2)));
    SSP_TRY(up(p->s2k_twB, twB.data(), twB.size() * sizeof(float2)));
    SSP_TRY(up(p->s2k_twS, twS.data(), twS.size() * sizeof(float2)));
    SSP_TRY(up(p->s2k_mel, mel.data(), mel.size()));
    SSP_TRY(up(p->s2k_minfo, minfo.data(), minfo.size() * sizeof(int32_t)));
    p->s2k_steps = gsteps[0];
    p->s2k_steps1 = gsteps[1];
    p->s2k_ready = true;
    return SSP_OK;
}

// every utterance of a batch is one chunk and the tables allow the in-wave finish: no second pass
bool mfcc_s2k_fuses(const ssp_mfcc_plan* p, int64_t max_T, int chunk_frames) {
    const ssp_mfcc_cfg& c = p->cfg;
    return max_T <= chunk_frames && c.n_ceps <= 16 && (c.n_filt & 3) == 0 && c.n_filt <= 128 && !getenv("SSP_2K_NO_FUSE");
}

int launch_mfcc_s2k(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream) {
    if (n_chunks <= 0) return SSP_OK;
    S2kArgs s{};
    s.fuse = p->cache_s2k_fused ? 1 : 0;
    s.top_db = p->cfg.top_db;
    s.twA = p->s2k_twA.as<float2>();
    s.twB = p->s2k_twB.as<float2>();
    s.twS = p->s2k_twS.as<float2>();
    s.mel = p->s2k_mel.as<char>();
    s.minfo = p->s2k_minfo.as<int32_t>();
    s.steps0 = p->s2k_steps;
    s.steps1 = p->s2k_steps1;
    s.n_chunks = n_chunks;
    s.table_bytes = S2K_TWB_BYTES + S2K_TWS_BYTES + S2K_WIN_BYTES + S2K_TWA_BYTES + (p->s2k_steps + p->s2k_steps1) * 64 * 16 + 2 * 64 * 8;
    const size_t dct_bytes = s.fuse ? (size_t)16 * 128 * 4 : 0;
    SSP_TRY(p->f_counter.reserve(64));
    s.work_counter = p->f_counter.as<int32_t>();
    // waves per workgroup: 12 (one workgroup per CU, three waves per SIMD) where the tables leave room, else 8 or 4
    int waves = n_chunks >= p->ctx->num_cu * S2K_WAVES ? S2K_WAVES : 4;  // (small batches: 4-wave workgroups spread the chunks over more CUs)
    while (waves > 4 && (size_t)s.table_bytes + dct_bytes + (size_t)waves * S2K_WAVE_BYTES > 160 * 1024) waves -= 4;
    const size_t lds = (size_t)s.table_bytes + dct_bytes + (size_t)waves * S2K_WAVE_BYTES;
    if (lds > 160 * 1024) SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc(stream2048): LDS footprint %zu B exceeds 160 KiB", lds);
    const int sh = p->cfg.hop == 512 ? 4 : (p->cfg.hop == 1024 ? 8 : 0);
    const bool mag = p->cfg.spec_power == 1;
    auto kern = sh == 4 ? (mag ? mfcc_stream2048_kernel<1, 4> : mfcc_stream2048_kernel<2, 4>)
                        : (sh == 8 ? (mag ? mfcc_stream2048_kernel<1, 8> : mfcc_stream2048_kernel<2, 8>)
                                   : (mag ? mfcc_stream2048_kernel<1, 0> : mfcc_stream2048_kernel<2, 0>));
    if (getenv("SSP_2K_NO_SLIDE")) kern = mag ? mfcc_stream2048_kernel<1, 0> : mfcc_stream2048_kernel<2, 0>;
    const void* kfn = reinterpret_cast<const void*>(kern);
    if (lds > 64 * 1024) SSP_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    SSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * waves, lds));
    const int grid = std::min((n_chunks + waves - 1) / waves, std::max(1, per_cu) * p->ctx->num_cu);
    SSP_HIP(hipMemsetAsync(s.work_counter, 0, 64, stream));
    if (getenv("SSP_DEBUG")) fprintf(stderr, "[ssp] mfcc stream2048: grid %d (%d per CU), lds %zu, %d + %d filterbank steps\n", grid, per_cu, lds, p->s2k_steps, p->s2k_steps1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, stream, args, s);
    SSP_HIP(hipGetLastError());
    return SSP_OK;
}

}  // namespace ssp

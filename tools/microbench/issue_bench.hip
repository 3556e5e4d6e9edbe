// Issue-cost microbenchmark for gfx950: cycles per wave-instruction per SIMD at 1..4 waves per SIMD for the instruction kinds the
// fused MFCC kernel is built from (plain / packed fp32 VALU, DPP moves, transcendental, LDS reads / writes, fp32 MFMA beside VALU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/issue_bench tools/microbench/issue_bench.hip && tools/microbench/issue_bench
// Every kernel runs ITERS iterations of a body of REP independent instructions (8 accumulator chains) in inline asm; lane 0 of
// every wave stores its s_memtime span; the host reports  span / (REP * ITERS * waves_per_simd)  = cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA, PKFMA, PKADD, PKMUL, ADD, MUL, DPPMOV, DPPADD, FMAC, LOGF, PKFMA_SEL, MIX_PK_PLAIN, DSR64, DSR128, DSW64, DSW32, DSR32,
          MFMA16, MFMA16_V4, MFMA16_V8, MFMA16_V12, MFMA4, MFMA4_V2, PKFMA_DEP, FMA_DEP, BPERM, MFMA16_PK2, MFMA16_PK4, MFMA16_PK6, SNOP, VMOV, MFMA32BF, MFMA32BF_PK4, CVTBF, PERM32, SPLIT2, NOPS };
static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32", "v_mul_f32", "v_mov_b32_dpp row_mirror",
                              "v_add_f32_dpp row_mirror", "v_fmac_f32", "v_log_f32", "v_pk_fma_f32 op_sel", "pk_fma + fma alternating (per pair)",
                              "ds_read_b64", "ds_read_b128", "ds_write_b64", "ds_write_b32", "ds_read_b32",
                              "mfma_16x16x4_f32 alone", "mfma_16x16x4 + 4 v_fma (per group)", "mfma_16x16x4 + 8 v_fma (per group)", "mfma_16x16x4 + 12 v_fma (per group)",
                              "mfma_4x4x1_16b_f32 alone", "mfma_4x4x1 + 2 v_fma (per group)", "v_pk_fma_f32 dependent chain", "v_fma_f32 dependent chain",
                              "ds_bpermute_b32", "mfma_16x16x4 + 2 v_pk_fma (per group)", "mfma_16x16x4 + 4 v_pk_fma (per group)", "mfma_16x16x4 + 6 v_pk_fma (per group)",
                              "s_nop 0", "v_mov_b32", "mfma_32x32x16_bf16 alone", "mfma_32x32x16_bf16 + 4 v_pk_fma (per group)", "v_cvt_pk_bf16_f32",
                              "v_permlane32_swap_b32", "bf16 hi/lo split of a float pair (cvt, 2 unpack, pk_add, cvt: per group of 5)"};
// instructions per body (for reporting): groups count as 1
static int body_count(int op) { return 32; }

template <int OP>
__global__ __launch_bounds__(256) void bench(unsigned long long* spans, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    v2f a[8], b = v2f{1.0001f, 0.9999f}, c = v2f{1e-7f, -1e-7f};
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = v2f{1.f + i + lane, 2.f + i}; s[i] = 1.f + i + lane * 0.5f; }
    v4f q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f acc4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v16f acc16[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc16[i][k] = 0.f;
    v4f opa = v4f{1.5f, 2.5f, 3.5f, 4.5f}, opb = v4f{0.5f, 0.25f, 0.125f, 1.f};
    unsigned hb[8], lb2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hb[i] = lb2[i] = i;
    const unsigned ldsaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (lane * 16) % 8192;
    const unsigned ldsaddr8 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + lane * 8;
    const unsigned ldsaddr4 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + lane * 4;
    const int bp = ((lane * 7) & 63) * 4;
    unsigned long long t0, t1;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
                R8(X)
#undef X
            } else if constexpr (OP == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R8(X)
#undef X
            } else if constexpr (OP == PKFMA_SEL) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
                R8(X)
#undef X
            } else if constexpr (OP == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                R8(X)
#undef X
            } else if constexpr (OP == PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                R8(X)
#undef X
            } else if constexpr (OP == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
                R8(X)
#undef X
            } else if constexpr (OP == SNOP) {
#define X(i) asm volatile("s_nop 0");
                R8(X)
#undef X
            } else if constexpr (OP == VMOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(s[(i + 1) & 7]));
                R8(X)
#undef X
            } else if constexpr (OP == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b.x));
                R8(X)
#undef X
            } else if constexpr (OP == DPPMOV) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(s[(i + 1) & 7]));
                R8(X)
#undef X
            } else if constexpr (OP == DPPADD) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(c.x));
                R8(X)
#undef X
            } else if constexpr (OP == FMAC) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
                R8(X)
#undef X
            } else if constexpr (OP == LOGF) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(s[i]));
                R8(X)
#undef X
            } else if constexpr (OP == MIX_PK_PLAIN) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(a[i]), "+v"(s[i]) : "v"(b), "v"(c), "v"(b.x), "v"(c.x));
                R8(X)
#undef X
            } else if constexpr (OP == PKFMA_DEP) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                R8(X)
#undef X
            } else if constexpr (OP == FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[0]) : "v"(b.x), "v"(c.x));
                R8(X)
#undef X
            } else if constexpr (OP == DSR64) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[i]) : "v"(ldsaddr8), "n"(i * 512));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == DSR32) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s[i]) : "v"(ldsaddr4), "n"(i * 256));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == DSR128) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(ldsaddr), "n"(i * 1024));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == DSW64) {
#define X(i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ldsaddr8), "v"(a[i]), "n"(i * 512) : "memory");
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == DSW32) {
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ldsaddr4), "v"(s[i]), "n"(i * 256) : "memory");
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == BPERM) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(s[i]) : "v"(bp));
                R8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == MFMA16 || OP == MFMA16_V4 || OP == MFMA16_V8 || OP == MFMA16_V12 || OP == MFMA16_PK2 || OP == MFMA16_PK4 || OP == MFMA16_PK6) {
                // 8 groups per r: one MFMA (4 independent accumulators round robin) + n plain VALU
#define VF(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
#define VP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X(i)                                                                                                   \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[i & 3]) : "v"(b.x), "v"(c.y));             \
    if constexpr (OP == MFMA16_V4 || OP == MFMA16_V8 || OP == MFMA16_V12) { VF(0) VF(1) VF(2) VF(3) }            \
    if constexpr (OP == MFMA16_V8 || OP == MFMA16_V12) { VF(4) VF(5) VF(6) VF(7) }                               \
    if constexpr (OP == MFMA16_V12) { VF(0) VF(1) VF(2) VF(3) }                                                  \
    if constexpr (OP == MFMA16_PK2 || OP == MFMA16_PK4 || OP == MFMA16_PK6) { VP(0) VP(1) }                      \
    if constexpr (OP == MFMA16_PK4 || OP == MFMA16_PK6) { VP(2) VP(3) }                                          \
    if constexpr (OP == MFMA16_PK6) { VP(4) VP(5) }
                R8(X)
#undef X
            } else if constexpr (OP == MFMA32BF || OP == MFMA32BF_PK4) {
                // the bf16 matrix-core product a hi/lo-split radix-16 pass would run on (two independent 32 x 32 accumulators)
#define VP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X(i)                                                                                                   \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[i & 1]) : "v"(opa), "v"(opb));           \
    if constexpr (OP == MFMA32BF_PK4) { VP(0) VP(1) VP(2) VP(3) }
                R8(X)
#undef X
#undef VP
            } else if constexpr (OP == CVTBF) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hb[i]) : "v"(a[i].x), "v"(a[i].y));
                R8(X)
#undef X
            } else if constexpr (OP == PERM32) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[i]), "+v"(s[(i + 1) & 7]));
                R8(X)
#undef X
            } else if constexpr (OP == SPLIT2) {
                // hi = bf16(x) for a pair, back to fp32 (shift / mask), lo = bf16(x - hi): what every operand of a split product costs
#define X(i)                                                                                                   \
    {                                                                                                          \
        v2f hf;                                                                                                \
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hb[i]) : "v"(a[i].x), "v"(a[i].y));                   \
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hf.x) : "v"(hb[i]));                                      \
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hf.y) : "v"(hb[i]));                                  \
        asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(hf) : "v"(a[i]));                \
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lb2[i]) : "v"(hf.x), "v"(hf.y));                      \
    }
                R8(X)
#undef X
            } else if constexpr (OP == MFMA4 || OP == MFMA4_V2) {
#define X(i)                                                                                                   \
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4[i & 3]) : "v"(b.x), "v"(c.y));           \
    if constexpr (OP == MFMA4_V2) { VF(0) VF(1) }
                R8(X)
#undef X
#undef VF
#undef VP
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += a[i].x + a[i].y + s[i] + q[i].x + q[i].w;
#pragma unroll
    for (int i = 0; i < 4; ++i) keep += acc4[i].x + acc4[i].y;
    keep += acc16[0][0] + acc16[1][5];
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += (float)(hb[i] + lb2[i]);
    if (keep == 123.456f) sink[threadIdx.x] = keep + lds[lane];
    if (lane == 0) spans[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
int run(int ncu, unsigned long long* d_spans, float* d_sink) {
    printf("%-44s", names[OP]);
    for (int wps = 1; wps <= 4; ++wps) {
        const int grid = ncu * wps;
        bench<OP><<<grid, 256>>>(d_spans, d_sink, 10);  // warm
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        bench<OP><<<grid, 256>>>(d_spans, d_sink, ITERS);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid * 4);
        CK(hipMemcpy(h.data(), d_spans, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double per = med / ((double)body_count(OP) * ITERS * wps);
        // wall-derived clock estimate: cycles / time
        printf("  w%d %6.2f (%.2f GHz)", wps, per, med / (ms * 1e6));
    }
    printf("\n");
    return 0;
}

// power mode: one instruction kind at `wps` waves per SIMD for `sec` seconds of back-to-back launches (tools/microbench/energy.sh samples
// rocm-smi next to it); prints the wave-instruction rate of the whole chip and the clock
template <int OP>
int power(int ncu, unsigned long long* d_spans, float* d_sink, int wps, double sec) {
    const int grid = ncu * wps, iters = 20000;
    bench<OP><<<grid, 256>>>(d_spans, d_sink, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double tot_ms = 0, tot_instr = 0, cyc = 0;
    int n = 0;
    while (tot_ms < sec * 1e3) {
        CK(hipEventRecord(e0));
        bench<OP><<<grid, 256>>>(d_spans, d_sink, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tot_ms += ms;
        tot_instr += (double)grid * 4 * 32.0 * iters;   // wave-instructions (groups count as one)
        std::vector<unsigned long long> h(grid * 4);
        CK(hipMemcpy(h.data(), d_spans, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        cyc = (double)h[h.size() / 2] / (ms * 1e6);
        ++n;
    }
    printf("RESULT op=\"%s\" wps=%d seconds=%.2f wave_instr_per_s=%.4e memtime_ghz=%.3f launches=%d\n", names[OP], wps, tot_ms * 1e-3, tot_instr / (tot_ms * 1e-3), cyc, n);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    if (argc >= 5 && !strcmp(argv[1], "power")) {
        unsigned long long* d_spans;
        float* d_sink;
        CK(hipMalloc(&d_spans, ncu * 4 * 4 * 8));
        CK(hipMalloc(&d_sink, 4096));
        const int op = atoi(argv[2]), wps = atoi(argv[3]);
        const double sec = atof(argv[4]);
        switch (op) {
#define P(O) case O: return power<O>(ncu, d_spans, d_sink, wps, sec);
            P(SNOP) P(VMOV) P(FMA) P(PKFMA) P(PKADD) P(PKMUL) P(ADD) P(DPPMOV) P(LOGF) P(DSR32) P(DSR64) P(DSR128) P(DSW32) P(DSW64) P(BPERM) P(MFMA16) P(MFMA16_PK4) P(MFMA4) P(MFMA4_V2) P(MFMA32BF) P(MFMA32BF_PK4) P(CVTBF) P(PERM32) P(SPLIT2)
#undef P
            default: printf("op %d not in the power list\n", op); return 1;
        }
    }
    printf("%s, %d CUs; cycles per instruction (or per group) per SIMD, at 1..4 waves per SIMD (s_memtime span; 100 MHz const clock? see GHz column)\n", p.gcnArchName, ncu);
    unsigned long long* d_spans;
    float* d_sink;
    CK(hipMalloc(&d_spans, ncu * 4 * 4 * 8));
    CK(hipMalloc(&d_sink, 4096));
    run<FMA>(ncu, d_spans, d_sink);
    run<PKFMA>(ncu, d_spans, d_sink);
    run<PKFMA_SEL>(ncu, d_spans, d_sink);
    run<PKADD>(ncu, d_spans, d_sink);
    run<PKMUL>(ncu, d_spans, d_sink);
    run<ADD>(ncu, d_spans, d_sink);
    run<MUL>(ncu, d_spans, d_sink);
    run<FMAC>(ncu, d_spans, d_sink);
    run<DPPMOV>(ncu, d_spans, d_sink);
    run<DPPADD>(ncu, d_spans, d_sink);
    run<LOGF>(ncu, d_spans, d_sink);
    run<MIX_PK_PLAIN>(ncu, d_spans, d_sink);
    run<PKFMA_DEP>(ncu, d_spans, d_sink);
    run<FMA_DEP>(ncu, d_spans, d_sink);
    run<DSR32>(ncu, d_spans, d_sink);
    run<DSR64>(ncu, d_spans, d_sink);
    run<DSR128>(ncu, d_spans, d_sink);
    run<DSW32>(ncu, d_spans, d_sink);
    run<DSW64>(ncu, d_spans, d_sink);
    run<BPERM>(ncu, d_spans, d_sink);
    run<MFMA16>(ncu, d_spans, d_sink);
    run<MFMA16_V4>(ncu, d_spans, d_sink);
    run<MFMA16_V8>(ncu, d_spans, d_sink);
    run<MFMA16_V12>(ncu, d_spans, d_sink);
    run<MFMA16_PK2>(ncu, d_spans, d_sink);
    run<MFMA16_PK4>(ncu, d_spans, d_sink);
    run<MFMA16_PK6>(ncu, d_spans, d_sink);
    run<MFMA4>(ncu, d_spans, d_sink);
    run<MFMA4_V2>(ncu, d_spans, d_sink);
    run<MFMA32BF>(ncu, d_spans, d_sink);
    run<MFMA32BF_PK4>(ncu, d_spans, d_sink);
    run<CVTBF>(ncu, d_spans, d_sink);
    run<PERM32>(ncu, d_spans, d_sink);
    run<SPLIT2>(ncu, d_spans, d_sink);
    return 0;
}

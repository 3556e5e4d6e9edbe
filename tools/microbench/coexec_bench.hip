// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?  512-thread workgroups, one per CU: waves 0-3 run a
// VALU-only stream, waves 4-7 an MFMA-only stream (one of each per SIMD).  Reported: the span of each role alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/coexec_bench tools/microbench/coexec_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2000;
// MODE: bit0 = VALU waves active, bit1 = MFMA waves active; MF: 0 = f32 16x16x4, 1 = bf16 16x16x32, 2 = f32 32x32x2; VK: 0 v_fma, 1 v_pk_fma
template <int MF, int VK>
__global__ __launch_bounds__(512) void k(unsigned long long* spans, float* sink, int iters, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s[8]; v2f a[8];
    for (int i = 0; i < 8; ++i) { s[i] = 1.f + i + lane; a[i] = v2f{1.f + i, 2.f + lane}; }
    v4f acc[4]; v16f acc16[2];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    const float b = 1.0001f, c = 1e-7f;
    const v2f b2 = v2f{1.0001f, 0.9999f}, c2 = v2f{1e-7f, -1e-7f};
    v8s ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (short)(0x3f80 + lane + i); hb[i] = (short)(0x3f80 + i); }
    const bool valu_role = wave < 4;
    const bool active = valu_role ? (mode & 1) : (mode & 2);
    unsigned long long t0 = 0, t1 = 0;
    __builtin_amdgcn_s_barrier();
    if (active) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if (valu_role) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if constexpr (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
                        else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b2), "v"(c2));
                    }
                }
            }
        } else {
            for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if constexpr (MF == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(b), "v"(c));
                    else if constexpr (MF == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(ha), "v"(hb));
                    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16[r & 1]) : "v"(b), "v"(c));
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    float keep = 0.f;
    for (int i = 0; i < 8; ++i) keep += s[i] + a[i].x + a[i].y;
    for (int i = 0; i < 4; ++i) keep += acc[i].x + acc[i].w;
    keep += acc16[0][3] + acc16[1][7];
    if (keep == 123.456f) sink[threadIdx.x] = keep;
    if (lane == 0) spans[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MF, int VK>
int run(const char* name, int ncu, unsigned long long* d, float* sink) {
    printf("%-44s", name);
    for (int mode = 1; mode <= 3; ++mode) {
        k<MF, VK><<<ncu, 512>>>(d, sink, 8, mode);
        CK(hipDeviceSynchronize());
        k<MF, VK><<<ncu, 512>>>(d, sink, ITERS, mode);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(ncu * 8), v, m;
        CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < ncu * 8; ++i) ((i & 7) < 4 ? v : m).push_back(h[i]);
        std::sort(v.begin(), v.end()); std::sort(m.begin(), m.end());
        printf("  mode %d: VALU %7.2f cyc/instr, MFMA %7.2f cyc/instr |", mode, (double)v[v.size() / 2] / (32.0 * ITERS), (double)m[m.size() / 2] / (8.0 * (ITERS / 4)));
    }
    printf("\n");
    return 0;
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, ncu * 8 * 8)); CK(hipMalloc(&sink, 4096));
    printf("mode 1 = VALU waves alone, 2 = MFMA waves alone, 3 = both (one VALU wave + one MFMA wave per SIMD)\n");
    run<0, 0>("f32 16x16x4 MFMA  vs v_fma_f32", ncu, d, sink);
    run<0, 1>("f32 16x16x4 MFMA  vs v_pk_fma_f32", ncu, d, sink);
    run<2, 0>("f32 32x32x2 MFMA  vs v_fma_f32", ncu, d, sink);
    run<1, 0>("bf16 16x16x32 MFMA vs v_fma_f32", ncu, d, sink);
    run<1, 1>("bf16 16x16x32 MFMA vs v_pk_fma_f32", ncu, d, sink);
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* st = (float*)smem;
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) st[i] = -7.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    // piece 0: elements 0..255 (in range), piece 1: elements 256..511 (partially / fully out of range)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(st), 16, lane * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(st + 256), 16, 1024 + lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = st[i];
}
int main() {
    const int n = 300;  // elements 300..511 are out of range
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = i + 1;
    float *dx, *dout; (void)hipMalloc(&dx, 2048); (void)hipMalloc(&dout, 2048);
    (void)hipMemcpy(dx, h, 2048, hipMemcpyHostToDevice);
    k<<<1, 64, 4096>>>(dx, n, dout);
    float o[512]; (void)hipMemcpy(o, dout, 2048, hipMemcpyDeviceToHost);
    printf("o[0]=%g o[255]=%g o[256]=%g o[299]=%g o[300]=%g o[303]=%g o[304]=%g o[511]=%g\n", o[0], o[255], o[256], o[299], o[300], o[303], o[304], o[511]);
    return 0;
}

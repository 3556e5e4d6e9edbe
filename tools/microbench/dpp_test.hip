#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    int v = lane * 10;
    int ror = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);
    int shr = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false);
    out[lane] = ror;
    out[64 + lane] = shr;
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("ror:"); for (int i = 0; i < 20; ++i) printf(" %d", h[i]); printf("\nshr:"); for (int i = 0; i < 20; ++i) printf(" %d", h[64 + i]); printf("\n");
    return 0;
}

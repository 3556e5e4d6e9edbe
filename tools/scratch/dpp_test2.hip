#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, float a) {
    const int lane = threadIdx.x;
    v2f y = *reinterpret_cast<const v2f*>(in + 2 * lane);
    const float xm1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, y.x), __builtin_bit_cast(int, y.y), 0x111, 0xF, 0xF, false));
    float xm1b = xm1; asm volatile("" : "+v"(xm1b));
    const float y0 = __builtin_fmaf(a, xm1b, y.x);
    const float y1 = __builtin_fmaf(a, y.x, y.y);
    *reinterpret_cast<v2f*>(out + 2 * lane) = v2f{y0, y1};
}

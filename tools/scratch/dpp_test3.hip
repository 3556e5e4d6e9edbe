#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, float a) {
    const int lane = threadIdx.x;
    v2f y = *reinterpret_cast<const v2f*>(in + 2 * lane);
    float yx = y.x, yy = y.y;
    const float xm1 = __builtin_amdgcn_update_dpp(yx, yy, 0x111, 0xF, 0xF, false);
    out[2 * lane] = xm1;
    out[2 * lane + 1] = yy * a + yx;
}

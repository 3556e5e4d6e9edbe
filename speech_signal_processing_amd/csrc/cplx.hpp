// Packed complex arithmetic + register-resident small DFTs shared by the MFCC kernels (gfx950: v_pk_mul_f32 / v_pk_fma_f32).
#pragma once
#include <hip/hip_runtime.h>

namespace ssp {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;


// Complex helpers written so that every swizzle folds into the op_sel / constant operand of ONE packed instruction
// (v_pk_fma_f32 / v_pk_mul_f32): no v_mov / v_xor to build swapped or sign-flipped pairs.
__device__ __forceinline__ v2f swap(v2f z) { return v2f{z.y, z.x}; }
__device__ __forceinline__ v2f xx(v2f z) { return v2f{z.x, z.x}; }
__device__ __forceinline__ v2f yy(v2f z) { return v2f{z.y, z.y}; }
// t + (-i) u  and  t - (-i) u     ((-i) u = (u.y, -u.x))
__device__ __forceinline__ v2f add_neg_i(v2f t, v2f u) { return __builtin_elementwise_fma(swap(u), v2f{1.f, -1.f}, t); }
__device__ __forceinline__ v2f sub_neg_i(v2f t, v2f u) { return __builtin_elementwise_fma(swap(u), v2f{-1.f, 1.f}, t); }
// z * (c + i s), constants known at compile time
__device__ __forceinline__ v2f cmulc(v2f z, float c, float s) {
    return __builtin_elementwise_fma(yy(z), v2f{-s, c}, xx(z) * v2f{c, s});
}
// z * w in TWO packed instructions with no second copy of w: the (-w.y, w.x) operand of the second FMA is formed by the
// instruction's own half selects and neg_lo modifier (the compiler otherwise keeps swap(w) * (-1, 1) in extra VGPRs)
__device__ __forceinline__ v2f cmul(v2f z, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(z), "v"(w));                       // (z.x w.x, z.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"              // + (-z.y w.y, z.y w.x)
        : "=v"(r) : "v"(z), "v"(w), "v"(t));
    return r;
}
// (-i z) * w = (z.y w.x + z.x w.y, z.y w.y - z.x w.x), same two-instruction form
__device__ __forceinline__ v2f cmul_negi(v2f z, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(z), "v"(w));         // (z.y w.x, z.y w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]"              // + (z.x w.y, -z.x w.x)
        : "=v"(r) : "v"(z), "v"(w), "v"(t));
    return r;
}
// z * w with both w = (wr, wi) and iw = (-wi, wr) at hand (registers): two packed instructions
__device__ __forceinline__ v2f cmul2(v2f z, v2f w, v2f iw) { return __builtin_elementwise_fma(yy(z), iw, xx(z) * w); }

__device__ __forceinline__ void dft4(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f t0 = a + c, t1 = a - c, t2 = b + d, u = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = add_neg_i(t1, u);
    d = sub_neg_i(t1, u);
}

// dft4 whose fourth input is known to be zero (t2 = u = b): 6 packed instructions instead of 8
__device__ __forceinline__ void dft4_d0(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f t0 = a + c, t1 = a - c, u = b;
    a = t0 + u;
    c = t0 - u;
    b = add_neg_i(t1, u);
    d = sub_neg_i(t1, u);
}

// forward 16-point DFT, natural order in, natural order out (4 x 4 Cooley-Tukey, constant twiddles).
// TAIL0: inputs 13, 14, 15 are known to be zero (a 400-sample window in a 512-point frame) — their butterflies lose an operand
template <bool TAIL0 = false>
__device__ __forceinline__ void fft16_in(v2f (&z)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    dft4(z[0], z[4], z[8], z[12]);
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2) {  // -> a[k1][n2] at z[n2 + 4 k1]
        if (TAIL0) dft4_d0(z[n2], z[n2 + 4], z[n2 + 8], z[n2 + 12]);
        else dft4(z[n2], z[n2 + 4], z[n2 + 8], z[n2 + 12]);
    }
    // twiddle W16^(n2 k1)
    z[5] = cmulc(z[5], C1, -S1);                  // n2=1,k1=1: W^1
    z[6] = cmulc(z[6], R, -R);                    // n2=2,k1=1: W^2
    z[7] = cmulc(z[7], S1, -C1);                  // n2=3,k1=1: W^3
    z[9] = cmulc(z[9], R, -R);                    // n2=1,k1=2: W^2
    z[10] = swap(z[10]) * v2f{1.f, -1.f};         // n2=2,k1=2: W^4 = -i
    z[11] = cmulc(z[11], -R, -R);                 // n2=3,k1=2: W^6
    z[13] = cmulc(z[13], S1, -C1);                // n2=1,k1=3: W^3
    z[14] = cmulc(z[14], -R, -R);                 // n2=2,k1=3: W^6
    z[15] = cmulc(z[15], -C1, S1);                // n2=3,k1=3: W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(z[4 * k1], z[4 * k1 + 1], z[4 * k1 + 2], z[4 * k1 + 3]);  // -> X[k1 + 4 k2] at z[4 k1 + k2]
    v2f o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = z[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = o[i];
}
__device__ __forceinline__ void fft16(v2f (&z)[16]) { fft16_in<false>(z); }

// forward 8-point DFT, natural order in and out (even / odd 4-point transforms, X[k] = E[k] + W8^k O[k], X[k+4] = E[k] - W8^k O[k])
__device__ __forceinline__ void fft8(v2f (&z)[8]) {
    constexpr float R = 0.70710678118654752f;
    dft4(z[0], z[2], z[4], z[6]);
    dft4(z[1], z[3], z[5], z[7]);
    const v2f o0 = z[1], o1 = cmulc(z[3], R, -R), o2 = swap(z[5]) * v2f{1.f, -1.f}, o3 = cmulc(z[7], -R, -R);
    const v2f e0 = z[0], e1 = z[2], e2 = z[4], e3 = z[6];
    z[0] = e0 + o0;
    z[1] = e1 + o1;
    z[2] = e2 + o2;
    z[3] = e3 + o3;
    z[4] = e0 - o0;
    z[5] = e1 - o1;
    z[6] = e2 - o2;
    z[7] = e3 - o3;
}

// forward DFT of R = 2 / 4 / 8 / 16 points in registers, natural order in and out
template <int R>
__device__ __forceinline__ void fft_small(v2f (&z)[R]) {
    if constexpr (R == 2) {
        const v2f a = z[0], b = z[1];
        z[0] = a + b;
        z[1] = a - b;
    } else if constexpr (R == 4) {
        dft4(z[0], z[1], z[2], z[3]);
    } else if constexpr (R == 8) {
        fft8(z);
    } else {
        fft16(z);
    }
}

// sample pair x window pair where a ZERO weight silences whatever the sample holds (v_mul_legacy_f32: 0 . x = 0 for NaN and inf too).
// The 512-point kernels read whole 32-sample rows: the window's last row runs past its last tap into the samples behind the frame,
// and a NaN there (a corrupt recording) must not reach a frame it does not belong to — in the reference's arithmetic it does not.
// (The LLVM intrinsic, not inline assembly: the scheduler knows its latency and places it like any other multiply.)
extern "C" __device__ float ssp_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ v2f wmul_edge(v2f y, v2f w) { return v2f{ssp_fmul_legacy(y.x, w.x), ssp_fmul_legacy(y.y, w.y)}; }

// window taps of a row that takes wmul_edge (taps first_tap, first_tap + 1 of the 16 x 32 sample matrix): the legacy product must silence
// only the PADDING behind the window.  A tap of the window itself that is exactly zero (numpy.hanning's ends: the sidekit dialects) lets a
// NaN sample through in numpy (0 . NaN = NaN, the frame is NaN there): the smallest normal number stands in for such a tap — a finite
// sample times it is 1e-38 of the frame's scale, nothing; NaN and inf stay non-finite.
__device__ __forceinline__ v2f edge_row_taps(v2f w, int first_tap, int win_len) {
    if (first_tap < win_len && w.x == 0.f) w.x = 1.17549435e-38f;
    if (first_tap + 1 < win_len && w.y == 0.f) w.y = 1.17549435e-38f;
    return w;
}

}  // namespace ssp

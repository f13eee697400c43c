// Register-resident complex butterflies on PAIRS of independent transforms (device code only).
//
// gfx950 issues one VALU instruction per SIMD every 4 cycles whether it is v_add_f32 or
// v_pk_add_f32, so the f32 vector peak needs packed math.  Packing (re, im) of one complex
// number wastes instructions on swizzles; packing the same quantity of TWO independent
// transforms (A, B) does not: every scalar operation of the algorithm becomes exactly one
// v_pk_* instruction, multiplication by -i is register renaming plus a neg modifier, and
// twiddles are scalar broadcasts.
// Forward transform convention: exp(-2 pi i n k / R), natural order in and out.
#pragma once

#include <hip/hip_runtime.h>

namespace ss {

using v2f = __attribute__((ext_vector_type(2))) float;  // (transform A, transform B)

struct cx2 {
    v2f x, y;  // real parts (A, B), imaginary parts (A, B)
};

__device__ __forceinline__ cx2 cadd(cx2 a, cx2 b) { return cx2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cx2 csub(cx2 a, cx2 b) { return cx2{a.x - b.x, a.y - b.y}; }
// multiply both transforms by the same complex scalar (wx + i wy)
__device__ __forceinline__ cx2 cmul(cx2 a, float wx, float wy)
{
    return cx2{a.x * wx - a.y * wy, a.x * wy + a.y * wx};
}
__device__ __forceinline__ cx2 mul_mi(cx2 a) { return cx2{a.y, -a.x}; }  // * -i

__device__ __forceinline__ void fft4(cx2 &v0, cx2 &v1, cx2 &v2, cx2 &v3)
{
    const cx2 a0 = cadd(v0, v2), a1 = csub(v0, v2);
    const cx2 a2 = cadd(v1, v3), a3 = mul_mi(csub(v1, v3));
    v0 = cadd(a0, a2);
    v1 = cadd(a1, a3);
    v2 = csub(a0, a2);
    v3 = csub(a1, a3);
}

// 16-point DFT: n = n1 + 4 n2, k = 4 k1 + k2: W16^(nk) = W4^(n1 k1) W16^(n1 k2) W4^(n2 k2)
__device__ __forceinline__ void fft16_pk(cx2 (&v)[16])
{
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) fft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos, sin(pi/8)
    constexpr float h = 0.70710678118654752440f;
    // Y[n1][k2] (at v[n1 + 4 k2]) *= W16^(n1 k2)
    v[5] = cmul(v[5], c1, -s1);
    v[9] = cx2{(v[9].x + v[9].y) * h, (v[9].y - v[9].x) * h};
    v[13] = cmul(v[13], s1, -c1);
    v[6] = cx2{(v[6].x + v[6].y) * h, (v[6].y - v[6].x) * h};
    v[10] = mul_mi(v[10]);
    v[14] = cx2{(v[14].y - v[14].x) * h, (v[14].x + v[14].y) * -h};
    v[7] = cmul(v[7], s1, -c1);
    v[11] = cx2{(v[11].y - v[11].x) * h, (v[11].x + v[11].y) * -h};
    v[15] = cmul(v[15], -c1, s1);
    cx2 y[16];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        cx2 a = v[4 * k2], b = v[4 * k2 + 1], c = v[4 * k2 + 2], d = v[4 * k2 + 3];
        fft4(a, b, c, d);
        y[k2] = a;
        y[4 + k2] = b;
        y[8 + k2] = c;
        y[12 + k2] = d;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = y[i];
}

}  // namespace ss

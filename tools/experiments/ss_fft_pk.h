// RETIRED EXPERIMENT (round 3; was mfcc-rust_amd/csrc/ss_fft_pk.h, used by a lab build of ss_mel_c1024_w12 through -DSS_PKFFT):
// cfg3 48.6 us against 44.6 us with the scalar butterflies on one box (profiles/r03/ab_cfg3_packed_butterflies.txt) -- 266 packed
// instructions at ~5.2 issue cycles keep the SIMD busy 1.66 x as long as 390 scalar ones at ~2.1, and SIMD time is what the
// twelve-wave kernels are short of.  Results were correct (cfg3 parity tests pass).
// Packed-FP32 forms of the register butterflies (device code only; gfx950 v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on
// (re, im) register pairs).  One packed instruction does a complex add, two a complex multiply, so a butterfly is about half
// the instructions of ss_fft_reg.h -- at ~5.2 issue cycles each against ~2.1 (two waves ready) or ~4.3 (one wave ready) per
// scalar instruction: a wave gets through its butterfly sooner, the SIMD is busy longer.  Same results as the scalar forms up
// to the rounding of the twiddle products (the scalar forms fold the twiddle magnitudes into later FMAs).
#pragma once

#include <hip/hip_runtime.h>

#include "ss_fft_reg.h"

namespace ss {
namespace pk {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 ld(float2 a) { return f2{a.x, a.y}; }
__device__ __forceinline__ float2 st(f2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ f2 swap(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ f2 mul_mi(f2 a) { return __builtin_shufflevector(a, -a, 1, 2); }  // a * -i = (a.y, -a.x)
// a * (c - i s), c and s compile-time constants
__device__ __forceinline__ f2 cmulc(f2 a, float c, float s) { return __builtin_elementwise_fma(swap(a), f2{s, -s}, a * c); }

__device__ __forceinline__ void fft4(f2 &v0, f2 &v1, f2 &v2, f2 &v3)
{
    const f2 a0 = v0 + v2, a1 = v0 - v2;
    const f2 a2 = v1 + v3, a3 = mul_mi(v1 - v3);
    v0 = a0 + a2;
    v1 = a1 + a3;
    v2 = a0 - a2;
    v3 = a1 - a3;
}

__device__ __forceinline__ void fft8(f2 *v)
{
    f2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    f2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    constexpr float h = 0.70710678118654752440f;
    // W8^1 = h (1 - i): (x + y, y - x) h; W8^3 = h (-1 - i): (y - x, -(x + y)) h; the factor h rides on the last butterfly's FMAs
    const f2 p1 = o1 + mul_mi(o1);            // (x + y, y - x)
    const f2 p3 = mul_mi(o3) - o3;            // (y - x, -x - y)
    o2 = mul_mi(o2);
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    v[1] = __builtin_elementwise_fma(f2{h, h}, p1, e1);
    v[5] = __builtin_elementwise_fma(f2{-h, -h}, p1, e1);
    v[2] = e2 + o2;
    v[6] = e2 - o2;
    v[3] = __builtin_elementwise_fma(f2{h, h}, p3, e3);
    v[7] = __builtin_elementwise_fma(f2{-h, -h}, p3, e3);
}

// z * W32^M
template <int M>
__device__ __forceinline__ f2 tw32(f2 z)
{
    constexpr W32 w = w32(M);
    if constexpr ((M & 31) == 0) return z;
    else if constexpr ((M & 31) == 8) return mul_mi(z);
    else if constexpr ((M & 31) == 16) return -z;
    else if constexpr ((M & 31) == 24) return -mul_mi(z);
    else return cmulc(z, w.c, w.s);
}

// 32-point DFT, natural order in and out: n = n1 + 4 n2, k = 8 k1 + k2 (the decomposition of fft_reg<32>)
__device__ __forceinline__ void fft32(float2 *vv)
{
    f2 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = ld(vv[i]);
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
        f2 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v[n1 + 4 * i];
        fft8(t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[n1 + 4 * i] = t[i];
    }
    f2 y[32];
    auto group = [&](auto k2c) {
        constexpr int k2 = decltype(k2c)::value;
        f2 a = v[4 * k2], b = tw32<k2>(v[4 * k2 + 1]), c = tw32<2 * k2>(v[4 * k2 + 2]), d = tw32<3 * k2>(v[4 * k2 + 3]);
        fft4(a, b, c, d);
        y[k2] = a;
        y[8 + k2] = b;
        y[16 + k2] = c;
        y[24 + k2] = d;
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
    group(std::integral_constant<int, 4>{});
    group(std::integral_constant<int, 5>{});
    group(std::integral_constant<int, 6>{});
    group(std::integral_constant<int, 7>{});
#pragma unroll
    for (int i = 0; i < 32; ++i) vv[i] = st(y[i]);
}

}  // namespace pk
}  // namespace ss

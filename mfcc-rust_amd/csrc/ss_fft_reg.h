// Register-resident complex butterflies shared by the HIP kernels (device code only).
// Forward transform convention: exp(-2 pi i n k / R), natural order in and out.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace ss {

__host__ __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__host__ __device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i
__host__ __device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// Forward 4-point DFT (exp(-2 pi i nk/4)), natural order in and out.
__host__ __device__ __forceinline__ void fft4(float2 &v0, float2 &v1, float2 &v2, float2 &v3)
{
    const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2);
    const float2 a2 = cadd(v1, v3), a3 = mul_mi(csub(v1, v3));
    v0 = cadd(a0, a2);
    v1 = cadd(a1, a3);
    v2 = csub(a0, a2);
    v3 = csub(a1, a3);
}

template <int R>
__host__ __device__ __forceinline__ void fft_reg(float2 *v);

template <>
__host__ __device__ __forceinline__ void fft_reg<2>(float2 *v)
{
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <>
__host__ __device__ __forceinline__ void fft_reg<4>(float2 *v)
{
    fft4(v[0], v[1], v[2], v[3]);
}

// n = n1 + 2 n2, k = 4 k1 + k2: W8^(nk) = W2^(n1 k1) W8^(n1 k2) W4^(n2 k2)
template <>
__host__ __device__ __forceinline__ void fft_reg<8>(float2 *v)
{
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    constexpr float h = 0.70710678118654752440f;
    // W8^1 = h (1 - i), W8^3 = h (-1 - i): the factor h rides on the FMAs of the last butterfly
    o1 = make_float2(o1.x + o1.y, o1.y - o1.x);
    o2 = mul_mi(o2);  // * -i
    o3 = make_float2(o3.y - o3.x, -(o3.x + o3.y));
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = make_float2(fmaf(h, o1.x, e1.x), fmaf(h, o1.y, e1.y)); v[5] = make_float2(fmaf(-h, o1.x, e1.x), fmaf(-h, o1.y, e1.y));
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = make_float2(fmaf(h, o3.x, e3.x), fmaf(h, o3.y, e3.y)); v[7] = make_float2(fmaf(-h, o3.x, e3.x), fmaf(-h, o3.y, e3.y));
}

// n = n1 + 4 n2, k = 4 k1 + k2: W16^(nk) = W4^(n1 k1) W16^(n1 k2) W4^(n2 k2)
// The twiddles between the two radix-4 stages never appear as separate multiplies: W16^2 = h(1 - i) and
// W16^6 = h(-1 - i) leave their factor h to the FMAs of the next butterfly, and the general ones are applied as
// c (1 - i t) with the factor c folded the same way (148 instructions instead of 160; VALU issue is what bounds the
// kernels built on this).
// The same 16-point DFT, handing each output to `emit(k, X[k])` as soon as its group of four is final (k = k2, k2 + 4, k2 + 8,
// k2 + 12 after group k2) and calling `fence()` after every group: a kernel that stores the outputs to LDS can spread the
// stores over the butterfly instead of issuing all of them behind it (the LDS queue then never sees a 16-store burst).
// Forward 4-point DFT whose trailing inputs are known to be zero (zero-padded frames): Z3: v3 == 0; Z2: v2 == 0 as well.
// Written out because the compiler may not fold x + 0.0f (it would turn -0.0f into +0.0f): each dropped term is one
// instruction per component.
template <bool Z2, bool Z3>
__host__ __device__ __forceinline__ void fft4z(float2 &v0, float2 &v1, float2 &v2, float2 &v3)
{
    if constexpr (!Z3) {
        fft4(v0, v1, v2, v3);
    } else if constexpr (!Z2) {
        const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2);
        const float2 a2 = v1, a3 = mul_mi(v1);
        v0 = cadd(a0, a2);
        v1 = cadd(a1, a3);
        v2 = csub(a0, a2);
        v3 = csub(a1, a3);
    } else {
        const float2 a0 = v0, a2 = v1, a3 = mul_mi(v1);
        v0 = cadd(a0, a2);
        v1 = cadd(a0, a3);
        v2 = csub(a0, a2);
        v3 = csub(a0, a3);
    }
}

// NZ: the first NZ of the 16 inputs may be non-zero, v[NZ..15] are exactly zero (NZ >= 8: elements n1 and n1 + 4 always count)
template <int NZ = 16, class Emit, class Fence>
__device__ __forceinline__ void fft16_emit(float2 *v, Emit &&emit, Fence &&fence)
{
    static_assert(NZ >= 8 && NZ <= 16, "pruning covers the last two input quarters only");
    // step A: for each n1, 4-point DFT over n2 (elements n1, n1+4, n1+8, n1+12) -> Y[n1][k2] kept in v[n1 + 4 k2]
    fft4z<(0 + 8 >= NZ), (0 + 12 >= NZ)>(v[0], v[4], v[8], v[12]);
    fft4z<(1 + 8 >= NZ), (1 + 12 >= NZ)>(v[1], v[5], v[9], v[13]);
    fft4z<(2 + 8 >= NZ), (2 + 12 >= NZ)>(v[2], v[6], v[10], v[14]);
    fft4z<(3 + 8 >= NZ), (3 + 12 >= NZ)>(v[3], v[7], v[11], v[15]);
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos, sin(pi/8)
    constexpr float t1 = 0.41421356237309504880f, t3 = 2.41421356237309504880f;  // tan(pi/8), cot(pi/8)
    constexpr float h = 0.70710678118654752440f;
    // k2 = 0: no twiddles
    {
        float2 a = v[0], b = v[1], c = v[2], d = v[3];
        fft4(a, b, c, d);
        emit(0, a); emit(4, b); emit(8, c); emit(12, d);
    }
    fence();
    // k2 = 1: b = Y1 W16^1 = c1 b'', c = Y2 W16^2 = h c', d = Y3 W16^3 = s1 d''
    {
        const float2 a = v[4];
        const float2 bb = make_float2(fmaf(t1, v[5].y, v[5].x), fmaf(-t1, v[5].x, v[5].y));
        const float2 cc = make_float2(v[6].x + v[6].y, v[6].y - v[6].x);
        const float2 dd = make_float2(fmaf(t3, v[7].y, v[7].x), fmaf(-t3, v[7].x, v[7].y));
        const float2 a0 = make_float2(fmaf(h, cc.x, a.x), fmaf(h, cc.y, a.y));
        const float2 a1 = make_float2(fmaf(-h, cc.x, a.x), fmaf(-h, cc.y, a.y));
        const float2 p = make_float2(s1 * dd.x, s1 * dd.y);
        const float2 a2 = make_float2(fmaf(c1, bb.x, p.x), fmaf(c1, bb.y, p.y));    // b + d
        const float2 bd = make_float2(fmaf(c1, bb.x, -p.x), fmaf(c1, bb.y, -p.y));  // b - d
        emit(1, cadd(a0, a2));
        emit(5, make_float2(a1.x + bd.y, a1.y - bd.x));  // a1 - i (b - d)
        emit(9, csub(a0, a2));
        emit(13, make_float2(a1.x - bd.y, a1.y + bd.x));
    }
    fence();
    // k2 = 2: b = Y1 W16^2 = h b', c = Y2 W16^4 = -i Y2, d = Y3 W16^6 = h d'
    {
        const float2 a = v[8];
        const float2 bb = make_float2(v[9].x + v[9].y, v[9].y - v[9].x);
        const float2 c = make_float2(v[10].y, -v[10].x);
        const float2 dd = make_float2(v[11].y - v[11].x, -(v[11].x + v[11].y));
        const float2 a0 = cadd(a, c), a1 = csub(a, c);
        const float2 u1 = cadd(bb, dd), u2 = csub(bb, dd);  // (b + d) / h, (b - d) / h
        emit(2, make_float2(fmaf(h, u1.x, a0.x), fmaf(h, u1.y, a0.y)));
        emit(6, make_float2(fmaf(h, u2.y, a1.x), fmaf(-h, u2.x, a1.y)));  // a1 - i (b - d)
        emit(10, make_float2(fmaf(-h, u1.x, a0.x), fmaf(-h, u1.y, a0.y)));
        emit(14, make_float2(fmaf(-h, u2.y, a1.x), fmaf(h, u2.x, a1.y)));
    }
    fence();
    // k2 = 3: b = Y1 W16^3 = s1 b'', c = Y2 W16^6 = h c', d = Y3 W16^9 = -c1 d''
    {
        const float2 a = v[12];
        const float2 bb = make_float2(fmaf(t3, v[13].y, v[13].x), fmaf(-t3, v[13].x, v[13].y));
        const float2 cc = make_float2(v[14].y - v[14].x, -(v[14].x + v[14].y));
        const float2 dd = make_float2(fmaf(t1, v[15].y, v[15].x), fmaf(-t1, v[15].x, v[15].y));
        const float2 a0 = make_float2(fmaf(h, cc.x, a.x), fmaf(h, cc.y, a.y));
        const float2 a1 = make_float2(fmaf(-h, cc.x, a.x), fmaf(-h, cc.y, a.y));
        const float2 p = make_float2(c1 * dd.x, c1 * dd.y);
        const float2 a2 = make_float2(fmaf(s1, bb.x, -p.x), fmaf(s1, bb.y, -p.y));  // b + d
        const float2 bd = make_float2(fmaf(s1, bb.x, p.x), fmaf(s1, bb.y, p.y));    // b - d
        emit(3, cadd(a0, a2));
        emit(7, make_float2(a1.x + bd.y, a1.y - bd.x));
        emit(11, csub(a0, a2));
        emit(15, make_float2(a1.x - bd.y, a1.y + bd.x));
    }
    fence();
}

template <>
__host__ __device__ __forceinline__ void fft_reg<16>(float2 *v)
{
    // step A: for each n1, 4-point DFT over n2 (elements n1, n1+4, n1+8, n1+12) -> Y[n1][k2] kept in v[n1 + 4 k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) fft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos, sin(pi/8)
    constexpr float t1 = 0.41421356237309504880f, t3 = 2.41421356237309504880f;  // tan(pi/8), cot(pi/8)
    constexpr float h = 0.70710678118654752440f;
    float2 y[16];
    // k2 = 0: no twiddles
    {
        float2 a = v[0], b = v[1], c = v[2], d = v[3];
        fft4(a, b, c, d);
        y[0] = a; y[4] = b; y[8] = c; y[12] = d;
    }
    // k2 = 1: b = Y1 W16^1 = c1 b'', c = Y2 W16^2 = h c', d = Y3 W16^3 = s1 d''
    {
        const float2 a = v[4];
        const float2 bb = make_float2(fmaf(t1, v[5].y, v[5].x), fmaf(-t1, v[5].x, v[5].y));
        const float2 cc = make_float2(v[6].x + v[6].y, v[6].y - v[6].x);
        const float2 dd = make_float2(fmaf(t3, v[7].y, v[7].x), fmaf(-t3, v[7].x, v[7].y));
        const float2 a0 = make_float2(fmaf(h, cc.x, a.x), fmaf(h, cc.y, a.y));
        const float2 a1 = make_float2(fmaf(-h, cc.x, a.x), fmaf(-h, cc.y, a.y));
        const float2 p = make_float2(s1 * dd.x, s1 * dd.y);
        const float2 a2 = make_float2(fmaf(c1, bb.x, p.x), fmaf(c1, bb.y, p.y));    // b + d
        const float2 bd = make_float2(fmaf(c1, bb.x, -p.x), fmaf(c1, bb.y, -p.y));  // b - d
        y[1] = cadd(a0, a2);
        y[5] = make_float2(a1.x + bd.y, a1.y - bd.x);  // a1 - i (b - d)
        y[9] = csub(a0, a2);
        y[13] = make_float2(a1.x - bd.y, a1.y + bd.x);
    }
    // k2 = 2: b = Y1 W16^2 = h b', c = Y2 W16^4 = -i Y2, d = Y3 W16^6 = h d'
    {
        const float2 a = v[8];
        const float2 bb = make_float2(v[9].x + v[9].y, v[9].y - v[9].x);
        const float2 c = make_float2(v[10].y, -v[10].x);
        const float2 dd = make_float2(v[11].y - v[11].x, -(v[11].x + v[11].y));
        const float2 a0 = cadd(a, c), a1 = csub(a, c);
        const float2 u1 = cadd(bb, dd), u2 = csub(bb, dd);  // (b + d) / h, (b - d) / h
        y[2] = make_float2(fmaf(h, u1.x, a0.x), fmaf(h, u1.y, a0.y));
        y[6] = make_float2(fmaf(h, u2.y, a1.x), fmaf(-h, u2.x, a1.y));  // a1 - i (b - d)
        y[10] = make_float2(fmaf(-h, u1.x, a0.x), fmaf(-h, u1.y, a0.y));
        y[14] = make_float2(fmaf(-h, u2.y, a1.x), fmaf(h, u2.x, a1.y));
    }
    // k2 = 3: b = Y1 W16^3 = s1 b'', c = Y2 W16^6 = h c', d = Y3 W16^9 = -c1 d''
    {
        const float2 a = v[12];
        const float2 bb = make_float2(fmaf(t3, v[13].y, v[13].x), fmaf(-t3, v[13].x, v[13].y));
        const float2 cc = make_float2(v[14].y - v[14].x, -(v[14].x + v[14].y));
        const float2 dd = make_float2(fmaf(t1, v[15].y, v[15].x), fmaf(-t1, v[15].x, v[15].y));
        const float2 a0 = make_float2(fmaf(h, cc.x, a.x), fmaf(h, cc.y, a.y));
        const float2 a1 = make_float2(fmaf(-h, cc.x, a.x), fmaf(-h, cc.y, a.y));
        const float2 p = make_float2(c1 * dd.x, c1 * dd.y);
        const float2 a2 = make_float2(fmaf(s1, bb.x, -p.x), fmaf(s1, bb.y, -p.y));  // b + d
        const float2 bd = make_float2(fmaf(s1, bb.x, p.x), fmaf(s1, bb.y, p.y));    // b - d
        y[3] = cadd(a0, a2);
        y[7] = make_float2(a1.x + bd.y, a1.y - bd.x);
        y[11] = csub(a0, a2);
        y[15] = make_float2(a1.x - bd.y, a1.y + bd.x);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = y[i];
}


// W32^m = cos(2 pi m / 32) - i sin(2 pi m / 32) from the first-octant table
struct W32 {
    float c, s;
};
__host__ __device__ constexpr W32 w32(int m)
{
    constexpr float c32[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                              0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f};
    m &= 31;
    if (m <= 8) return W32{c32[m], c32[8 - m]};
    if (m <= 16) return W32{-c32[16 - m], c32[m - 8]};
    if (m <= 24) return W32{-c32[m - 16], -c32[24 - m]};
    return W32{c32[32 - m], -c32[m - 24]};
}
// z * W32^M = scale * rot(z): the rotation costs two FMAs (or nothing for the axis angles); `scale` is left to the caller's
// butterfly FMAs.  |cos| >= |sin|: W = c (1 - i t), t = s/c; otherwise W = s (t - i), t = c/s.
template <int M>
__host__ __device__ __forceinline__ float2 rot32(float2 z, float &scale)
{
    constexpr W32 w = w32(M);
    if constexpr ((M & 7) == 0) {
        scale = 1.0f;
        if constexpr ((M & 31) == 0) return z;
        else if constexpr ((M & 31) == 8) return make_float2(z.y, -z.x);
        else if constexpr ((M & 31) == 16) return make_float2(-z.x, -z.y);
        else return make_float2(-z.y, z.x);
    } else if constexpr ((w.c < 0 ? -w.c : w.c) >= (w.s < 0 ? -w.s : w.s)) {
        constexpr float t = w.s / w.c;
        scale = w.c;
        return make_float2(fmaf(t, z.y, z.x), fmaf(-t, z.x, z.y));
    } else {
        constexpr float t = w.c / w.s;
        scale = w.s;
        return make_float2(fmaf(t, z.x, z.y), fmaf(t, z.y, -z.x));
    }
}

// 32-point DFT: n = n1 + 4 n2 (n1 < 4, n2 < 8), k = 8 k1 + k2: W32^(nk) = W4^(n1 k1) W32^(n1 k2) W8^(n2 k2)
template <>
__host__ __device__ __forceinline__ void fft_reg<32>(float2 *v)
{
    // step A: for each n1 an 8-point DFT over n2 (elements n1 + 4 n2) -> Y[n1][k2] written back to v[n1 + 4 k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
        float2 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v[n1 + 4 * i];
        fft_reg<8>(t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[n1 + 4 * i] = t[i];
    }
    // steps B + C: for each k2 the twiddles W32^(n1 k2) and the 4-point DFT over n1 (elements 4 k2 + n1) -> X[8 k1 + k2];
    // the twiddle magnitudes are folded into the butterfly's FMAs
    float2 y[32];
    auto group = [&](auto k2c) {
        constexpr int k2 = decltype(k2c)::value;
        float sb, sc, sd;
        const float2 a = v[4 * k2];
        const float2 b = rot32<k2>(v[4 * k2 + 1], sb);
        const float2 c = rot32<2 * k2>(v[4 * k2 + 2], sc);
        const float2 d = rot32<3 * k2>(v[4 * k2 + 3], sd);
        const float2 a0 = make_float2(fmaf(sc, c.x, a.x), fmaf(sc, c.y, a.y));
        const float2 a1 = make_float2(fmaf(-sc, c.x, a.x), fmaf(-sc, c.y, a.y));
        const float2 p = make_float2(sd * d.x, sd * d.y);
        const float2 a2 = make_float2(fmaf(sb, b.x, p.x), fmaf(sb, b.y, p.y));    // b + d
        const float2 bd = make_float2(fmaf(sb, b.x, -p.x), fmaf(sb, b.y, -p.y));  // b - d
        y[k2] = cadd(a0, a2);
        y[8 + k2] = make_float2(a1.x + bd.y, a1.y - bd.x);  // a1 - i (b - d)
        y[16 + k2] = csub(a0, a2);
        y[24 + k2] = make_float2(a1.x - bd.y, a1.y + bd.x);
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
    for (int i = 0; i < 32; ++i) v[i] = y[i];
}

// 16-point DFT on a 16-element register array
__host__ __device__ __forceinline__ void fft16_reg(float2 (&v)[16]) { fft_reg<16>(v); }

}  // namespace ss

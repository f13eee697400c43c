// Wave-level device helpers shared by the dedicated kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace ss {
namespace wv {

constexpr float kEps = 1.1920929e-7f;   // f32::EPSILON, functions.rs:70
constexpr float kTwo32 = 4294967296.f;  // 2^32: see ln_scaled

// Wave-private LDS hand-off: the hardware keeps one wave's LDS operations in order; this only stops the compiler from
// reordering the accesses around the point.
__device__ __forceinline__ void wave_order()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ds_bpermute_b32: every lane reads `v` of the lane whose number is addr / 4 (LDS crossbar, no memory round trip)
__device__ __forceinline__ float bperm(int addr, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// ln(x) for a value handed over as x * 2^32 (the factor rides on the scale multiply that produced it): no denormal test is
// needed, v_log_f32 sees a normal number for every non-zero f32 x.
__device__ __forceinline__ float ln_scaled(float xs)
{
    return fmaf(__builtin_amdgcn_logf(xs), 0.69314718055994530942f, -32.f * 0.69314718055994530942f);
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row; every lane ends with the same bits
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);  // row_half_mirror
    v += dpp<0x140>(v);  // row_mirror
    return v;
}

// sum over the 32 lanes of a half-wave; every lane of the half ends with the same bits
__device__ __forceinline__ float half_sum(float v)
{
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// sum over the wave
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

}  // namespace wv
}  // namespace ss

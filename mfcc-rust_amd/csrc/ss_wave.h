// Wave-level device helpers shared by the dedicated kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "ss_device.h"

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

// Workgroup barrier that orders LDS accesses only.  __syncthreads() also drains the wave's outstanding global loads
// (s_waitcnt vmcnt(0) in front of s_barrier), which would make every wave wait for the slowest wave's first samples.
__device__ __forceinline__ void wg_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Output stores that the compiler can COUNT.  vmcnt retires loads and stores together, in issue order (gfx9 family), and a store
// is acknowledged only when it has reached L2 -- hundreds of nanoseconds under load.  A store inside `if (lane is valid)` sits
// behind a branch, so the compiler cannot know whether it was issued and every later wait for the next unit's prefetched samples
// becomes s_waitcnt vmcnt(0): the wave waits for its own store's acknowledgement in every iteration.  A buffer store through a
// resource descriptor is issued unconditionally by every lane (straight-line code: the wait for the samples becomes
// vmcnt(<stores behind them>) and the stores stay in flight); lanes that must not write pass kOobOffset and the range check
// of the descriptor drops them (offset >= num_records), as it drops rows beyond `bytes`.
constexpr int kOobOffset = 0x7fffffff;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(const void *base, unsigned bytes)
{
    // word 3: DATA_FORMAT = 32 (gfx9 / CDNA raw buffer), stride 0: byte offsets, range-checked against num_records
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, int byte_off)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 0);
}
// AUX: cache-policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); 0 = the default write-back policy
template <int AUX = 0>
__device__ __forceinline__ void buf_store(float2 v, __amdgpu_buffer_rsrc_t r, int byte_off)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 d;
    d.x = __builtin_bit_cast(unsigned, v.x);
    d.y = __builtin_bit_cast(unsigned, v.y);
    __builtin_amdgcn_raw_buffer_store_b64(d, r, byte_off, 0, AUX);
}
__device__ __forceinline__ void buf_store(float4 v, __amdgpu_buffer_rsrc_t r, int byte_off)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 d;
    d.x = __builtin_bit_cast(unsigned, v.x);
    d.y = __builtin_bit_cast(unsigned, v.y);
    d.z = __builtin_bit_cast(unsigned, v.z);
    d.w = __builtin_bit_cast(unsigned, v.w);
    __builtin_amdgcn_raw_buffer_store_b128(d, r, byte_off, 0, 0);
}

// MULTI builds (a BatchTable as second kernel argument, ss_device.h): which batch a unit of the launch's concatenated unit range
// belongs to.  Scalar work (the unit index is uniform): a wave keeps the batch of its current unit (and of the unit it prefetches)
// in SGPRs and looks the table up again only when a claimed unit lies past that batch's last one -- units are claimed in increasing
// order, so that is once per batch boundary a wave crosses.  Inside a batch everything is the single-batch arithmetic on the
// batch-local unit index: results are bit-identical to one launch per batch.
struct Seg {
    const float *x;
    float *out;
    unsigned u0, u1, total;  // the batch's units are [u0, u1) of the launch; total = its clips * n_frames
};
__device__ __forceinline__ Seg seg_of(const BatchTable &m, unsigned g)  // g: uniform
{
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < kMaxLaunchBatches - 1; ++k) s += g >= m.uend[k] ? 1u : 0u;  // (entries past the last batch hold 0xffffffff)
    Seg r;
    r.x = m.x[s];
    r.out = m.out[s];
    r.u1 = m.uend[s];
    r.u0 = s ? m.uend[s - 1] : 0u;
    r.total = m.total[s];
    return r;
}

// Wave-lifetime stamps (the *_timed_region diagnostics): a wave notes the shader-cycle counter and the constant 100 MHz counter
// as it starts and writes the two differences into its record as it ends -- scalar work outside the main loop, nothing when the
// pointer is null (every ordinary launch).
struct LifeStamp {
    unsigned long long c0, t0;
};
__device__ __forceinline__ LifeStamp life_begin(const unsigned long long *stamps)
{
    LifeStamp s{0ull, 0ull};
    if (stamps) {
        s.c0 = __builtin_amdgcn_s_memtime();
        s.t0 = __builtin_amdgcn_s_memrealtime();
    }
    return s;
}
__device__ __forceinline__ void life_end(unsigned long long *stamps, const LifeStamp &s, unsigned wave_index)
{
    if (stamps && (threadIdx.x & 63) == 0) {
        stamps[2ull * wave_index] = __builtin_amdgcn_s_memtime() - s.c0;
        stamps[2ull * wave_index + 1] = __builtin_amdgcn_s_memrealtime() - s.t0;
    }
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

// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b`.  Inline assembly: hipcc 7.2 drops the
// second result of __builtin_amdgcn_permlane32_swap (both extracts read the first register).  The s_nop covers the
// VALU-write -> permlane-read hazard, which the assembler does not see.
__device__ __forceinline__ void swap_halves(float &a, float &b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// A register that the compiler has to treat as defined, at no cost: an empty asm statement "writes" it.  For arrays that are
// filled under complementary lane masks (the two half-wave phases of the transposing exchange): left half-defined, the
// "undefined" halves are carried around the loop as if they were values and spilled.
__device__ __forceinline__ float defined_garbage()
{
    float v;
    asm volatile("" : "=v"(v));
    return v;
}

// sum over the wave without an LDS round trip: four DPP adds give every lane its row's sum, the four row sums are read as
// scalars and added; every lane ends with the same bits.  (The __shfl_xor form below is six dependent ds_bpermute round trips:
// ~1200 cycles of a wave's time where only two waves share a SIMD.)
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v = row16_sum(v);
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (s0 + s1) + (s2 + s3);
}

// sum over the 32 lanes of a half-wave (lanes 0..31 / 32..63) the same way; every lane of the half ends with the same bits
__device__ __forceinline__ float half_sum_dpp(float v, int half)
{
    v = row16_sum(v);
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return half ? s2 + s3 : s0 + s1;
}

// One lane's mel slots with compile-time tap counts (float4 units per slot): every weight and every P tap of all four slots is
// requested before the first FMA -- one LDS wait for the stage instead of one per chunk of the run-time loops below (with two
// waves per SIMD nobody hides those round trips).  w4: the lane's weight row; p0..p3: the P row at each slot's first bin.
template <int Q0, int Q1, int Q2, int Q3>
__device__ __forceinline__ void mel4_fixed(const float4 *w4, const float4 *p0, const float4 *p1, const float4 *p2, const float4 *p3, float (&m)[4])
{
    constexpr int QT = Q0 + Q1 + Q2 + Q3;
    float4 w[QT], t[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) w[i] = w4[i];
#pragma unroll
    for (int i = 0; i < Q0; ++i) t[i] = p0[i];
#pragma unroll
    for (int i = 0; i < Q1; ++i) t[Q0 + i] = p1[i];
#pragma unroll
    for (int i = 0; i < Q2; ++i) t[Q0 + Q1 + i] = p2[i];
#pragma unroll
    for (int i = 0; i < Q3; ++i) t[Q0 + Q1 + Q2 + i] = p3[i];
    constexpr int lo[5] = {0, Q0, Q0 + Q1, Q0 + Q1 + Q2, QT};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float acc = 0.f;
#pragma unroll
        for (int i = lo[s]; i < lo[s + 1]; ++i) {
            acc = fmaf(w[i].x, t[i].x, acc);
            acc = fmaf(w[i].y, t[i].y, acc);
            acc = fmaf(w[i].z, t[i].z, acc);
            acc = fmaf(w[i].w, t[i].w, acc);
        }
        m[s] = acc;
    }
}

// The same in two groups (slot 0, then slots 1..3): two LDS waits, half the registers in flight -- for kernels that hold a
// prefetched unit in registers across the mel stage.
template <int Q0, int Q1, int Q2, int Q3>
__device__ __forceinline__ void mel4_fixed2(const float4 *w4, const float4 *p0, const float4 *p1, const float4 *p2, const float4 *p3, float (&m)[4])
{
    {
        float4 w[Q0], t[Q0];
#pragma unroll
        for (int i = 0; i < Q0; ++i) {
            w[i] = w4[i];
            t[i] = p0[i];
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < Q0; ++i) {
            acc = fmaf(w[i].x, t[i].x, acc);
            acc = fmaf(w[i].y, t[i].y, acc);
            acc = fmaf(w[i].z, t[i].z, acc);
            acc = fmaf(w[i].w, t[i].w, acc);
        }
        m[0] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int QR = Q1 + Q2 + Q3;
    float4 w[QR], t[QR];
#pragma unroll
    for (int i = 0; i < QR; ++i) w[i] = w4[Q0 + i];
#pragma unroll
    for (int i = 0; i < Q1; ++i) t[i] = p1[i];
#pragma unroll
    for (int i = 0; i < Q2; ++i) t[Q1 + i] = p2[i];
#pragma unroll
    for (int i = 0; i < Q3; ++i) t[Q1 + Q2 + i] = p3[i];
    constexpr int lo[4] = {0, Q1, Q1 + Q2, QR};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float acc = 0.f;
#pragma unroll
        for (int i = lo[s]; i < lo[s + 1]; ++i) {
            acc = fmaf(w[i].x, t[i].x, acc);
            acc = fmaf(w[i].y, t[i].y, acc);
            acc = fmaf(w[i].z, t[i].z, acc);
            acc = fmaf(w[i].w, t[i].w, acc);
        }
        m[1 + s] = acc;
    }
}

// sum over the wave
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Fused pre-emphasis (processing.rs:31-53, np.roll semantics over the clip): the tap x[(pos - sh) mod L] of sample `pos`, sh < L
__device__ __forceinline__ float preemph_tap(const float *xc, int pos, unsigned sh, unsigned len)
{
    const unsigned p = static_cast<unsigned>(pos);
    return xc[p >= sh ? p - sh : p + len - sh];
}

// One mel slot of a lane: q4 float4s of weights against the same span of the P row, both 16-byte aligned (the host rounds a
// filter's first bin down to a multiple of 4).  Four float4 pairs are requested per wait, so the loop is not one LDS round trip
// per four taps.
__device__ __forceinline__ float mel_slot4(const float4 *w4, const float4 *p4, int q4)
{
    float acc = 0.f;
    int i = 0;
    for (; i + 4 <= q4; i += 4) {
        float4 w[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            w[u] = w4[i + u];
            t[u] = p4[i + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = fmaf(w[u].x, t[u].x, acc);
            acc = fmaf(w[u].y, t[u].y, acc);
            acc = fmaf(w[u].z, t[u].z, acc);
            acc = fmaf(w[u].w, t[u].w, acc);
        }
    }
    // the remainder (1..3 pairs) in ONE batch as well: a pair at a time it was one exposed LDS round trip per pair
    const int rem = q4 - i;
    if (rem > 0) {
        float4 w[3], t[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u < rem) {
                w[u] = w4[i + u];
                t[u] = p4[i + u];
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u < rem) {
                acc = fmaf(w[u].x, t[u].x, acc);
                acc = fmaf(w[u].y, t[u].y, acc);
                acc = fmaf(w[u].z, t[u].z, acc);
                acc = fmaf(w[u].w, t[u].w, acc);
            }
        }
    }
    return acc;
}

// Same for a P row span that starts at any bin (the taps load as single words): two weight / tap groups per LDS wait.
__device__ __forceinline__ float mel_slot1(const float4 *w4, const float *p, int q4)
{
    float acc = 0.f;
    int i = 0;
    for (; i + 2 <= q4; i += 2) {
        const float4 w0 = w4[i], w1 = w4[i + 1];
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[4 * i + u];
        acc = fmaf(w0.x, t[0], acc);
        acc = fmaf(w0.y, t[1], acc);
        acc = fmaf(w0.z, t[2], acc);
        acc = fmaf(w0.w, t[3], acc);
        acc = fmaf(w1.x, t[4], acc);
        acc = fmaf(w1.y, t[5], acc);
        acc = fmaf(w1.z, t[6], acc);
        acc = fmaf(w1.w, t[7], acc);
    }
    if (i < q4) {
        const float4 w0 = w4[i];
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = p[4 * i + u];
        acc = fmaf(w0.x, t[0], acc);
        acc = fmaf(w0.y, t[1], acc);
        acc = fmaf(w0.z, t[2], acc);
        acc = fmaf(w0.w, t[3], acc);
    }
    return acc;
}

}  // namespace wv
}  // namespace ss

// ss_mfcc_c256_mx: fused MFCC for fft_points = 512 on gfx950 -- second-generation mapping.
//
// A wave owns a contiguous run of frame quads.  Per quad (4 frames, 16 lanes = one DPP row each):
//   A1  10 x 8-byte coalesced loads per lane (128 B per frame row); next quad prefetched.
//   A2  radix-16 register butterfly with the zero padding folded at compile time.
//   A3  ONE transposing exchange through wave-private LDS: ds_write_b64 scatter into a layout
//       whose read side is 8 x ds_read_b128 per lane (element (n1,k1) at 34*(n1>>1) + 2*k1 + (n1&1),
//       frame stride 2304 B = 9 bank rows: conflict-free on both sides).
//   A4  twiddle + second radix-16 butterfly: lane j now holds Z[j + 16 r].
//   A5  the real-FFT untangle needs Z[256-k], which lives in lane 16-j, register 15-r: fetched
//       with ds_bpermute_b32 (LDS crossbar only -- no second LDS round trip).
//   A6  |X|/N; bins 0..128 go to the wave's P tile [16 frames x 130]; all 257 feed the frame
//       energy, reduced over the DPP row.
// After four quads (16 frames) the wave runs the two small contractions of the reference
// (feature.rs:229 P.fb^T and :123 DCT-II) on the matrix pipe, which is otherwise idle:
//   B1  mel^T[filter][frame] = sum_bin W[filter][bin] P[frame][bin] as v_mfma_f32_16x16x4_f32 over
//       ONLY the non-zero 16-filter x 4-bin blocks of the banded bank (35 of 99 at the defaults):
//       a block-sparse product, exact f32 (each MFMA is an fmaf chain).
//   B2  zero handling + ln on the 12 accumulator registers.
//   B3  DCT: out^T[ceps][frame] = sum_filter cos[ceps][filter] L[filter][frame]; the accumulator
//       registers of B1 ARE the B operand of B3 (the contraction runs over B1's row index), so no
//       data moves between the two products.
//   B4  scaling, column-0 replacement, staging through LDS, coalesced store of 16 x n_ceps floats.
// No workgroup barrier in the main loop: all exchanges are wave-private and LDS operations of one
// wave execute in order.
//
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.
#include "ss_device.h"
#include "ss_fft_reg.h"

namespace ss {

namespace {

constexpr float kEpsM = 1.1920929e-7f;   // f32::EPSILON, functions.rs:70
constexpr int kWaves = 8;                // waves per workgroup (512 threads, one workgroup per CU)
constexpr int kZStride = 288;            // float2 per frame exchange region (2304 B)
constexpr int kPPitch = 132;             // floats per P row: bins 0..128 + 3 zero pad bins read by the last k-step
constexpr int kWaveFloats = 4 * kZStride * 2 + 16 * kPPitch + 16;  // zbuf | P tile | ln(energy)
constexpr int kWaveBytes = ((kWaveFloats * 4 + 255) / 256) * 256;

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float row16_sum_m(float v)
{
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}

// Wave-private LDS hand-off: LDS operations of one wave execute in order, so all that is needed is
// that the compiler keeps the program order of the accesses around this point.
__device__ __forceinline__ void wave_sync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float bperm(int addr, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

template <int NE, bool EXACT>
__device__ __forceinline__ void load_quad(const Fast512MArgs &a, unsigned quad, unsigned total, int f, int j, float2 (&vin)[NE])
{
    unsigned gf = quad * 4 + f;
    gf = gf < total ? gf : total - 1;
    const unsigned clip = gf / a.n_frames;
    const unsigned t = gf - clip * a.n_frames;
    // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
    const float2 *src = reinterpret_cast<const float2 *>(a.x + static_cast<unsigned long long>(clip) * a.ld + t * a.step);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int n = j + 16 * e;
        if (EXACT) vin[e] = src[n];
        else vin[e] = 2 * n < static_cast<int>(a.flen) ? src[n] : make_float2(0.f, 0.f);
    }
}

template <int NE, bool EXACT>
__global__ __launch_bounds__(kWaves * 64) void ss_mfcc_c256_mx(const Fast512MArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;  // frame within the quad
    const int j = lane & 15;  // lane within the frame (DPP row)

    // ---- LDS carve ----
    float *wbase = reinterpret_cast<float *>(smem + wave * kWaveBytes);
    float2 *zfr = reinterpret_cast<float2 *>(wbase) + f * kZStride;  // this frame's exchange region
    float *ptile = wbase + 4 * kZStride * 2;                         // [16][132]
    float *elog = ptile + 16 * kPPitch;                              // ln(frame energy) [16]
    float *s_wt = reinterpret_cast<float *>(smem + kWaves * kWaveBytes);  // mel MFMA A operands [n_mm][64]
    float *s_ct = s_wt + a.n_mm * 64;                                     // DCT MFMA A operands [12][64]

    for (int i = tid; i < a.n_mm * 64; i += kWaves * 64) s_wt[i] = a.wt[i];
    for (int i = tid; i < 12 * 64; i += kWaves * 64) s_ct[i] = a.ct[i];
    for (int i = lane; i < 16 * kPPitch; i += 64) ptile[i] = 0.f;  // pad bins 129..131 of every row stay zero for good

    // per-lane constants, live in registers for the whole kernel
    float2 tw2[15];  // exp(-2 pi i j r / 256), r = 1..15
#pragma unroll
    for (int r = 1; r < 16; ++r) tw2[r - 1] = a.tw_c[j * r];
    float2 twn[8];   // exp(-2 pi i (j + 16 r) / 512)
#pragma unroll
    for (int r = 0; r < 8; ++r) twn[r] = a.tw_n[j + 16 * r];
    const int partner = (lane & 48) | ((16 - j) & 15);  // lane holding Z[256 - k]
    const int paddr = partner << 2;
    const int wbase1 = 34 * (j >> 1) + (j & 1);  // exchange write base (float2 units)
    __syncthreads();

    const unsigned total = a.batch * a.n_frames;
    const unsigned quads = (total + 3) / 4;
    // contiguous quad range of this wave (balanced to within one quad)
    const unsigned wid = blockIdx.x * kWaves + wave, nw = gridDim.x * kWaves;
    const unsigned qlo = static_cast<unsigned>(static_cast<unsigned long long>(quads) * wid / nw);
    const unsigned qhi = static_cast<unsigned>(static_cast<unsigned long long>(quads) * (wid + 1) / nw);
    const int Cc = static_cast<int>(a.n_ceps);

    float2 vin[NE];
    if (qlo < qhi) load_quad<NE, EXACT>(a, qlo, total, f, j, vin);

    for (unsigned q0 = qlo; q0 < qhi; q0 += 4) {
        const unsigned nq = min(4u, qhi - q0);
        for (unsigned qi = 0; qi < nq; ++qi) {
            const unsigned q = q0 + qi;
            float2 v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = e < NE ? vin[e] : make_float2(0.f, 0.f);  // zero pad, processing.rs:147-156
            if (q + 1 < qhi) load_quad<NE, EXACT>(a, q + 1, total, f, j, vin);              // prefetch

            // ---- 256-point complex FFT ----
            fft16_reg(v);
#pragma unroll
            for (int r = 0; r < 16; ++r) zfr[wbase1 + 2 * r] = v[r];
            wave_sync();
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&zfr[34 * p + 2 * j]);
                v[2 * p] = make_float2(t4.x, t4.y);
                v[2 * p + 1] = make_float2(t4.z, t4.w);
            }
            wave_sync();
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], tw2[r - 1]);
            fft16_reg(v);  // v[r] = Z[j + 16 r]

            // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
            float esum = 0.f;
            float *prow = ptile + (qi * 4 + f) * kPPitch;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 zk = v[r];
                // partner register 15 - r; lane 0 pairs with itself: Z[256 - 16 r] = own register (16 - r) & 15
                float2 zc = make_float2(bperm(paddr, v[15 - r].x), bperm(paddr, v[15 - r].y));
                if (j == 0) zc = v[(16 - r) & 15];
                const float2 w = twn[r];
                const float2 s = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
                const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
                const float2 wd = cmul(w, d);
                const float xa_r = s.x + wd.y, xa_i = s.y - wd.x;  // X[k]
                const float xb_r = s.x - wd.y, xb_i = s.y + wd.x;  // conj X[256-k]
                const float ma = __builtin_amdgcn_sqrtf(xa_r * xa_r + xa_i * xa_i);
                const float mb = __builtin_amdgcn_sqrtf(xb_r * xb_r + xb_i * xb_i);
                const float pa = a.spectrum_exponent == 2 ? a.scale * (ma * ma) : a.scale * ma;
                const float pb = a.spectrum_exponent == 2 ? a.scale * (mb * mb) : a.scale * mb;
                prow[j + 16 * r] = pa;  // only bins <= 128 can carry mel weight (bank ends at (F+1)/2, feature.rs:69-70)
                esum += pa + pb;
            }
            if (j == 0) {
                const float2 z = v[8];  // X[128] = conj Z[128]
                const float m = __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);
                const float p128 = a.spectrum_exponent == 2 ? a.scale * (m * m) : a.scale * m;
                prow[128] = p128;
                // lane 0's pair (k = 0) produced X[0] and X[256]; X[128] is the one extra bin
                esum += p128;
            }
            float energy = row16_sum_m(esum);
            energy = energy == 0.f ? kEpsM : energy;  // zero_handling, feature.rs:219
            if (j == 0) elog[qi * 4 + f] = __logf(energy);
        }
        wave_sync();

        // ---- B1: block-sparse mel product on the matrix pipe (feature.rs:229) ----
        f32x4 acc[3];
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *pb = ptile + (lane & 15) * kPPitch + (lane >> 4);
        const float *wt = s_wt + lane;
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            for (int s = a.ks_lo[tl]; s < a.ks_hi[tl]; ++s) {
                acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(*wt, pb[4 * s], acc[tl], 0, 0, 0);
                wt += 64;
            }
        }
        // ---- B2: zero handling (feature.rs:230) + ln (:105);  B3: DCT-II (:120-123) ----
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[tl][i];
                x = x == 0.f ? kEpsM : x;
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(s_ct[(tl * 4 + i) * 64 + lane], __logf(x), o, 0, 0, 0);
            }
        }
        // ---- B4: scaling + column-0 replacement (feature.rs:126-146), staged coalesced store ----
        wave_sync();
        float *stage = wbase;  // the exchange regions are idle during phase B
        {
            const int fr = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * g + i;
                float val = o[i] * a.dct_scale_k;
                if (c == 0) {
                    if (a.dc_elimination) {
                        val = elog[fr];
                    } else {
                        const unsigned gfr = min(q0 * 4 + fr, total - 1);
                        const unsigned t = gfr % a.n_frames;
                        val = o[i] * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                    }
                }
                if (c < Cc) stage[fr * Cc + c] = val;
            }
        }
        wave_sync();
        {
            const unsigned first = q0 * 4;
            const unsigned nfr = min(nq * 4, total - first);
            const int nout = static_cast<int>(nfr) * Cc;
            float *dst = a.out + static_cast<unsigned long long>(first) * Cc;
            for (int i = lane; i < nout; i += 64) dst[i] = stage[i];
        }
        wave_sync();
    }
}

}  // namespace

hipError_t launch_mfcc_c256_mx(const Fast512MArgs &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = static_cast<size_t>(kWaves) * kWaveBytes + static_cast<size_t>(a.n_mm + 12) * 64 * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    const unsigned long long quads = (total + 3) / 4;
    // one 8-wave workgroup per CU; fewer when there is not at least one quad per wave
    unsigned long long blocks = (quads + kWaves - 1) / kWaves;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    const bool exact10 = a.flen == 320, full = a.flen == 512;
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(kWaves * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), lds, stream, a);
        return hipGetLastError();
    };
    if (exact10) return go(ss_mfcc_c256_mx<10, true>, "ss_mfcc_c256_mx<10,true>");
    if (full) return go(ss_mfcc_c256_mx<16, true>, "ss_mfcc_c256_mx<16,true>");
    if (a.flen <= 320) return go(ss_mfcc_c256_mx<10, false>, "ss_mfcc_c256_mx<10,false>");
    return go(ss_mfcc_c256_mx<16, false>, "ss_mfcc_c256_mx<16,false>");
}

}  // namespace ss

// ss_mfcc_c256: the headline kernel -- fused MFCC for fft_points = 512 (C = 256 packed complex
// points) on gfx950, wave64.
//
// Mapping (why it looks like this on CDNA4):
//   * 16 lanes (one DPP row) own one frame, 16 complex points per lane; a wave carries 4 frames,
//     a 256-thread workgroup 16.  The 256-point FFT is two radix-16 register butterflies with ONE
//     transposing exchange between them.  Every exchange is private to a wave, so the main loop
//     has no workgroup barrier at all: LDS operations of one wave execute in order.
//   * exchange buffer index i + (i >> 4) (one pad slot per 16): the stride-16 scatter of the
//     first pass hits 16 distinct bank pairs per ds_write_b64 lane group, and a frame's slice is
//     2176 B = 34 bank rows + 32 banks, so the two frames of a 32-lane ds_read_b64 group sit on
//     complementary halves of the 64 banks.
//   * zero padding is compile-time: a 320-sample frame fills only 10 of the 16 inputs of each
//     first-pass butterfly (template NE), the rest fold away.
//   * the frame energy is reduced across the 16 lanes with DPP row operations (no LDS).
//   * mel: each lane owns up to three filters (host-sorted by length so the lock-step loop
//     count is small); weights come from a [tap][lane] LDS table (conflict-free).
//   * DCT-II: lane c < n_ceps accumulates its coefficient from the log-mel row (LDS broadcast).
//   * HBM traffic: samples once (the 50 % frame overlap is served by L1/L2), 13 floats out.
//
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.
#include "ss_device.h"
#include "ss_fft_reg.h"

namespace ss {

namespace {

constexpr float kEpsF = 1.1920929e-7f;  // f32::EPSILON, functions.rs:70
constexpr int kFrameSlots = 272;        // float2 per frame region (256 + 16 pad)
constexpr int kTableOffset = 16 * kFrameSlots + 2;  // float2 units; +2: lane 0 of the last frame reads one slot past its region

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row; every lane ends with the same bits
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}

template <int NE, bool EXACT>
__global__ __launch_bounds__(256) void ss_mfcc_c256(const Fast512Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;  // frame within the wave
    const int j = lane & 15;  // lane within the frame

    // ---- LDS carve: one 2176-B region per frame first (compile-time 8-byte alignment -> ds_read/write_b64),
    //      then the read-only tables ----
    float2 *zfr = reinterpret_cast<float2 *>(smem) + (wave * 4 + f) * kFrameSlots;
    float *prow = reinterpret_cast<float *>(zfr);  // P[0..256] reuses the frame region after the untangle reads
    float *frow = prow + 260;                      // log-mel row
    float2 *s_twn = reinterpret_cast<float2 *>(smem) + kTableOffset;          // 129 (+3 pad) float2
    float *s_dct = reinterpret_cast<float *>(s_twn + 132);                    // [M][16]
    float *s_melw = s_dct + a.n_filters * 16;                                 // [sum maxlen][16]
    int *s_melst = reinterpret_cast<int *>(s_melw + a.mel_wrows * 16);        // [3][16]
    int *s_melf = s_melst + 48;                                               // [3][16]

    for (int i = tid; i < 129; i += 256) s_twn[i] = a.tw_n[i];
    for (int i = tid; i < static_cast<int>(a.n_filters) * 16; i += 256) s_dct[i] = a.dct16[i];
    for (int i = tid; i < a.mel_wrows * 16; i += 256) s_melw[i] = a.mel_w[i];
    if (tid < 48) {
        s_melst[tid] = a.mel_start[tid];
        s_melf[tid] = a.mel_filter[tid];
    }
    // second-pass twiddles exp(-2 pi i j r / 256), r = 1..15, live in registers for the whole kernel
    float2 tw2[15];
#pragma unroll
    for (int r = 1; r < 16; ++r) tw2[r - 1] = a.tw_c[j * r];
    __syncthreads();

    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    const unsigned long long groups = (total + 15) / 16;
    const int M = static_cast<int>(a.n_filters);
    const int Cc = static_cast<int>(a.n_ceps);
    // partner index base for Z[256-k]: lanes j >= 1 read phys(256-k) = (271 - j) - 17 i, lane 0 reads 272 - 17 i
    const int cbase = j == 0 ? 272 : 271 - j;

    for (unsigned long long g = blockIdx.x; g < groups; g += gridDim.x) {
        const unsigned long long gf = g * 16 + wave * 4 + f;
        const bool active = gf < total;
        const unsigned long long gfc = active ? gf : total - 1;
        const unsigned clip = static_cast<unsigned>(gfc / a.n_frames);
        const unsigned t = static_cast<unsigned>(gfc - static_cast<unsigned long long>(clip) * a.n_frames);
        // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
        const float2 *src = reinterpret_cast<const float2 *>(a.x + static_cast<unsigned long long>(clip) * a.ld + t * a.step);

        float2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (e < NE) {
                const int n = j + 16 * e;
                if (EXACT) v[e] = src[n];
                else v[e] = 2 * n < static_cast<int>(a.flen) ? src[n] : make_float2(0.f, 0.f);
            } else {
                v[e] = make_float2(0.f, 0.f);  // zero padding to fft_points (processing.rs:147-156)
            }
        }
        // ---- 256-point complex FFT: radix-16, transpose through LDS, twiddle, radix-16 ----
        fft16_reg(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) zfr[17 * j + r] = v[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = zfr[j + 17 * r];
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], tw2[r - 1]);
        fft16_reg(v);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) zfr[j + 17 * r] = v[r];  // natural order: Z[k] at k + (k >> 4)
        __builtin_amdgcn_wave_barrier();

        // ---- untangle Z -> X, magnitude (processing.rs:168), * 1/N (:180), row sum (feature.rs:216) ----
        float pk[8], pc[8];
        float esum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = j + 16 * i;
            const float2 zk = zfr[j + 17 * i];
            float2 zc = zfr[cbase - 17 * i];
            if (i == 0 && j == 0) zc = zk;  // Z[256] == Z[0]
            const float2 w = s_twn[k];
            const float2 s = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
            const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
            const float2 wd = cmul(w, d);
            const float xa_r = s.x + wd.y, xa_i = s.y - wd.x;  // X[k]
            const float xb_r = s.x - wd.y, xb_i = s.y + wd.x;  // conj X[256-k]
            const float ma = __builtin_amdgcn_sqrtf(xa_r * xa_r + xa_i * xa_i);
            const float mb = __builtin_amdgcn_sqrtf(xb_r * xb_r + xb_i * xb_i);
            pk[i] = a.spectrum_exponent == 2 ? a.scale * (ma * ma) : a.scale * ma;
            pc[i] = a.spectrum_exponent == 2 ? a.scale * (mb * mb) : a.scale * mb;
            esum += pk[i] + pc[i];
        }
        float p128 = 0.f;
        if (j == 0) {
            const float2 z = zfr[128 + 8];  // X[128] = conj Z[128]
            const float m = __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);
            p128 = a.spectrum_exponent == 2 ? a.scale * (m * m) : a.scale * m;
            esum += p128;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            prow[j + 16 * i] = pk[i];
            prow[256 - (j + 16 * i)] = pc[i];
        }
        if (j == 0) prow[128] = p128;
        __builtin_amdgcn_wave_barrier();
        float energy = row16_sum(esum);
        energy = energy == 0.f ? kEpsF : energy;  // zero_handling, feature.rs:219

        // ---- banded mel reduction (feature.rs:229), zero handling (:230), ln (:105) ----
        int wrow = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int len = a.mel_maxlen[s];
            if (len > 0) {
                const float *pp = prow + s_melst[s * 16 + j];
                const float *ww = s_melw + wrow * 16 + j;
                float acc = 0.f;
                for (int q = 0; q < len; ++q) acc = fmaf(ww[q * 16], pp[q], acc);
                const int m = s_melf[s * 16 + j];
                acc = acc == 0.f ? kEpsF : acc;
                if (m >= 0) frow[m] = __logf(acc);
                wrow += len;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- DCT-II, first n_ceps coefficients, scaling + column-0 replacement (feature.rs:120-146) ----
        if (j < Cc) {
            float acc = 0.f;
            for (int m = 0; m < M; ++m) acc = fmaf(frow[m], s_dct[m * 16 + j], acc);
            float o;
            if (j == 0) o = a.dc_elimination ? __logf(energy) : acc * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
            else o = acc * a.dct_scale_k;
            if (active) a.out[gf * Cc + j] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

size_t fast512_lds_bytes(const Fast512Args &a)
{
    return a.table_bytes + kTableOffset * sizeof(float2);
}

hipError_t launch_mfcc_c256(const Fast512Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = fast512_lds_bytes(a);
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    const unsigned long long groups = (total + 15) / 16;
    if (groups == 0) return hipSuccess;
    // persistent grid: at most 4 workgroups per CU, sized so that every workgroup gets the same number of groups
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256) * 4;
    const unsigned long long per = (groups + cap - 1) / cap;
    const unsigned grid = static_cast<unsigned>((groups + per - 1) / per);
    const bool exact10 = a.flen == 320, full = a.flen == 512;
    auto go = [&](auto kern, const char *name) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds));
            if (e != hipSuccess) return e;
        }
        if (info) *info = LaunchInfo{name, grid, 256u, lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
        return hipGetLastError();
    };
    if (exact10) return go(ss_mfcc_c256<10, true>, "ss_mfcc_c256<10,true>");
    if (full) return go(ss_mfcc_c256<16, true>, "ss_mfcc_c256<16,true>");
    if (a.flen <= 320) return go(ss_mfcc_c256<10, false>, "ss_mfcc_c256<10,false>");
    return go(ss_mfcc_c256<16, false>, "ss_mfcc_c256<16,false>");
}

}  // namespace ss

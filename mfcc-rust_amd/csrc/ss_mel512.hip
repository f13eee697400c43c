// ss_mel_c256: fused mel-spectrogram for fft_points = 512 on gfx950 -- frame_analysis + |X wnorm|^2 + the mel einsum
// (functions.rs:125-170, feature.rs:151-174) in one launch; the STFT-path sibling of ss_mfcc512.hip, one size down from
// ss_mel2048.hip.
//
//   * 16 lanes (one DPP row) own a 512-sample window = 256 packed complex points, 16 per lane; a wave carries four
//     consecutive rows of one clip, the work unit pulled from the LDS counter is a (clip, row quad).  One persistent
//     12-wave workgroup per CU; the next unit's samples are prefetched into the dead window registers.
//   * Row r of a clip covers the 512 samples that end at chunk r + n_pad (functions.rs:137-151): zero initial state,
//     zero tail, rows past the real ones are all zero (D3).  Windows inside the clip load as 8-byte pairs at constant
//     offsets; at the clip edges one masked range per lane.  Vorbis window pairs come from the table block in LDS.
//   * FFT / untangle exactly as in ss_mfcc512.hip (radix-16, one transposing exchange through the row's wave-private
//     2304-B slot, twiddle, radix-16; partner by ds_bpermute_b32).  |X|^2 wnorm^2 of bins 0..128 -- or all 257 when the
//     bank reaches past (F+1)/2 -- goes to a P row inside the slot.
//   * banded mel, up to 5 filters per lane (80 filters), host-sorted by tap count; the four rows of a wave are adjacent
//     words of out[clip][m][.].  No zero handling, no log (feature.rs:164-173).
//   * stft build (ss_stft_device): skips the mel stage and writes X[k] wnorm for all 257 bins of a row.
// Tables: ss::mel512_layout (ss_internal.h).
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

namespace ss {

namespace {

using namespace wv;

namespace L = mel512_layout;
constexpr int kSlotFloatsE = 576;  // per row: exchange slot (288 float2); afterwards the P row [260]
constexpr int kWaveFloatsE = 4 * kSlotFloatsE;


template <int WAVES, bool STFT>
__global__ __launch_bounds__(WAVES * 64) void ss_mel_c256(const Mel512Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;  // row within the quad
    const int j = lane & 15;  // lane within the row (DPP row)

    float *slot = reinterpret_cast<float *>(smem) + wave * kWaveFloatsE + f * kSlotFloatsE;
    float2 *zh = reinterpret_cast<float2 *>(slot);
    float *prow = slot;
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWaveFloatsE;
    const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kWin);
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_melw = s_tab + L::kMelW;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 16 * a.mel_wpitch);

    const int R = static_cast<int>(a.rows), Rreal = static_cast<int>(a.real_rows), M = static_cast<int>(a.n_filters);
    // work unit: four consecutive rows of one clip; the workgroup owns a contiguous range of units
    const unsigned qpc = (a.rows + 3) / 4;
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * qpc;
    const unsigned u_lo = static_cast<unsigned>(units * blockIdx.x / gridDim.x);
    const unsigned u_hi = static_cast<unsigned>(units * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 16 * a.mel_wpitch) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = u_lo + WAVES;
    }
    // (clip, row) of this lane group within a unit, and the loads of its window: functions.rs:137-151, the window covers
    // the last 512 samples ending at chunk r + n_pad
    auto load_unit = [&](unsigned un, float2 (&vv)[16]) {
        const unsigned clip = un / qpc;
        const int r = static_cast<int>(un - clip * qpc) * 4 + f;
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        const bool active = r < Rreal;
        const int start = static_cast<int>(r + a.n_pad + 1) * static_cast<int>(a.hop) - 512;
        const bool inside = active && start >= 0 && start + 512 <= static_cast<int>(a.n_samples);
        const float2 *src = reinterpret_cast<const float2 *>(xc + start) + j;
        if (__all(inside)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) vv[e] = src[16 * e];
        } else {
            // clip edges (zero initial state, zero padding of the last chunk, D3) and inactive rows.  start and n_samples are
            // even, so a sample pair is inside or outside as a whole; the valid pairs form one range [e_lo, e_hi) per lane
            // (the address of a masked load may lie outside the clip; it is never dereferenced)
            const int base = start + 2 * j;
            const int n = static_cast<int>(a.n_samples);
            if (((start | n) & 1) == 0) {
                const int e_lo = base >= 0 ? 0 : (31 - base) >> 5;
                int e_hi = base >= n ? 0 : min(16, (n - base + 31) >> 5);
                if (!active) e_hi = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float2 s = make_float2(0.f, 0.f);
                    if (e >= e_lo && e < e_hi) s = src[16 * e];
                    vv[e] = s;
                }
            } else {  // odd hop or clip length: a pair may straddle the clip edge, bounds per sample
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int p0 = base + 32 * e;
                    vv[e] = make_float2(active && p0 >= 0 && p0 < n ? xc[p0] : 0.f, active && p0 + 1 >= 0 && p0 + 1 < n ? xc[p0 + 1] : 0.f);
                }
            }
        }
    };
    unsigned unit = __builtin_amdgcn_readfirstlane(u_lo + wave);  // uniform: kept scalar
    float2 vin[16];
    if (unit < u_hi) load_unit(unit, vin);

    const int paddr = ((lane & 48) | ((16 - j) & 15)) << 2;  // lane holding Z[256 - k]
    const int wbase1 = 34 * (j >> 1) + (j & 1);              // exchange write base (float2 units)
    const float hs = 0.25f * a.scale * a.scale;              // |X wnorm|^2 = (wnorm^2 / 4) |2X|^2
    __syncthreads();
    int st[5], fi[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        st[s] = s_start[s * 16 + j];
        fi[s] = s_filt[s * 16 + j];
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + j * a.mel_wpitch);
    float2 twn[8];  // the untangle twiddles stay in registers; the pass-2 ones would spill next to the 32 prefetch registers
#pragma unroll
    for (int r = 0; r < 8; ++r) twn[r] = s_twn[r * 16 + j];

    while (unit < u_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);
        const unsigned clip = unit / qpc;
        const int r = static_cast<int>(unit - clip * qpc) * 4 + f;

        float2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float2 w = s_win[j + 16 * e];
            v[e] = make_float2(vin[e].x * w.x, vin[e].y * w.y);
        }
        // ---- 256-point complex FFT: radix-16, transpose through LDS, twiddle, radix-16 ----
        fft16_reg(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) zh[wbase1 + 2 * q] = v[q];
        wave_order();
        if (next < u_hi) load_unit(next, vin);
        float2 u[16];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 t4 = *reinterpret_cast<const float4 *>(&zh[34 * p + 2 * j]);
            u[2 * p] = make_float2(t4.x, t4.y);
            u[2 * p + 1] = make_float2(t4.z, t4.w);
        }
        wave_order();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 w2 = s_tw2[p * 16 + j];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 7) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
        }
        fft16_reg(u);  // u[q] = Z[j + 16 q]

        // ---- untangle Z -> X; (|X| wnorm)^2 (functions.rs:166-169 + feature.rs:164) ----
        // stft build (functions.rs:86-123, :166-169): X[k] * wnorm for all 257 bins of the row, interleaved re / im
        float2 *srow = nullptr;
        if (STFT && r < R) srow = reinterpret_cast<float2 *>(a.out) + (static_cast<unsigned long long>(clip) * R + r) * 257ull;
        const float cs = 0.5f * a.scale;
        float2 zcs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) zcs[q] = make_float2(bperm(paddr, u[15 - q].x), bperm(paddr, u[15 - q].y));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 zk = u[q];
            // lane 0 pairs with itself: Z[256 - 16 q] = own register (16 - q) & 15
            const float2 zc = j == 0 ? u[(16 - q) & 15] : zcs[q];
            const float2 w = twn[q];
            const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
            const float2 d = make_float2(zk.x - zc.x, zk.y + zc.y);
            // 2 X[k] = s - i w d, 2 conj X[256-k] = 2 s - 2 X[k]
            const float xr = fmaf(w.y, d.x, fmaf(w.x, d.y, s.x));
            const float xi = fmaf(w.y, d.y, fmaf(-w.x, d.x, s.y));
            if (STFT) {
                if (srow) {
                    srow[j + 16 * q] = make_float2(cs * xr, cs * xi);
                    srow[256 - j - 16 * q] = make_float2(cs * fmaf(2.f, s.x, -xr), -cs * fmaf(2.f, s.y, -xi));
                }
                continue;
            }
            prow[j + 16 * q] = hs * fmaf(xr, xr, xi * xi);
            if (a.fullp) {  // the bank reaches past (F+1)/2: bins 129..256 as well
                const float yr = fmaf(2.f, s.x, -xr), yi = fmaf(2.f, s.y, -xi);
                prow[256 - j - 16 * q] = hs * fmaf(yr, yr, yi * yi);
            }
        }
        if (j == 0) {
            const float2 z = u[8];  // X[128] = conj Z[128]
            if (STFT) {
                if (srow) srow[128] = make_float2(a.scale * z.x, -a.scale * z.y);
            } else {
                prow[128] = hs * 4.f * fmaf(z.x, z.x, z.y * z.y);
            }
        }
        if (STFT) {
            wave_order();
            unit = next;
            continue;
        }
        if (j < 3) prow[(a.fullp ? 257 : 129) + j] = 0.f;  // pad bins read (with zero weight) by the mel stage
        wave_order();

        // ---- banded mel reduction (feature.rs:173); the four rows of the wave are adjacent words of out[clip][m][.] ----
        if (r < R) {
            float *dst = a.out + static_cast<unsigned long long>(clip) * M * R + r;
            int off = 0;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                if (a.mel_q4[s] == 0) break;
                const float m = mel_slot1(w4 + off, prow + st[s], a.mel_q4[s]);
                if (fi[s] >= 0) dst[static_cast<unsigned long long>(fi[s]) * R] = m;
                off += a.mel_q4[s];
            }
        }
        wave_order();
        unit = next;
    }
}

}  // namespace

hipError_t launch_mel_c256(const Mel512Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    constexpr int WAVES = 12;
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloatsE + L::kMelW + 4 + 16 * static_cast<size_t>(a.mel_wpitch)) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.batch == 0 || a.rows == 0) return hipSuccess;
    const unsigned cap = static_cast<unsigned>(num_cus > 0 ? num_cus : 256);
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * ((a.rows + 3) / 4);
    if (units >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned long long blocks = (units + WAVES - 1) / WAVES;
    const unsigned grid = static_cast<unsigned>(blocks < cap ? blocks : cap);
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(WAVES * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
        return hipGetLastError();
    };
    return a.out_stft ? go(ss_mel_c256<WAVES, true>, "ss_mel_c256<stft>") : go(ss_mel_c256<WAVES, false>, "ss_mel_c256");
}

}  // namespace ss

// RETIRED EXPERIMENT (not built): needs the mel2048w_layout / build_mel2048w host tables of commit history; see DESIGN.md 4.1 findings.
// ss_mel_c1024w: fused mel spectrogram for fft_points = 2048 (C = 1024 packed complex points), "wide" mapping: ONE row
// per 64 lanes with 16 complex points per lane (ss_mel2048.hip holds 32 per lane, two rows per wave).  Half the registers
// per lane (<= 128 VGPRs) buys 3 waves per SIMD instead of 2, which is what the 2048-point kernel lacks: it is bound by
// VALU issue but only 77 % busy, for want of waves to cover its LDS round trips.
//
//   * 1024-point FFT = 16 x 64: a radix-16 register butterfly over n2 (n = n1 + 64 n2, n1 = lane), ONE transposing exchange
//     through wave-private LDS (four independent 16 x 16 transposes, n1 mod 4), then the 64-point transform over
//     n1 = a + 4b: lane (k1, a) does a radix-16 butterfly over b; the radix-4 over a runs ACROSS the four 16-lane rows:
//     two swap stages (v_permlane16_swap for bit 0 of a, v_permlane32_swap for bit 1) transpose the 4 x 4 blocks
//     {row a} x {quarter of c}, after which lane (k1, a') holds G_s[c] for all four s and the quarter c = 4a' + ci, applies
//     W1024^(s (k1 + 16 c)) and finishes with an in-lane 4-point DFT:
//       Z[k1 + 16 c + 256 d] = sum_s W4^(s d) W1024^(s (k1 + 16 c)) G_s[c],  G_s[c] = FFT16_b(A[s+4b][k1] W256^(b k1))[c].
//   * real-FFT untangle of the bins the bank can touch (k <= 512): lane (k1, a') register (d, ci), d < 2, holds
//     k = k1 + 64 a' + 16 ci + 256 d; its partner 1024 - k is register (3 - d, 3 - ci) of lane (16 - k1, 3 - a'), fetched
//     with ds_bpermute_b32.  The four k1 = 0 lanes (bins 0, 16, ..., 1008) pair among themselves in an irregular pattern and
//     go through 512 B of LDS instead.
//   * (|X| wnorm)^2 -> P row in LDS -> banded mel reduction, 2 filters per lane; a wave does the two rows of a pair one
//     after the other and stores them as adjacent words of out[clip][m][.].
// Reference semantics: functions.rs:86-170 (frame_analysis / stft2), feature.rs:151-174.  Tables: ss::mel2048w_layout.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"

#include <cstdlib>

namespace ss {

namespace {

namespace L = mel2048w_layout;
constexpr int kClsW = 8 * 34 + 8;            // float2 per class slice of the exchange (+8: neighbouring classes 16 banks apart)
constexpr int kWaveFloatsW = 4 * kClsW * 2;  // exchange region: four classes (8960 B); P row + k1 = 0 scratch reuse it
constexpr int kScratchOff = 528;             // float offset of the k1 = 0 scratch (64 float2) behind the P row [520]

__device__ __forceinline__ void wave_order_w()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float bperm_w(int addr, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// inline assembly with their own hazard s_nop (see ss_mfcc4096.hip / ss_mfcc1024.hip)
__device__ __forceinline__ void swap_rows_w(float &a, float &b)  // odd DPP rows of a <-> even rows of b
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_halves_w(float &a, float &b)  // lanes 32..63 of a <-> lanes 0..31 of b
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ float mel_slot_w(const float4 *w4, const float4 *p4, int q4)
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
    for (; i < q4; ++i) {
        const float4 w = w4[i], t = p4[i];
        acc = fmaf(w.x, t.x, acc);
        acc = fmaf(w.y, t.y, acc);
        acc = fmaf(w.z, t.z, acc);
        acc = fmaf(w.w, t.w, acc);
    }
    return acc;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ss_mel_c1024w(const Mel2048Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int k1 = lane & 15, ap = lane >> 4;  // reader view: column k1, row a (then quarter a')
    const int cls = lane & 3, bw = lane >> 2;  // writer view: n1 = lane = cls + 4 bw

    float *wbase = reinterpret_cast<float *>(smem) + wave * kWaveFloatsW;
    float2 *ex = reinterpret_cast<float2 *>(wbase);
    float *prow = wbase;                                            // P[0..512] + zero pad bins, after the exchange
    float2 *zscr = reinterpret_cast<float2 *>(wbase + kScratchOff);  // Z[16 cc] of the k1 = 0 lanes, cc < 64
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWaveFloatsW;
    const float4 *s_t1 = reinterpret_cast<const float4 *>(s_tab + L::kT1);
    const float2 *s_t2 = reinterpret_cast<const float2 *>(s_tab + L::kT2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kWin);
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_melw = s_tab + L::kMelW;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 64 * a.mel_wpitch);

    const unsigned pairs = (a.rows + 1) / 2;
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * pairs;
    const unsigned u_lo = static_cast<unsigned>(units * blockIdx.x / gridDim.x);
    const unsigned u_hi = static_cast<unsigned>(units * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 64 * a.mel_wpitch) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = u_lo + WAVES;
    }
    __syncthreads();
    const int st0 = s_start[lane], st1 = s_start[64 + lane];
    const int fi0 = s_filt[lane], fi1 = s_filt[64 + lane];
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + lane * a.mel_wpitch);
    const int paddr = (((16 - k1) & 15) | ((3 - ap) << 4)) << 2;  // lane holding Z[1024 - k] (k1 != 0)
    float2 *exw = ex + cls * kClsW + 34 * (bw >> 1) + (bw & 1);  // writer base (float2 units)
    const float2 *exr = ex + ap * kClsW + 2 * k1;                 // reader base
    const float hs = 0.25f * a.scale * a.scale;                   // |X wnorm|^2 = (wnorm^2 / 4) |2X|^2
    const bool k1z = k1 == 0;
    const int R = static_cast<int>(a.rows), Rreal = static_cast<int>(a.real_rows);
    const int M = static_cast<int>(a.n_filters);

    unsigned unit = u_lo + wave;
    while (unit < u_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);
        const unsigned clip = unit / pairs;
        const int r0row = static_cast<int>(unit - clip * pairs) * 2;
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        float mel_a0 = 0.f, mel_a1 = 0.f;  // first row's two mel values of this lane
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int r = r0row + rr;
            const bool active = r < Rreal;
            // functions.rs:137-151: window over the last W samples ending at chunk r + n_pad
            const int start = static_cast<int>(r + a.n_pad + 1) * static_cast<int>(a.hop) - 2048;
            const bool inside = active && start >= 0 && start + 2048 <= static_cast<int>(a.n_samples);
            const float2 *src = reinterpret_cast<const float2 *>(xc + start) + lane;
            float2 v[16];
            if (inside) {  // uniform: the row is the same for the whole wave
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = src[64 * e];
            } else {
                // clip edges (zero initial state, zero padding of the last chunk, D3) and inactive rows: one masked range per lane
                const int base = start + 2 * lane;
                const int n = static_cast<int>(a.n_samples);
                int e_lo = base >= 0 ? 0 : (127 - base) >> 7;
                int e_hi = base >= n ? 0 : min(16, (n - base + 127) >> 7);
                if (!active) e_hi = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float2 s = make_float2(0.f, 0.f);
                    if (e >= e_lo && e < e_hi) s = src[64 * e];
                    v[e] = s;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float2 w = s_win[lane + 64 * e];
                v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
            }
            // ---- pass 1: radix-16 over n2; transpose (four 16 x 16 problems: n1 mod 4) ----
            fft_reg<16>(v);
#pragma unroll
            for (int k = 0; k < 16; ++k) exw[2 * k] = v[k];
            wave_order_w();
            float2 u[16];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
            wave_order_w();
            // ---- twiddle W256^(b k1), radix-16 over b: u[c] = G_a[c] ----
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 w2 = s_t1[p * 16 + k1];
                u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
                if (p < 7) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
            }
            fft_reg<16>(u);
            // ---- 4 x 4 transposition across the rows: slot s = c >> 2 becomes source row s, the lane keeps quarter a' ----
#pragma unroll
            for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    swap_rows_w(u[8 * b1 + ci].x, u[8 * b1 + 4 + ci].x);
                    swap_rows_w(u[8 * b1 + ci].y, u[8 * b1 + 4 + ci].y);
                }
#pragma unroll
            for (int b0 = 0; b0 < 2; ++b0)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    swap_halves_w(u[4 * b0 + ci].x, u[8 + 4 * b0 + ci].x);
                    swap_halves_w(u[4 * b0 + ci].y, u[8 + 4 * b0 + ci].y);
                }
            // ---- twiddle W1024^(s (k1 + 16 c)), in-lane radix-4 over s: z[d][ci] = Z[k1 + 16 (4a' + ci) + 256 d] ----
            float2 z[4][4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                float2 t0 = u[ci];
                float2 t1 = cmul(u[4 + ci], s_t2[(0 * 4 + ci) * 64 + lane]);
                float2 t2 = cmul(u[8 + ci], s_t2[(1 * 4 + ci) * 64 + lane]);
                float2 t3 = cmul(u[12 + ci], s_t2[(2 * 4 + ci) * 64 + lane]);
                fft4(t0, t1, t2, t3);
                z[0][ci] = t0;
                z[1][ci] = t1;
                z[2][ci] = t2;
                z[3][ci] = t3;
            }
            // ---- untangle the bins the bank can touch: registers (d, ci), d < 2 ----
            if (k1z) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) zscr[4 * ap + ci + 16 * d] = z[d][ci];
            }
            wave_order_w();
            float2 zcs[8];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const float2 sv = z[3 - d][3 - ci];
                    float2 zc = make_float2(bperm_w(paddr, sv.x), bperm_w(paddr, sv.y));
                    if (k1z) zc = zscr[(64 - (4 * ap + ci + 16 * d)) & 63];
                    zcs[4 * d + ci] = zc;
                }
            wave_order_w();  // the scratch sits inside the region the P row is about to fill? no: behind it; the barrier orders the reads
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const float2 zk = z[d][ci], zc = zcs[4 * d + ci];
                    const float2 w = s_twn[(4 * d + ci) * 64 + lane];
                    const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                    const float2 dd = make_float2(zk.x - zc.x, zk.y + zc.y);
                    // 2 X[k] = s - i w dd
                    const float xr = fmaf(w.y, dd.x, fmaf(w.x, dd.y, s.x));
                    const float xi = fmaf(w.y, dd.y, fmaf(-w.x, dd.x, s.y));
                    prow[k1 + 64 * ap + 16 * ci + 256 * d] = hs * (xr * xr + xi * xi);  // (|X| wnorm)^2
                }
            if (lane == 0) {
                const float2 zz = z[2][0];  // X[512] = conj Z[512]
                prow[512] = hs * 4.f * (zz.x * zz.x + zz.y * zz.y);
            }
            if (lane < 3) prow[513 + lane] = 0.f;  // pad bins read (with zero weight) by the mel stage
            wave_order_w();
            // ---- banded mel reduction (feature.rs:173), two filters per lane ----
            const float m0 = mel_slot_w(w4, reinterpret_cast<const float4 *>(prow + st0), a.mel_q4[0]);
            const float m1 = mel_slot_w(w4 + a.mel_q4[0], reinterpret_cast<const float4 *>(prow + st1), a.mel_q4[1]);
            wave_order_w();
            if (rr == 0) {
                mel_a0 = m0;
                mel_a1 = m1;
            } else {
                // the pair's two rows are adjacent words of out[clip][m][.]
                float *dst = a.out + static_cast<unsigned long long>(clip) * M * R + r0row;
                if (fi0 >= 0) {
                    float *q = dst + static_cast<unsigned long long>(fi0) * R;
                    q[0] = mel_a0;
                    if (r < R) q[1] = m0;
                }
                if (fi1 >= 0) {
                    float *q = dst + static_cast<unsigned long long>(fi1) * R;
                    q[0] = mel_a1;
                    if (r < R) q[1] = m1;
                }
            }
        }
        unit = next;
    }
}

template <int WAVES>
hipError_t launch_mel_ww(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloatsW + L::kMelW + 4 + 64 * static_cast<size_t>(a.mel_wpitch)) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.batch == 0) return hipSuccess;
    const unsigned cap = static_cast<unsigned>(num_cus > 0 ? num_cus : 256);
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * ((a.rows + 1) / 2);
    if (units >= 0xffffffffull) return hipErrorInvalidValue;
    unsigned long long blocks = (units + WAVES - 1) / WAVES;
    const unsigned grid = static_cast<unsigned>(blocks < cap ? blocks : cap);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ss_mel_c1024w<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    if (info) *info = LaunchInfo{"ss_mel_c1024w", grid, static_cast<unsigned>(WAVES * 64), lds};
    hipLaunchKernelGGL(ss_mel_c1024w<WAVES>, dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_mel_c1024w(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    return launch_mel_ww<12>(a, stream, num_cus, info);
}

}  // namespace ss

// ss_mfcc_c512: fused MFCC / mfe for fft_points = 1024 (C = 512 packed complex points) on gfx950 -- the structure of the
// 4096-point kernel (ss_mfcc4096.hip) at a quarter of the size, two frames per wave.
//
//   * 32 lanes own a frame, 16 complex points per lane.  512-point FFT = 16 x 32: a radix-16 register butterfly over n2
//     (n = n1 + 32 n2, n1 = lane), ONE transposing exchange through wave-private LDS, then the 32-point transform over n1
//     split as n1 = a + 2b: lane (k1, a) does a radix-16 butterfly over b, and the last radix-2 over a pairs lanes L and
//     L + 16 with v_permlane16_swap (odd DPP rows of one register trade places with even rows of another: one swap of
//     registers (i, 8 + i) gives the even-row lane both halves' column i and the odd-row lane both halves' column 8 + i):
//       Z[k1 + 16 c + 256 d] = G0[c] + (-1)^d W512^(k1 + 16 c) G1[c],  Ga[c] = FFT16_b(A[a+2b][k1] W256^(b k1))[c].
//     The exchange is two independent 16 x 16 transposes (even / odd n1), ds_write_b64 scatter to
//     34*(b>>1) + 2*k1 + (b&1) per class, ds_read_b128 back; conflict-free on both sides.
//   * real-FFT untangle: lane (k1, h) register r0[i] holds bin k = k1 + 16 i + 128 h (< 256); its partner 512 - k is
//     r1[7 - i] of lane (16 - k1, 1 - h) (r1[8 - i] for the k1 = 0 lanes), fetched with ds_bpermute_b32.
//   * |X|/N: bins 0..256 go to the P row (the mel bank ends at (F+1)/2, feature.rs:69-70), all 513 feed the frame energy.
//   * banded mel (4 filters per lane, aligned float4 taps), zero handling, ln, symmetric DCT-II (sum / difference rows),
//     reference scaling, column-0 replacement; optional frame window; mfe build.
//   * 12 waves per CU, frame pairs from an LDS counter; no workgroup barrier in the main loop.
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.  Tables: ss::mfcc1024_layout.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

namespace ss {

namespace {

using namespace wv;

namespace L = mfcc1024_layout;
constexpr int kClsK = 8 * 34 + 8;        // float2 per class slice (+8: the two classes of a write group sit 16 banks apart)
constexpr int kFrameF2 = 560;            // float2 per frame region (4480 B = 128 mod 256: the two frames of a wave use different banks)
constexpr int kFrameFloats = 2 * kFrameF2;  // after the exchange: P row [260] | ln(mel) row [128] | s [64] | d [64]
constexpr int kWaveFloatsK = 2 * kFrameFloats;

// v_permlane16_swap: the odd DPP rows (lanes 16..31, 48..63) of `a` trade places with the even rows of `b`
// (tools/ubench/permlane16.hip).  Inline assembly with its own hazard s_nop, as for v_permlane32_swap in ss_mfcc4096.hip.
__device__ __forceinline__ void swap_rows(float &a, float &b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <bool POW2, bool MFE, bool WIN, int WAVES, bool LIB = false>
__global__ __launch_bounds__(WAVES * 64) void ss_mfcc_c512(const Mfcc1024Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int fr = lane >> 5;                 // frame within the wave
    const int jj = lane & 31;                 // lane within the frame
    const int k1 = jj & 15, h = jj >> 4;      // reader view: column k1, half a = h
    const int cls = jj & 1, bw = jj >> 1;     // writer view: n1 = jj = cls + 2 bw

    float *fbase = reinterpret_cast<float *>(smem) + wave * kWaveFloatsK + fr * kFrameFloats;
    float2 *exf = reinterpret_cast<float2 *>(fbase);
    // LIB builds (librosa-compatible switches) keep all 513 bins of the P row: [516] | ln(mel) [128] | s [64] | d [64]
    float *prow = fbase, *frow = fbase + (LIB ? 516 : 260), *srow = frow + 128, *drow = srow + 64;
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWaveFloatsK;
    const float4 *s_t1 = reinterpret_cast<const float4 *>(s_tab + L::kT1);
    const float2 *s_t2 = reinterpret_cast<const float2 *>(s_tab + L::kT2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kWin);
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_cos = s_tab + L::kCos;
    const float *s_melw = s_tab + L::kMelW;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 32 * a.mel_wpitch);

    const unsigned total = a.batch * a.n_frames;
    const unsigned units = (total + 1) / 2;
    const unsigned u_lo = static_cast<unsigned>(static_cast<unsigned long long>(units) * blockIdx.x / gridDim.x);
    const unsigned u_hi = static_cast<unsigned>(static_cast<unsigned long long>(units) * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 32 * a.mel_wpitch) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = u_lo + WAVES;
    }
    __syncthreads();
    int st[4], fi[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st[s] = s_start[s * 32 + jj];
        fi[s] = s_filt[s * 32 + jj];
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + jj * a.mel_wpitch);
    const int paddr = ((lane & 32) | ((16 - k1) & 15) | ((1 - h) << 4)) << 2;  // lane holding Z[512 - k]
    float2 *exw = exf + cls * kClsK + 34 * (bw >> 1) + (bw & 1);  // writer base (float2 units)
    const float2 *exr = exf + h * kClsK + 2 * k1;                  // reader base
    const float hscale32 = (POW2 ? 0.25f * a.scale : 0.5f * a.scale) * kTwo32;
    const bool k1z = k1 == 0;
    const int M = static_cast<int>(a.n_filters), Cc = static_cast<int>(a.n_ceps), Mh = M / 2, Mc = (M + 1) / 2;
    // valid sample pairs of this lane: n = jj + 32 e with 2 n < flen (zero pad to fft_points, processing.rs:147-156)
    const int e_hi = min(16, max(0, (static_cast<int>(a.flen) / 2 - jj + 31) >> 5));
    const int half_pairs = static_cast<int>(a.flen) / 2;
    const bool odd_tail = (a.flen & 1) != 0;
    const bool pre = a.preemph != 0.f;  // fused pre-emphasis (run-time: a uniform branch in the loader)
    const unsigned psh = a.preemph_shift % a.n_samples;

    unsigned unit = __builtin_amdgcn_readfirstlane(u_lo + wave);  // uniform: kept scalar
    while (unit < u_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);

        const unsigned gf_raw = 2 * unit + fr;
        const bool live = gf_raw < total;
        const unsigned gf = live ? gf_raw : total - 1;
        const unsigned clip = gf / a.n_frames;
        const unsigned t = gf - clip * a.n_frames;
        // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        float2 v[16];
        // librosa center=True (LIB builds): frame t is centred on sample t*step; frames inside the clip load like contract
        // frames from their (even) start, the few at the clip edges mirror (np.pad 'reflect') or zero their missing samples
        const int s0 = static_cast<int>(t * a.step) - (LIB && a.center ? static_cast<int>(a.flen / 2) : 0);
        const int ns = static_cast<int>(a.n_samples);
        if (!LIB || __all(s0 >= 0 && s0 + static_cast<int>(a.flen) <= ns)) {
            const float2 *src = reinterpret_cast<const float2 *>(xc + s0) + jj;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float2 s = make_float2(0.f, 0.f);
                if (e < e_hi) s = src[32 * e];
                // odd frame length: the last sample is the first half of a pair (its partner is zero padding)
                if (odd_tail && jj + 32 * e == half_pairs) s = make_float2(xc[s0 + 2 * half_pairs], 0.f);
                if (pre) {  // fused pre-emphasis (processing.rs:31-53) of the samples that exist
                    const int pos = s0 + 2 * (jj + 32 * e), rem = static_cast<int>(a.flen) - 2 * (jj + 32 * e);
                    if (rem >= 1) s.x = fmaf(-a.preemph, preemph_tap(xc, pos, psh, a.n_samples), s.x);
                    if (rem >= 2) s.y = fmaf(-a.preemph, preemph_tap(xc, pos + 1, psh, a.n_samples), s.y);
                }
                v[e] = s;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float sv[2] = {0.f, 0.f};
                {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        int pos = s0 + 2 * (jj + 32 * e) + hh;
                        bool ok = 2 * (jj + 32 * e) + hh < static_cast<int>(a.flen);  // zero pad; an odd frame length ends in a half pair
                        if (pos < 0 || pos >= ns) {
                            if (a.pad_reflect) pos = pos < 0 ? -pos : 2 * (ns - 1) - pos;
                            else ok = false;
                        }
                        if (ok) sv[hh] = pre ? fmaf(-a.preemph, preemph_tap(xc, pos, psh, a.n_samples), xc[pos]) : xc[pos];
                    }
                }
                v[e] = make_float2(sv[0], sv[1]);
            }
        }
        if (WIN) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float2 w = s_win[jj + 32 * e];
                v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
            }
        }
        // ---- pass 1: radix-16 over n2; transpose (two 16 x 16 problems: even and odd n1) ----
        fft_reg<16>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) exw[2 * k] = v[k];
        wave_order();
        float2 u[16];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
            u[2 * p] = make_float2(t4.x, t4.y);
            u[2 * p + 1] = make_float2(t4.z, t4.w);
        }
        wave_order();
        // ---- twiddle W256^(b k1), radix-16 over b ----
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 w2 = s_t1[p * 16 + k1];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 7) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
        }
        fft_reg<16>(u);  // u[c] = G_a[c], a = h
        // ---- radix-2 over a with one row swap per register pair: r0[i] (d = 0) and r1[i] (d = 1) hold
        //      Z[k1 + 16 c + 256 d] = G0[c] + (-1)^d W512^(k1 + 16 c) G1[c],  c = i + 8 h ----
        float2 r0[8], r1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float px = u[i].x, qx = u[8 + i].x, py = u[i].y, qy = u[8 + i].y;
            swap_rows(px, qx);
            swap_rows(py, qy);
            const float2 wq = cmul(make_float2(qx, qy), s_t2[i * 32 + jj]);
            r0[i] = make_float2(px + wq.x, py + wq.y);
            r1[i] = make_float2(px - wq.x, py - wq.y);
        }

        // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
        // k1 = 0 lanes pair with r1[8 - i] of the other k1 = 0 lane of the frame; their i = 0 pairs are in-lane: (k1, h) = (0, 0)
        // has k = 0 (X[0] and X[512] come from Z[0] alone), (0, 1) has (128, 384) = (r0[0], r1[0]).  Z[256] = r1[0] of (0, 0).
        float esum = 0.f;
        float2 zcs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2 sv = k1z ? r1[(8 - i) & 7] : r1[7 - i];
            zcs[i] = make_float2(bperm(paddr, sv.x), bperm(paddr, sv.y));
        }
        float *pdst = prow + k1 + 128 * h;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2 zk = r0[i];
            float2 zc = zcs[i];
            if (i == 0) zc = k1z ? (h ? r1[0] : zk) : zc;
            const float2 w = s_twn[i * 32 + jj];
            const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
            const float2 dd = make_float2(zk.x - zc.x, zk.y + zc.y);
            // 2 X[k] = s - i w dd, 2 conj X[512-k] = 2 s - 2 X[k]
            const float xa_r = fmaf(w.y, dd.x, fmaf(w.x, dd.y, s.x));
            const float xa_i = fmaf(w.y, dd.y, fmaf(-w.x, dd.x, s.y));
            const float xb_r = fmaf(2.f, s.x, -xa_r), xb_i = fmaf(2.f, s.y, -xa_i);
            const float na = xa_r * xa_r + xa_i * xa_i, nb = xb_r * xb_r + xb_i * xb_i;
            const float pa = POW2 ? na : __builtin_amdgcn_sqrtf(na);
            const float pb = POW2 ? nb : __builtin_amdgcn_sqrtf(nb);
            pdst[16 * i] = pa;  // bins 0..255 (the bank ends at (F+1)/2, feature.rs:69-70); 256 below
            if (LIB) prow[512 - (k1 + 128 * h + 16 * i)] = pb;  // bins 257..512 for banks over the whole spectrum
            esum += pa + pb;
        }
        if (jj == 0) {
            const float2 z = r1[0];  // X[256] = conj Z[256]
            const float n = 4.f * (z.x * z.x + z.y * z.y);
            const float p256 = POW2 ? n : __builtin_amdgcn_sqrtf(n);
            prow[256] = p256;
            esum += p256;
        }
        if (jj < 3) prow[(LIB ? 513 : 257) + jj] = 0.f;  // pad bins read (with zero weight) by the mel stage
        float energy = hscale32 * half_sum(esum);            // E * 2^32
        energy = energy == 0.f ? kEps * kTwo32 : energy;     // zero_handling, feature.rs:219
        wave_order();

        // ---- banded mel reduction (feature.rs:229), zero handling (:230), ln (:105) -> row in filter order ----
        {
            int off = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float m = hscale32 * mel_slot4(w4 + off, reinterpret_cast<const float4 *>(prow + st[s]), a.mel_q4[s]);
                m = m == 0.f ? kEps * kTwo32 : m;
                if (fi[s] >= 0) {
                    if (MFE) {
                        if (live) a.out[static_cast<unsigned long long>(gf) * M + fi[s]] = m * (1.0f / kTwo32);  // exact: power of two
                    } else {
                        frow[fi[s]] = ln_scaled(m);
                    }
                }
                off += a.mel_q4[s];
            }
        }
        if (MFE) {
            if (jj == 0 && live) a.out_energy[gf] = energy * (1.0f / kTwo32);
            wave_order();
            unit = next;
            continue;
        }
        wave_order();
        // ---- DCT-II (feature.rs:120-123) with the m <-> M-1-m symmetry of the cosine ----
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int m = jj + 32 * h2;
            if (m < Mh) {
                const float lo = frow[m], hi = frow[M - 1 - m];
                srow[m] = lo + hi;
                drow[m] = lo - hi;
            } else if (m < Mc) {  // odd filter count: the middle filter pairs with itself (its odd-coefficient cosines are zero)
                srow[m] = frow[m];
                drow[m] = 0.f;
            } else if (m < ((Mc + 3) & ~3)) {  // the product below runs over whole float4s
                srow[m] = 0.f;
                drow[m] = 0.f;
            }
        }
        wave_order();
        if (jj < Cc) {
            const int c = jj;
            const float4 *r4 = reinterpret_cast<const float4 *>((jj & 1) ? drow : srow);
            const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + c * L::kCosPitch);
            float acc = 0.f;
            const int nq = (Mc + 3) / 4;
            for (int i = 0; i < nq; ++i) {
                const float4 r = r4[i], cw = c4[i];
                acc = fmaf(r.x, cw.x, acc);
                acc = fmaf(r.y, cw.y, acc);
                acc = fmaf(r.z, cw.z, acc);
                acc = fmaf(r.w, cw.w, acc);
            }
            // scaling + column-0 replacement (feature.rs:126-146)
            float o = acc * a.dct_scale_k;
            if (c == 0) o = a.dc_elimination ? ln_scaled(energy) : acc * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
            if (live) a.out[static_cast<unsigned long long>(gf) * Cc + c] = o;
        }
        if (Cc > 32) {  // 33..64 cepstra: the lane also forms coefficient c + 32 (same parity: same row)
            wave_order();
            if (jj + 32 < Cc) {
                const int c = jj + 32;
                const float4 *r4 = reinterpret_cast<const float4 *>((jj & 1) ? drow : srow);
                const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + c * L::kCosPitch);
                float acc = 0.f;
                const int nq = (Mc + 3) / 4;
                for (int i = 0; i < nq; ++i) {
                    const float4 r = r4[i], cw = c4[i];
                    acc = fmaf(r.x, cw.x, acc);
                    acc = fmaf(r.y, cw.y, acc);
                    acc = fmaf(r.z, cw.z, acc);
                    acc = fmaf(r.w, cw.w, acc);
                }
                if (live) a.out[static_cast<unsigned long long>(gf) * Cc + c] = acc * a.dct_scale_k;
            }
        }
        wave_order();
        unit = next;
    }
}

// ss_mel_c512: the mel-spectrogram path (frame_analysis + |X wnorm|^2 + mel einsum, functions.rs:125-170, feature.rs:151-174)
// at fft_points = 1024 on the same FFT mapping: 32 lanes own a 1024-sample window, a wave carries two consecutive rows of
// one clip, the work unit is a (clip, row pair).  Row r covers the 1024 samples that end at chunk r + n_pad: zero initial
// state, zero tail, rows past the real ones all zero (D3); windows inside the clip load at constant offsets, clip edges
// through one masked range per lane.  The table block is mfcc1024_layout with the Vorbis window in kWin (kCos unused).
template <int WAVES, bool STFT>
__global__ __launch_bounds__(WAVES * 64) void ss_mel_c512(const Mel2048Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int fr = lane >> 5;                 // row within the pair
    const int jj = lane & 31;                 // lane within the row
    const int k1 = jj & 15, h = jj >> 4;      // reader view: column k1, half a = h
    const int cls = jj & 1, bw = jj >> 1;     // writer view: n1 = jj = cls + 2 bw

    float *fbase = reinterpret_cast<float *>(smem) + wave * kWaveFloatsK + fr * kFrameFloats;
    float2 *exf = reinterpret_cast<float2 *>(fbase);
    float *prow = fbase;  // [516] after the exchange
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWaveFloatsK;
    const float4 *s_t1 = reinterpret_cast<const float4 *>(s_tab + L::kT1);
    const float2 *s_t2 = reinterpret_cast<const float2 *>(s_tab + L::kT2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kWin);
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_melw = s_tab + L::kMelW;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 32 * a.mel_wpitch);

    const int R = static_cast<int>(a.rows), Rreal = static_cast<int>(a.real_rows), M = static_cast<int>(a.n_filters);
    const unsigned pairs = (a.rows + 1) / 2;
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * pairs;
    const unsigned u_lo = static_cast<unsigned>(units * blockIdx.x / gridDim.x);
    const unsigned u_hi = static_cast<unsigned>(units * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 32 * a.mel_wpitch) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = u_lo + WAVES;
    }
    __syncthreads();
    int st[4], fi[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st[s] = s_start[s * 32 + jj];
        fi[s] = s_filt[s * 32 + jj];
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + jj * a.mel_wpitch);
    const int paddr = ((lane & 32) | ((16 - k1) & 15) | ((1 - h) << 4)) << 2;  // lane holding Z[512 - k]
    float2 *exw = exf + cls * kClsK + 34 * (bw >> 1) + (bw & 1);  // writer base (float2 units)
    const float2 *exr = exf + h * kClsK + 2 * k1;                  // reader base
    const float hs = 0.25f * a.scale * a.scale;                    // |X wnorm|^2 = (wnorm^2 / 4) |2X|^2
    const bool k1z = k1 == 0;

    unsigned unit = __builtin_amdgcn_readfirstlane(u_lo + wave);  // uniform: kept scalar
    while (unit < u_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);

        const unsigned clip = unit / pairs;
        const int r = static_cast<int>(unit - clip * pairs) * 2 + fr;
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        const bool active = r < Rreal;
        // functions.rs:137-151: the window covers the last 1024 samples ending at chunk r + n_pad
        const int start = static_cast<int>(r + a.n_pad + 1) * static_cast<int>(a.hop) - 1024;
        const bool inside = active && start >= 0 && start + 1024 <= static_cast<int>(a.n_samples);
        const float2 *src = reinterpret_cast<const float2 *>(xc + start) + jj;
        float2 v[16];
        if (__all(inside)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = src[32 * e];
        } else {
            // clip edges and inactive rows: start and n_samples are even, the valid sample pairs form one range per lane
            // (the address of a masked load may lie outside the clip; it is never dereferenced)
            const int base = start + 2 * jj;
            const int n = static_cast<int>(a.n_samples);
            if (((start | n) & 1) == 0) {
                const int e_lo = base >= 0 ? 0 : (63 - base) >> 6;
                int e_hi = base >= n ? 0 : min(16, (n - base + 63) >> 6);
                if (!active) e_hi = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float2 s = make_float2(0.f, 0.f);
                    if (e >= e_lo && e < e_hi) s = src[32 * e];
                    v[e] = s;
                }
            } else {  // odd hop or clip length: a pair may straddle the clip edge, bounds per sample
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int p0 = base + 64 * e;
                    v[e] = make_float2(active && p0 >= 0 && p0 < n ? xc[p0] : 0.f, active && p0 + 1 >= 0 && p0 + 1 < n ? xc[p0 + 1] : 0.f);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float2 w = s_win[jj + 32 * e];
            v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
        }
        // ---- 512-point complex FFT as in ss_mfcc_c512 ----
        fft_reg<16>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) exw[2 * k] = v[k];
        wave_order();
        float2 u[16];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
            u[2 * p] = make_float2(t4.x, t4.y);
            u[2 * p + 1] = make_float2(t4.z, t4.w);
        }
        wave_order();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 w2 = s_t1[p * 16 + k1];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 7) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
        }
        fft_reg<16>(u);
        float2 r0[8], r1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float px = u[i].x, qx = u[8 + i].x, py = u[i].y, qy = u[8 + i].y;
            swap_rows(px, qx);
            swap_rows(py, qy);
            const float2 wq = cmul(make_float2(qx, qy), s_t2[i * 32 + jj]);
            r0[i] = make_float2(px + wq.x, py + wq.y);
            r1[i] = make_float2(px - wq.x, py - wq.y);
        }
        // ---- untangle Z -> X; (|X| wnorm)^2 (functions.rs:166-169 + feature.rs:164) ----
        float2 zcs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2 sv = k1z ? r1[(8 - i) & 7] : r1[7 - i];
            zcs[i] = make_float2(bperm(paddr, sv.x), bperm(paddr, sv.y));
        }
        const int kb = k1 + 128 * h;
        // stft build (functions.rs:86-123, :166-169): X[k] * wnorm for all 513 bins of the row, interleaved re / im
        float2 *srow = nullptr;
        if (STFT && r < R) srow = reinterpret_cast<float2 *>(a.out) + (static_cast<unsigned long long>(clip) * R + r) * 513ull;
        const float cs = 0.5f * a.scale;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2 zk = r0[i];
            float2 zc = zcs[i];
            if (i == 0) zc = k1z ? (h ? r1[0] : zk) : zc;
            const float2 w = s_twn[i * 32 + jj];
            const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
            const float2 dd = make_float2(zk.x - zc.x, zk.y + zc.y);
            // 2 X[k] = s - i w dd, 2 conj X[512-k] = 2 s - 2 X[k]
            const float xr = fmaf(w.y, dd.x, fmaf(w.x, dd.y, s.x));
            const float xi = fmaf(w.y, dd.y, fmaf(-w.x, dd.x, s.y));
            if (STFT) {
                if (srow) {
                    srow[kb + 16 * i] = make_float2(cs * xr, cs * xi);
                    srow[512 - (kb + 16 * i)] = make_float2(cs * fmaf(2.f, s.x, -xr), -cs * fmaf(2.f, s.y, -xi));
                }
                continue;
            }
            prow[kb + 16 * i] = hs * fmaf(xr, xr, xi * xi);
            if (a.fullp) {  // the bank reaches past (F+1)/2: bins 257..512 as well
                const float yr = fmaf(2.f, s.x, -xr), yi = fmaf(2.f, s.y, -xi);
                prow[512 - (kb + 16 * i)] = hs * fmaf(yr, yr, yi * yi);
            }
        }
        if (jj == 0) {
            const float2 z = r1[0];  // X[256] = conj Z[256]
            if (STFT) {
                if (srow) srow[256] = make_float2(a.scale * z.x, -a.scale * z.y);
            } else {
                prow[256] = hs * 4.f * fmaf(z.x, z.x, z.y * z.y);
            }
        }
        if (STFT) {
            wave_order();
            unit = next;
            continue;
        }
        if (jj < 3) prow[(a.fullp ? 513 : 257) + jj] = 0.f;  // pad bins read (with zero weight) by the mel stage
        wave_order();
        // ---- banded mel reduction (feature.rs:173); the two rows of the wave are adjacent words of out[clip][m][.] ----
        if (r < R) {
            float *dst = a.out + static_cast<unsigned long long>(clip) * M * R + r;
            int off = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float m = mel_slot4(w4 + off, reinterpret_cast<const float4 *>(prow + st[s]), a.mel_q4[s]);
                if (fi[s] >= 0) dst[static_cast<unsigned long long>(fi[s]) * R] = m;
                off += a.mel_q4[s];
            }
        }
        wave_order();
        unit = next;
    }
}

template <int WAVES>
hipError_t launch_k(const Mfcc1024Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloatsK + L::kMelW + 32 * static_cast<size_t>(a.mel_wpitch) + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    if (total >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned long long units = (total + 1) / 2;
    unsigned long long blocks = (units + WAVES - 1) / WAVES;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(WAVES * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
        return hipGetLastError();
    };
    const bool pow2 = a.spectrum_exponent == 2, win = a.windowed != 0, lib = a.center != 0 || a.fullp != 0;
#define SS_K(P, M, W, LB, NAME) go(ss_mfcc_c512<P, M, W, WAVES, LB>, NAME)
    if (lib) {
        if (a.out_mfe) {
            if (pow2) return win ? SS_K(true, true, true, true, "ss_mfcc_c512<pow2,mfe,win,lib>") : SS_K(true, true, false, true, "ss_mfcc_c512<pow2,mfe,lib>");
            return win ? SS_K(false, true, true, true, "ss_mfcc_c512<mfe,win,lib>") : SS_K(false, true, false, true, "ss_mfcc_c512<mfe,lib>");
        }
        if (pow2) return win ? SS_K(true, false, true, true, "ss_mfcc_c512<pow2,win,lib>") : SS_K(true, false, false, true, "ss_mfcc_c512<pow2,lib>");
        return win ? SS_K(false, false, true, true, "ss_mfcc_c512<win,lib>") : SS_K(false, false, false, true, "ss_mfcc_c512<lib>");
    }
    if (a.out_mfe) {
        if (pow2) return win ? SS_K(true, true, true, false, "ss_mfcc_c512<pow2,mfe,win>") : SS_K(true, true, false, false, "ss_mfcc_c512<pow2,mfe>");
        return win ? SS_K(false, true, true, false, "ss_mfcc_c512<mfe,win>") : SS_K(false, true, false, false, "ss_mfcc_c512<mfe>");
    }
    if (pow2) return win ? SS_K(true, false, true, false, "ss_mfcc_c512<pow2,win>") : SS_K(true, false, false, false, "ss_mfcc_c512<pow2>");
    return win ? SS_K(false, false, true, false, "ss_mfcc_c512<win>") : SS_K(false, false, false, false, "ss_mfcc_c512");
#undef SS_K
}

}  // namespace

hipError_t launch_mfcc_c512(const Mfcc1024Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    return launch_k<12>(a, stream, num_cus, info);
}

hipError_t launch_mel_c512(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    constexpr int WAVES = 12;
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloatsK + L::kMelW + 32 * static_cast<size_t>(a.mel_wpitch) + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.batch == 0 || a.rows == 0) return hipSuccess;
    const unsigned cap = static_cast<unsigned>(num_cus > 0 ? num_cus : 256);
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * ((a.rows + 1) / 2);
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
    return a.out_stft ? go(ss_mel_c512<WAVES, true>, "ss_mel_c512<stft>") : go(ss_mel_c512<WAVES, false>, "ss_mel_c512");
}

}  // namespace ss

// ss_mfcc_c1024: fused MFCC / mfe for fft_points = 2048 (C = 1024 packed complex points) on gfx950 -- the frame path
// (feature.rs:99-148, :200-233; processing.rs:65-181) one size up from ss_mfcc512.hip and one down from ss_mfcc4096.hip.
//
//   * 32 lanes own a frame, 32 complex points per lane; a wave carries two consecutive frames of the flat frame list.  One
//     persistent 8-wave workgroup per CU; waves pull frame pairs from an LDS counter.
//   * FFT exactly as in ss_mel2048.hip: radix-32, ONE transposing exchange through wave-private LDS in two register halves
//     (ds_write_b64 scatter to 34*(n1>>1) + 2*k1' + (n1&1), ds_read_b128 back), twiddle, radix-32.  No workgroup barrier
//     in the main loop.
//   * untangle with ds_bpermute_b32 (partner = lane 32-j, register 31-r): bins 0..512 go to the P row (the mel bank ends at
//     (F+1)/2, feature.rs:69-70), all 1025 feed the frame energy (feature.rs:216-219), summed over the half-wave.
//   * banded mel (4 filters per lane, aligned float4 taps), zero handling, ln -> row in filter order; DCT-II with the
//     m <-> M-1-m symmetry of the cosine (sum / difference rows, half the table), reference scaling, column-0 replacement.
//   * optional frame window (mfcc_window switch) from the table block; mfe build stops after the mel stage.
//   * LIB builds (librosa-compatible switches): centred frames with reflect / zero padding at the clip edges, and P rows of
//     all 1025 bins for banks over the whole spectrum; the rows then outgrow the exchange region (10304 B per wave).
// Reference semantics as in ss_mfcc512.hip; tables: ss::mfcc2048_layout (ss_internal.h).
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

namespace ss {

namespace {

using namespace wv;

namespace L = mfcc2048_layout;
constexpr int kExSlotsG = 2 * 16 * 34;      // float2 in the wave's exchange region: two frames x half the columns (8704 B)
constexpr int kWaveFloatsG = kExSlotsG * 2;
constexpr int kHalfFloats = kWaveFloatsG / 2;  // per frame after the exchange: P row [520] | ln(mel) row [128] | s [64] | d [64]
constexpr int kHalfFloatsLib = 1288;           // LIB: P row [1028] | ln(mel) row [128] | s [64] | d [64]
template <bool LIB> constexpr int wave_floats() { return LIB ? 2 * kHalfFloatsLib : kWaveFloatsG; }


template <bool POW2, bool MFE, bool WIN, int WAVES, bool LIB = false, bool PRE = false>
__global__ __launch_bounds__(WAVES * 64) void ss_mfcc_c1024(const Mfcc2048Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int half = lane >> 5;  // frame within the wave
    const int j = lane & 31;     // lane within the frame

    constexpr int kWF = wave_floats<LIB>();
    float *wbase = reinterpret_cast<float *>(smem) + wave * kWF;
    float2 *ex = reinterpret_cast<float2 *>(wbase);
    float *hb = wbase + half * (LIB ? kHalfFloatsLib : kHalfFloats);  // this frame's rows after the exchange
    float *prow = hb, *frow = hb + (LIB ? 1028 : 520), *srow = frow + 128, *drow = srow + 64;
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWF;
    const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2);
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
        st[s] = s_start[s * 32 + j];
        fi[s] = s_filt[s * 32 + j];
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + j * a.mel_wpitch);
    const int paddr = ((lane & 32) | ((32 - j) & 31)) << 2;  // lane holding Z[1024 - k]
    const float hscale32 = (POW2 ? 0.25f * a.scale : 0.5f * a.scale) * kTwo32;
    const int M = static_cast<int>(a.n_filters), Cc = static_cast<int>(a.n_ceps), Mh = M / 2, Mc = (M + 1) / 2;
    // valid sample pairs of this lane: n = j + 32 e with 2 n < flen (zero pad to fft_points, processing.rs:147-156)
    const int e_hi = min(32, max(0, (static_cast<int>(a.flen) / 2 - j + 31) >> 5));
    const int half_pairs = static_cast<int>(a.flen) / 2;
    const bool odd_tail = (a.flen & 1) != 0;
    constexpr bool pre = PRE;  // fused pre-emphasis: builds of their own (LIB layout: they also serve centred frames)
    const unsigned psh = PRE ? a.preemph_shift % a.n_samples : 0u;

    unsigned unit = __builtin_amdgcn_readfirstlane(u_lo + wave);  // uniform: kept scalar
    while (unit < u_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);

        const unsigned gf_raw = 2 * unit + half;
        const bool live = gf_raw < total;
        const unsigned gf = live ? gf_raw : total - 1;
        const unsigned clip = gf / a.n_frames;
        const unsigned t = gf - clip * a.n_frames;
        // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        float2 v[32];
        // librosa center=True: frame t is centred on sample t*step; frames inside the clip load like contract frames from
        // their (even) start, the few at the clip edges mirror (np.pad 'reflect') or zero their out-of-range samples
        const int s0 = static_cast<int>(t * a.step) - (LIB && a.center ? static_cast<int>(a.flen / 2) : 0);
        const int ns = static_cast<int>(a.n_samples);
        if (!LIB || __all(s0 >= 0 && s0 + static_cast<int>(a.flen) <= ns)) {
            const float2 *src = reinterpret_cast<const float2 *>(xc + s0) + j;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                float2 s = make_float2(0.f, 0.f);
                if (e < e_hi) s = src[32 * e];
                // odd frame length: the last sample is the first half of a pair (its partner is zero padding)
                if (odd_tail && j + 32 * e == half_pairs) s = make_float2(xc[s0 + 2 * half_pairs], 0.f);
                if (pre) {  // fused pre-emphasis (processing.rs:31-53) of the samples that exist
                    const int pos = s0 + 2 * (j + 32 * e), rem = static_cast<int>(a.flen) - 2 * (j + 32 * e);
                    if (rem >= 1) s.x = fmaf(-a.preemph, preemph_tap(xc, pos, psh, a.n_samples), s.x);
                    if (rem >= 2) s.y = fmaf(-a.preemph, preemph_tap(xc, pos + 1, psh, a.n_samples), s.y);
                }
                v[e] = s;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                float sv[2] = {0.f, 0.f};
                {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        int pos = s0 + 2 * (j + 32 * e) + h;
                        bool ok = 2 * (j + 32 * e) + h < static_cast<int>(a.flen);  // zero pad; an odd frame length ends in a half pair
                        if (pos < 0 || pos >= ns) {
                            if (a.pad_reflect) pos = pos < 0 ? -pos : 2 * (ns - 1) - pos;
                            else ok = false;
                        }
                        if (ok) sv[h] = pre ? fmaf(-a.preemph, preemph_tap(xc, pos, psh, a.n_samples), xc[pos]) : xc[pos];
                    }
                }
                v[e] = make_float2(sv[0], sv[1]);
            }
        }
        if (WIN) {
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const float2 w = s_win[j + 32 * e];
                v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
            }
        }
        // ---- 1024-point complex FFT: radix-32, transpose through LDS (two register halves), twiddle, radix-32 ----
        fft_reg<32>(v);
        float2 u[32];
        float2 *exf = ex + half * (16 * 34);
        const int wbh = 34 * (j >> 1) + (j & 1);
        const int jl = j & 15;
#pragma unroll
        for (int k = 0; k < 16; ++k) exf[wbh + 2 * k] = v[k];
        wave_order();
        if (j < 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exf[34 * p + 2 * jl]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
        }
        wave_order();
#pragma unroll
        for (int k = 0; k < 16; ++k) exf[wbh + 2 * k] = v[16 + k];
        wave_order();
        if (j >= 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exf[34 * p + 2 * jl]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
        }
        wave_order();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float4 w2 = s_tw2[p * 32 + j];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 15) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
        }
        fft_reg<32>(u);  // u[r] = Z[j + 32 r]

        // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
        float esum = 0.f;
#pragma unroll
        for (int hb2 = 0; hb2 < 2; ++hb2) {
            float2 zcs[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) zcs[q] = make_float2(bperm(paddr, u[31 - (8 * hb2 + q)].x), bperm(paddr, u[31 - (8 * hb2 + q)].y));
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const int q = 8 * hb2 + qq;
                const float2 zk = u[q];
                // lane 0 pairs with itself: Z[1024 - 32 q] = own register (32 - q) & 31
                const float2 zc = j == 0 ? u[(32 - q) & 31] : zcs[qq];
                const float2 w = s_twn[q * 32 + j];
                const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                const float2 d = make_float2(zk.x - zc.x, zk.y + zc.y);
                // 2 X[k] = s - i w d, 2 conj X[1024-k] = 2 s - 2 X[k]
                const float xa_r = fmaf(w.y, d.x, fmaf(w.x, d.y, s.x));
                const float xa_i = fmaf(w.y, d.y, fmaf(-w.x, d.x, s.y));
                const float xb_r = fmaf(2.f, s.x, -xa_r), xb_i = fmaf(2.f, s.y, -xa_i);
                const float na = xa_r * xa_r + xa_i * xa_i, nb = xb_r * xb_r + xb_i * xb_i;
                const float pa = POW2 ? na : __builtin_amdgcn_sqrtf(na);
                const float pb = POW2 ? nb : __builtin_amdgcn_sqrtf(nb);
                prow[j + 32 * q] = pa;  // bins 0..511 (the bank ends at (F+1)/2, feature.rs:69-70); 512 below
                if (LIB) prow[1024 - (j + 32 * q)] = pb;  // bins 513..1024 for banks over the whole spectrum
                esum += pa + pb;        // X[0] and X[1024] come from lane 0's self pair (q = 0)
            }
        }
        if (j == 0) {
            const float2 z = u[16];  // X[512] = conj Z[512]
            const float n = 4.f * (z.x * z.x + z.y * z.y);
            const float p512 = POW2 ? n : __builtin_amdgcn_sqrtf(n);
            prow[512] = p512;
            esum += p512;
        }
        if (j < 3) prow[(LIB ? 1025 : 513) + j] = 0.f;  // pad bins read (with zero weight) by the mel stage
        float energy = hscale32 * half_sum(esum);              // E * 2^32
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
            if (j == 0 && live) a.out_energy[gf] = energy * (1.0f / kTwo32);
            wave_order();
            unit = next;
            continue;
        }
        wave_order();
        // ---- DCT-II (feature.rs:120-123), cos(pi c (2(M-1-m)+1) / 2M) = (-1)^c cos(pi c (2m+1) / 2M), M even: sum and
        // difference rows once per frame, then a M/2-term product per coefficient ----
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int m = j + 32 * h2;
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
        if (j < Cc) {
            const int c = j;
            const float4 *r4 = reinterpret_cast<const float4 *>((j & 1) ? drow : srow);
            const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + c * L::kCosPitch);
            float acc = 0.f;
            const int nq = (Mc + 3) / 4;  // the rows are zero-padded to a multiple of 4 by the host table / the loop below
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
            if (j + 32 < Cc) {
                const int c = j + 32;
                const float4 *r4 = reinterpret_cast<const float4 *>((j & 1) ? drow : srow);
                const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + c * L::kCosPitch);
                float acc = 0.f;
                const int nq = (Mc + 3) / 4;  // the rows are zero-padded to a multiple of 4 by the host table / the loop below
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

template <int WAVES>
hipError_t launch_g(const Mfcc2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const bool lib = a.center != 0 || a.fullp != 0 || a.preemph != 0.f;  // the pre-emphasis builds use the LIB layout
    const size_t lds = (static_cast<size_t>(WAVES) * (lib ? wave_floats<true>() : wave_floats<false>()) + L::kMelW + 32 * static_cast<size_t>(a.mel_wpitch) + 4) * sizeof(float);
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
    const bool pow2 = a.spectrum_exponent == 2, win = a.windowed != 0;
#define SS_G(P, M, W, LB, NAME) go(ss_mfcc_c1024<P, M, W, WAVES, LB>, NAME)
#define SS_GP(P, M, W, NAME) go(ss_mfcc_c1024<P, M, W, WAVES, true, true>, NAME)
    if (a.preemph != 0.f) {
        if (a.out_mfe) {
            if (pow2) return win ? SS_GP(true, true, true, "ss_mfcc_c1024<pow2,mfe,win,lib,pre>") : SS_GP(true, true, false, "ss_mfcc_c1024<pow2,mfe,lib,pre>");
            return win ? SS_GP(false, true, true, "ss_mfcc_c1024<mfe,win,lib,pre>") : SS_GP(false, true, false, "ss_mfcc_c1024<mfe,lib,pre>");
        }
        if (pow2) return win ? SS_GP(true, false, true, "ss_mfcc_c1024<pow2,win,lib,pre>") : SS_GP(true, false, false, "ss_mfcc_c1024<pow2,lib,pre>");
        return win ? SS_GP(false, false, true, "ss_mfcc_c1024<win,lib,pre>") : SS_GP(false, false, false, "ss_mfcc_c1024<lib,pre>");
    }
#undef SS_GP
    if (lib) {
        if (a.out_mfe) {
            if (pow2) return win ? SS_G(true, true, true, true, "ss_mfcc_c1024<pow2,mfe,win,lib>") : SS_G(true, true, false, true, "ss_mfcc_c1024<pow2,mfe,lib>");
            return win ? SS_G(false, true, true, true, "ss_mfcc_c1024<mfe,win,lib>") : SS_G(false, true, false, true, "ss_mfcc_c1024<mfe,lib>");
        }
        if (pow2) return win ? SS_G(true, false, true, true, "ss_mfcc_c1024<pow2,win,lib>") : SS_G(true, false, false, true, "ss_mfcc_c1024<pow2,lib>");
        return win ? SS_G(false, false, true, true, "ss_mfcc_c1024<win,lib>") : SS_G(false, false, false, true, "ss_mfcc_c1024<lib>");
    }
    if (a.out_mfe) {
        if (pow2) return win ? SS_G(true, true, true, false, "ss_mfcc_c1024<pow2,mfe,win>") : SS_G(true, true, false, false, "ss_mfcc_c1024<pow2,mfe>");
        return win ? SS_G(false, true, true, false, "ss_mfcc_c1024<mfe,win>") : SS_G(false, true, false, false, "ss_mfcc_c1024<mfe>");
    }
    if (pow2) return win ? SS_G(true, false, true, false, "ss_mfcc_c1024<pow2,win>") : SS_G(true, false, false, false, "ss_mfcc_c1024<pow2>");
    return win ? SS_G(false, false, true, false, "ss_mfcc_c1024<win>") : SS_G(false, false, false, false, "ss_mfcc_c1024");
#undef SS_G
}

}  // namespace

hipError_t launch_mfcc_c1024(const Mfcc2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    return launch_g<8>(a, stream, num_cus, info);
}

}  // namespace ss

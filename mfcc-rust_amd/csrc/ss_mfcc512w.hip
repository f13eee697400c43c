// ss_mfcc_c256w: fused MFCC / mfe for fft_points = 512 with WIDE banks -- up to 80 filters (16 kHz log-mel front ends with
// 64 / 80 mels, 25 ms frames) -- on gfx950.  ss_mfcc512.hip, the headline kernel, is specialised for up to 48 filters; this
// one trades its register-resident tables for five filters per lane and a P row of all 257 bins.
//
//   * Work unit: a QUAD of 4 consecutive frames of the flat frame list; 16 lanes (one DPP row) own a frame = 256 packed
//     complex points, 16 per lane.  One persistent 12-wave workgroup per CU, quads from an LDS counter, the next quad's
//     samples prefetched into the dead input registers.  Zero padding is compile-time (template NE).
//   * FFT / untangle exactly as in ss_mfcc512.hip (radix-16, one transposing exchange through the frame's wave-private
//     2304-B slot, twiddle, radix-16; partner Z[256-k] by ds_bpermute_b32).  All 257 bins go to the P row inside the slot
//     and to the frame energy.
//   * banded mel, five filters per lane (host-sorted by tap count), zero handling, ln -> (slot, lane)-ordered row of 80;
//     DCT-II as an 80-term product per lane with the lane's cosine row; reference scaling and column-0 replacement.
//     mfe builds stop after the mel stage.  Optional frame window from the table block; centred frames (librosa
//     center=True, reflect / zero padding at the clip edges) at run time.  Up to 32 cepstra (two coefficients
//     per lane beyond 16), which also brings configurations with at most 48 filters but more than 16 cepstra here.
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.  Tables: ss::mfcc512w_layout.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

namespace ss {

namespace {

using namespace wv;

namespace L = mfcc512w_layout;
constexpr int kSlotFloats = 576;        // per frame: exchange slot (288 float2); afterwards P row [260] | ln(mel) row [80]
constexpr int kWaveFloatsX = 4 * kSlotFloats;
constexpr int kPRowX = 260;             // bins 0..256 + three zero pad bins



// Issues the loads of one quad: vin[e] = (x[2n], x[2n+1]), n = j + 16 e, of this lane group's frame (zero beyond flen,
// processing.rs:147-156).  Returns the frame's index within its clip.
template <int NE>
__device__ __forceinline__ unsigned load_quad_w(const Mfcc256Args &a, unsigned quad, unsigned total, int f, int j, float2 (&vin)[NE])
{
    const unsigned q4 = quad * 4;  // uniform
    const unsigned fl = min(static_cast<unsigned>(f), total - 1 - q4);  // lanes past the last frame redo it
    unsigned clip, t;
    if (a.nf_magic) {
        // scalar quotient of the quad's first frame (multiply-high by the host's reciprocal), one conditional wrap per lane
        clip = __umulhi(q4, a.nf_magic) >> a.nf_shift;
        t = q4 - clip * a.n_frames + fl;
        const bool wrap = t >= a.n_frames;
        t -= wrap ? a.n_frames : 0u;
        clip += wrap ? 1u : 0u;
    } else {
        const unsigned gf = q4 + fl;
        clip = gf / a.n_frames;
        t = gf - clip * a.n_frames;
    }
    const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
    const bool pre = a.preemph != 0.f;  // fused pre-emphasis: the caller then loads at the top of the loop (no prefetch)
    const unsigned sh = a.preemph_shift % a.n_samples;
    // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step.  librosa center=True (the
    // `framing` switch): frame t is centred on sample t*step; frames inside the clip load the same way from their (even)
    // start, the few at the clip edges mirror (np.pad 'reflect') or zero their out-of-range samples
    const int s0 = static_cast<int>(t * a.step) - (a.center ? static_cast<int>(a.flen / 2) : 0), ns = static_cast<int>(a.n_samples);
    if (!a.center || __all(s0 >= 0 && s0 + static_cast<int>(a.flen) <= ns)) {
        const float2 *src = reinterpret_cast<const float2 *>(xc + s0) + j;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            // zero pad beyond flen; an odd frame length ends in a half pair
            const int rem = static_cast<int>(a.flen) - 2 * (j + 16 * e);
            vin[e] = rem >= 2 ? src[16 * e] : make_float2(rem == 1 ? reinterpret_cast<const float *>(src)[32 * e] : 0.f, 0.f);
            if (pre) {
                const int pos = s0 + 2 * (j + 16 * e);
                if (rem >= 1) vin[e].x = fmaf(-a.preemph, preemph_tap(xc, pos, sh, a.n_samples), vin[e].x);
                if (rem >= 2) vin[e].y = fmaf(-a.preemph, preemph_tap(xc, pos + 1, sh, a.n_samples), vin[e].y);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int n = j + 16 * e;
            float sv[2] = {0.f, 0.f};
            if (2 * n < static_cast<int>(a.flen)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int pos = s0 + 2 * n + h;
                    bool ok = 2 * n + h < static_cast<int>(a.flen);  // an odd frame length ends in a half pair
                    if (pos < 0 || pos >= ns) {
                        if (a.pad_reflect) pos = pos < 0 ? -pos : 2 * (ns - 1) - pos;
                        else ok = false;
                    }
                    if (ok) sv[h] = pre ? fmaf(-a.preemph, preemph_tap(xc, pos, sh, a.n_samples), xc[pos]) : xc[pos];
                }
            }
            vin[e] = make_float2(sv[0], sv[1]);
        }
    }
    return t;
}

template <int NE, bool POW2, bool MFE, bool WIN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ss_mfcc_c256w(const Mfcc256Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;  // frame within the quad
    const int j = lane & 15;  // lane within the frame (DPP row)

    float *slot = reinterpret_cast<float *>(smem) + wave * kWaveFloatsX + f * kSlotFloats;
    float2 *zh = reinterpret_cast<float2 *>(slot);
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWaveFloatsX;
    const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const float *s_cos = s_tab + L::kCos;
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_melw = s_tab + L::kMelW;
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kMelW + 16 * a.mel_wpitch);  // WIN: 256 window sample pairs (zero beyond flen)
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 16 * a.mel_wpitch + (WIN ? 512 : 0));

    const unsigned total = a.batch * a.n_frames;
    const unsigned quads = (total + 3) / 4;
    const unsigned q_lo = static_cast<unsigned>(static_cast<unsigned long long>(quads) * blockIdx.x / gridDim.x);
    const unsigned q_hi = static_cast<unsigned>(static_cast<unsigned long long>(quads) * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 16 * a.mel_wpitch + (WIN ? 512 : 0)) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = q_lo + WAVES;
    }
    unsigned quad = __builtin_amdgcn_readfirstlane(q_lo + wave);  // uniform: kept scalar
    float2 vin[NE];
    unsigned t_next = 0;
    const bool pre = a.preemph != 0.f;  // pre-emphasised samples are formed at load time: no prefetch across the iteration then
    if (!pre && quad < q_hi) t_next = load_quad_w<NE>(a, quad, total, f, j, vin);

    const int paddr = ((lane & 48) | ((16 - j) & 15)) << 2;  // lane holding Z[256 - k]
    const int wbase1 = 34 * (j >> 1) + (j & 1);              // exchange write base (float2 units)
    const int Cc = static_cast<int>(a.n_ceps), M = static_cast<int>(a.n_filters);
    // |X| = (1/2)|2X|: the 1/2 of the untangle is folded into the scale (1/4 for the squared form)
    const float hscale32 = (POW2 ? 0.25f * a.scale : 0.5f * a.scale) * kTwo32;
    __syncthreads();
    int st[5], fi[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        st[s] = s_start[s * 16 + j];
        fi[s] = s_filt[s * 16 + j];
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + j * a.mel_wpitch);
    const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + j * L::kCosPitch);
    float2 twn[8];  // exp(-2 pi i (j + 16 r) / 512) stays in registers
#pragma unroll
    for (int r = 0; r < 8; ++r) twn[r] = s_twn[r * 16 + j];

    while (quad < q_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);
        if (pre) t_next = load_quad_w<NE>(a, quad, total, f, j, vin);
        const unsigned t_cur = t_next;

        float2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float2 s = e < NE ? vin[e] : make_float2(0.f, 0.f);
            if (WIN && e < NE) {
                const float2 w = s_win[j + 16 * e];
                s = make_float2(s.x * w.x, s.y * w.y);
            }
            v[e] = s;
        }
        // ---- 256-point complex FFT of the packed frame: radix-16, transpose through LDS, twiddle, radix-16 ----
        fft16_reg(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) zh[wbase1 + 2 * r] = v[r];
        wave_order();
        if (!pre && next < q_hi) t_next = load_quad_w<NE>(a, next, total, f, j, vin);
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
        fft16_reg(u);  // u[r] = Z[j + 16 r]

        // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
        float2 zcs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) zcs[r] = make_float2(bperm(paddr, u[15 - r].x), bperm(paddr, u[15 - r].y));
        float *prow = slot;
        float esum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float2 zk = u[r];
            // lane 0 pairs with itself: Z[256 - 16 r] = own register (16 - r) & 15
            const float2 zc = j == 0 ? u[(16 - r) & 15] : zcs[r];
            const float2 w = twn[r];
            const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
            const float2 d = make_float2(zk.x - zc.x, zk.y + zc.y);
            // 2 X[k] = s - i w d, 2 conj X[256-k] = 2 s - 2 X[k]
            const float xa_r = fmaf(w.y, d.x, fmaf(w.x, d.y, s.x));
            const float xa_i = fmaf(w.y, d.y, fmaf(-w.x, d.x, s.y));
            const float xb_r = fmaf(2.f, s.x, -xa_r), xb_i = fmaf(2.f, s.y, -xa_i);
            const float na = fmaf(xa_r, xa_r, xa_i * xa_i), nb = fmaf(xb_r, xb_r, xb_i * xb_i);
            const float pa = POW2 ? na : __builtin_amdgcn_sqrtf(na);  // unscaled; hscale is applied to the sums below
            const float pb = POW2 ? nb : __builtin_amdgcn_sqrtf(nb);
            prow[j + 16 * r] = pa;
            prow[256 - j - 16 * r] = pb;
            esum += pa + pb;
        }
        if (j == 0) {
            // lane 0's pair k = 0 produced X[0] and X[256]; X[128] = conj Z[128] is the one extra bin
            const float2 z = u[8];
            const float n = 4.f * fmaf(z.x, z.x, z.y * z.y);
            const float p128 = POW2 ? n : __builtin_amdgcn_sqrtf(n);
            prow[128] = p128;
            esum += p128;
        }
        if (j < 3) prow[257 + j] = 0.f;  // pad bins read (with zero weight) by the mel stage
        float en = hscale32 * row16_sum(esum);        // E * 2^32
        en = en == 0.f ? kEps * kTwo32 : en;          // zero_handling, feature.rs:219
        wave_order();

        // ---- banded mel reduction (feature.rs:229), zero handling (:230), ln (:105) ----
        {
            float *frow = slot + kPRowX;
            float m[5];
            int off = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                m[k] = hscale32 * mel_slot1(w4 + off, prow + st[k], a.mel_q4[k]);
                m[k] = m[k] == 0.f ? kEps * kTwo32 : m[k];
                off += a.mel_q4[k];
            }
            const unsigned gf = quad * 4 + f;
            if (MFE) {
                if (gf < total) {
                    float *row = a.out + static_cast<unsigned long long>(gf) * M;
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        if (fi[k] >= 0) row[fi[k]] = m[k] * (1.0f / kTwo32);  // exact: power of two
                    if (j == 0) a.out_energy[gf] = en * (1.0f / kTwo32);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 5; ++k) frow[16 * k + j] = ln_scaled(m[k]);
                wave_order();
                // ---- DCT-II, first n_ceps coefficients (feature.rs:120-123): lane c against the 80-entry row ----
                if (Cc <= 16) {
                    float acc = 0.f;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        float4 lq[5], cq[5];
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            lq[i] = *reinterpret_cast<const float4 *>(&frow[4 * (5 * h + i)]);
                            cq[i] = c4[5 * h + i];
                        }
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            acc = fmaf(lq[i].x, cq[i].x, acc);
                            acc = fmaf(lq[i].y, cq[i].y, acc);
                            acc = fmaf(lq[i].z, cq[i].z, acc);
                            acc = fmaf(lq[i].w, cq[i].w, acc);
                        }
                    }
                    // scaling + column-0 replacement (feature.rs:126-146)
                    float o = acc * a.dct_scale_k;
                    if (j == 0) o = a.dc_elimination ? ln_scaled(en) : acc * (t_cur == 0 ? a.dct_scale_00 : a.dct_scale_0);
                    if (j < Cc && gf < total) a.out[static_cast<unsigned long long>(gf) * Cc + j] = o;
                } else {
                    // 17..32 cepstra: the lane also forms coefficient 16 + j from the same row fetches
                    const float4 *d4 = c4 + 16 * (L::kCosPitch / 4);
                    float acc = 0.f, acc2 = 0.f;
#pragma unroll 1
                    for (int h = 0; h < 10; ++h) {  // small batches: this path runs next to the prefetch registers
                        float4 lq[2], cq[2], dq[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            lq[i] = *reinterpret_cast<const float4 *>(&frow[4 * (2 * h + i)]);
                            cq[i] = c4[2 * h + i];
                            dq[i] = d4[2 * h + i];
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            acc = fmaf(lq[i].x, cq[i].x, acc);
                            acc = fmaf(lq[i].y, cq[i].y, acc);
                            acc = fmaf(lq[i].z, cq[i].z, acc);
                            acc = fmaf(lq[i].w, cq[i].w, acc);
                            acc2 = fmaf(lq[i].x, dq[i].x, acc2);
                            acc2 = fmaf(lq[i].y, dq[i].y, acc2);
                            acc2 = fmaf(lq[i].z, dq[i].z, acc2);
                            acc2 = fmaf(lq[i].w, dq[i].w, acc2);
                        }
                    }
                    float o = acc * a.dct_scale_k;
                    if (j == 0) o = a.dc_elimination ? ln_scaled(en) : acc * (t_cur == 0 ? a.dct_scale_00 : a.dct_scale_0);
                    if (gf < total) {
                        a.out[static_cast<unsigned long long>(gf) * Cc + j] = o;
                        if (16 + j < Cc) a.out[static_cast<unsigned long long>(gf) * Cc + 16 + j] = acc2 * a.dct_scale_k;
                    }
                }
            }
        }
        wave_order();
        quad = next;
    }
}

template <int WAVES>
hipError_t launch_w5(const Mfcc256Args &a_in, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    Mfcc256Args a = a_in;
    a.nf_magic = 0;
    a.nf_shift = 0;
    {
        // floor(x / d) for x < 2^31 as umulhi(x, ceil(2^(31+l) / d)) >> (l - 1), l = ceil(log2 d) (Granlund-Montgomery);
        // the kernel's one-wrap lane fix-up needs d >= 4
        const unsigned long long tot = static_cast<unsigned long long>(a.batch) * a.n_frames, d = a.n_frames;
        if (d >= 4 && d < (1ull << 31) && tot + 4 < (1ull << 31)) {
            unsigned l = 0;
            while ((1ull << l) < d) ++l;
            const unsigned __int128 num = static_cast<unsigned __int128>(1) << (31 + l);
            a.nf_magic = static_cast<uint32_t>((num + d - 1) / d);
            a.nf_shift = l - 1;
        }
    }
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloatsX + L::kMelW + 16 * static_cast<size_t>(a.mel_wpitch) + (a.windowed ? 512 : 0) + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    if (total + 8 >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned long long quads = (total + 3) / 4;
    unsigned long long blocks = (quads + WAVES - 1) / WAVES;
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
#define SS_W(NE, P, M, W, NAME) go(ss_mfcc_c256w<NE, P, M, W, WAVES>, NAME)
#define SS_WN(NE, TAG)                                                                                                                          \
    if (a.out_mfe) {                                                                                                                            \
        if (pow2) return win ? SS_W(NE, true, true, true, "ss_mfcc_c256w<" TAG ",pow2,mfe,win>") : SS_W(NE, true, true, false, "ss_mfcc_c256w<" TAG ",pow2,mfe>"); \
        return win ? SS_W(NE, false, true, true, "ss_mfcc_c256w<" TAG ",mfe,win>") : SS_W(NE, false, true, false, "ss_mfcc_c256w<" TAG ",mfe>"); \
    }                                                                                                                                           \
    if (pow2) return win ? SS_W(NE, true, false, true, "ss_mfcc_c256w<" TAG ",pow2,win>") : SS_W(NE, true, false, false, "ss_mfcc_c256w<" TAG ",pow2>"); \
    return win ? SS_W(NE, false, false, true, "ss_mfcc_c256w<" TAG ",win>") : SS_W(NE, false, false, false, "ss_mfcc_c256w<" TAG ">");
    if (a.flen <= 320) { SS_WN(10, "10") }
    if (a.flen <= 416) { SS_WN(13, "13") }
    SS_WN(16, "16")
#undef SS_WN
#undef SS_W
}

}  // namespace

hipError_t launch_mfcc_c256w(const Mfcc256Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    return launch_w5<12>(a, stream, num_cus, info);
}

}  // namespace ss

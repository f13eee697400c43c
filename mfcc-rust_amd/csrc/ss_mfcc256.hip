// ss_mfcc_c256x2: fused MFCC / mfe for fft_points = 256 (8 kHz telephony front ends: 20-25 ms frames, 10 ms hop) on gfx950.
//
// The 256-point REAL transform of a frame is not packed into a 128-point complex one here: TWO frames ride one 256-point
// complex transform instead, z[n] = a[n] + i b[n], so the butterfly core is the one of ss_mfcc512.hip and the untangle
// needs no twiddles at all:
//     2 A[k] = Z[k] + conj Z[256-k],   2 B[k] = -i (Z[k] - conj Z[256-k])   =>   |2A| = |s|, |2B| = |d|
// with s, d the sum / difference the 512-point kernel forms anyway.  The price is a rounding error relative to the louder
// frame of a pair; pairs more than 30 dB apart (silence next to speech, zero padding) are therefore transformed one frame
// at a time (kPairGuard).
//
//   * Work unit: an OCT of 8 consecutive frames of the flat frame list; 16 lanes (one DPP row) own a frame pair, 16
//     complex points per lane.  One persistent workgroup per CU, waves pull octs from an LDS counter, the next oct's
//     samples are prefetched into the dead input registers.  Samples load as single floats (one per frame of the pair),
//     so frame starts need no alignment.
//   * FFT: radix-16, one transposing exchange through the pair's wave-private 2304-B slot (ds_write_b64 scatter to
//     34*(n1>>1) + 2*k1 + (n1&1), ds_read_b128 back), twiddle, radix-16.  No workgroup barrier in the main loop.
//   * Partner Z[256-k] by ds_bpermute_b32 (lane 16-j, register 15-r); all 129 bins of both frames go to P rows inside
//     the slot (the exchange is over by then), so banks over the whole spectrum need no separate build.
//   * banded mel (3 filters per lane, host-sorted by tap count), zero handling, ln -> (slot, lane)-ordered row; DCT-II as
//     a 48-term product per lane with the lane's cosine row;
//     reference scaling and column-0 replacement.  mfe builds stop after the mel stage.
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.  Tables: ss::mfcc256_layout.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

namespace ss {

namespace {

using namespace wv;

namespace L = mfcc256_layout;
constexpr int kSlotFloats = 576;        // per frame pair: exchange slot (288 float2); afterwards P rows [2][132] | ln(mel) rows [2][48]
constexpr int kWaveFloatsX = 4 * kSlotFloats;
constexpr int kPRowX = 132;             // bins 0..128 + three zero pad bins
constexpr float kPairGuard = 1000.f;    // energy ratio (30 dB) beyond which the two frames of a pair are transformed one at a time
// A pair transform leaves rounding noise of ~3e-7 of the pair's largest bin in EVERY bin of both frames.  A bin below
// kTinyBin of that maximum (squared form: kTinyBin^2) -- in particular a bin that cancels exactly in a transform of its own
// (full-scale square wave, pure tones on bin centres: the reference then has an exact 0 -> f32::EPSILON, functions.rs:66-71) --
// would come out with a visible relative error, so an oct in which any pair has such a bin is run again, each frame alone.
constexpr float kTinyBin = 1e-4f;



// Issues the loads of one oct: vin[e] = (a[n], b[n]), n = j + 16 e, of this lane group's frame pair (zero beyond flen,
// processing.rs:147-156).  Returns the pair's frame indices within their clips in tA / tB.
template <int NE>
__device__ __forceinline__ void load_oct(const Mfcc256Args &a, unsigned oct, unsigned total, int f, int j, float2 (&vin)[NE], unsigned &tA,
                                         unsigned &tB)
{
    const unsigned o8 = oct * 8;  // uniform
    const float *src[2], *xcl[2];
    unsigned tt[2];
    const bool pre = a.preemph != 0.f;  // fused pre-emphasis: the caller then loads at the top of the loop (no prefetch)
    const unsigned sh = a.preemph_shift % a.n_samples;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned fl = min(static_cast<unsigned>(2 * f + s), total - 1 - o8);  // lanes past the last frame redo it
        unsigned clip, t;
        if (a.nf_magic) {
            // scalar quotient of the oct's first frame (multiply-high by the host's reciprocal), one conditional wrap per lane
            clip = __umulhi(o8, a.nf_magic) >> a.nf_shift;
            t = o8 - clip * a.n_frames + fl;
            const bool wrap = t >= a.n_frames;
            t -= wrap ? a.n_frames : 0u;
            clip += wrap ? 1u : 0u;
        } else {
            const unsigned gf = o8 + fl;
            clip = gf / a.n_frames;
            t = gf - clip * a.n_frames;
        }
        // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
        xcl[s] = a.x + static_cast<unsigned long long>(clip) * a.ld;
        src[s] = xcl[s] + static_cast<unsigned long long>(t) * a.step + j;
        tt[s] = t;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const bool in = j + 16 * e < static_cast<int>(a.flen);
        vin[e] = in ? make_float2(src[0][16 * e], src[1][16 * e]) : make_float2(0.f, 0.f);
        if (pre && in) {
            vin[e].x = fmaf(-a.preemph, preemph_tap(xcl[0], static_cast<int>(tt[0] * a.step) + j + 16 * e, sh, a.n_samples), vin[e].x);
            vin[e].y = fmaf(-a.preemph, preemph_tap(xcl[1], static_cast<int>(tt[1] * a.step) + j + 16 * e, sh, a.n_samples), vin[e].y);
        }
    }
    tA = tt[0];
    tB = tt[1];
}

template <int NE, bool POW2, bool MFE, bool WIN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ss_mfcc_c256x2(const Mfcc256Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;  // frame pair within the oct
    const int j = lane & 15;  // lane within the pair (DPP row)

    float *slot = reinterpret_cast<float *>(smem) + wave * kWaveFloatsX + f * kSlotFloats;
    float2 *zh = reinterpret_cast<float2 *>(slot);
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kWaveFloatsX;
    const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2);
    const float *s_cos = s_tab + L::kCos;
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_melw = s_tab + L::kMelW;
    const float *s_win = s_tab + L::kMelW + 16 * a.mel_wpitch;  // WIN: 256 window samples (zero beyond flen)
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 16 * a.mel_wpitch + (WIN ? 256 : 0));

    const unsigned total = a.batch * a.n_frames;
    const unsigned octs = (total + 7) / 8;
    const unsigned o_lo = static_cast<unsigned>(static_cast<unsigned long long>(octs) * blockIdx.x / gridDim.x);
    const unsigned o_hi = static_cast<unsigned>(static_cast<unsigned long long>(octs) * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 16 * a.mel_wpitch + (WIN ? 256 : 0)) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = o_lo + WAVES;
    }
    unsigned oct = __builtin_amdgcn_readfirstlane(o_lo + wave);  // uniform: kept scalar
    float2 vin[NE];
    unsigned tA_next = 0, tB_next = 0;
    const bool pre = a.preemph != 0.f;  // pre-emphasised samples are formed at load time: no prefetch across the iteration then
    if (!pre && oct < o_hi) load_oct<NE>(a, oct, total, f, j, vin, tA_next, tB_next);

    const int paddr = ((lane & 48) | ((16 - j) & 15)) << 2;  // lane holding Z[256 - k]
    const int wbase1 = 34 * (j >> 1) + (j & 1);              // exchange write base (float2 units)
    const int Cc = static_cast<int>(a.n_ceps), M = static_cast<int>(a.n_filters);
    // |2A| = |s|: the 1/2 of the untangle is folded into the scale (1/4 for the squared form)
    const float hscale32 = (POW2 ? 0.25f * a.scale : 0.5f * a.scale) * kTwo32;
    __syncthreads();
    int st[3], fi[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        st[s] = s_start[s * 16 + j];
        fi[s] = s_filt[s * 16 + j];
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + j * a.mel_wpitch);
    const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + j * 52);
    constexpr bool RES_TW = NE <= 10;  // the 15 pass-2 twiddles stay in registers when the prefetch leaves room for them
    float4 tw2r[RES_TW ? 8 : 1];
    if (RES_TW) {
#pragma unroll
        for (int p = 0; p < 8; ++p) tw2r[p] = s_tw2[p * 16 + j];
    }
    float wn[WIN ? NE : 1];
    if (WIN) {
#pragma unroll
        for (int e = 0; e < NE; ++e) wn[e] = s_win[j + 16 * e];
    }

    constexpr unsigned kNone = 0xffffffffu;
    unsigned held = kNone;  // an oct this wave has claimed but postponed, because the current one is run a second time
    bool alone = false;     // run the current oct one frame per transform (set by the tiny-bin check of its first run)
    while (oct < o_hi) {
        unsigned next = 0;
        if (held != kNone) {
            next = held;
            held = kNone;
        } else {
            if (lane == 0) next = atomicAdd(s_next, 1u);
            next = __builtin_amdgcn_readfirstlane(next);
        }
        if (pre) load_oct<NE>(a, oct, total, f, j, vin, tA_next, tB_next);
        const unsigned tA = tA_next, tB = tB_next;

        // Guard of the two-for-one transform: its rounding error is relative to the LOUDER frame of a pair, so a frame of
        // (near) silence next to a loud one would come out with the loud one's noise floor instead of its own spectrum --
        // and an all-zero frame with a non-zero one instead of the exact zeros (-> f32::EPSILON, functions.rs:66-71) the
        // reference produces.  When the energies of a pair differ by more than 30 dB the wave runs the oct in two passes,
        // a + i 0 and then 0 + i b, each frame alone in its transform (error relative to itself, exact zeros stay exact).
        float ea = 0.f, eb = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            ea = fmaf(vin[e].x, vin[e].x, ea);
            eb = fmaf(vin[e].y, vin[e].y, eb);
        }
        ea = row16_sum(ea);
        eb = row16_sum(eb);
        const int npass = (alone || __any(fmaxf(ea, eb) > kPairGuard * fminf(ea, eb))) ? 2 : 1;
        alone = false;
        bool redo = false;
        for (int pass = 0; pass < npass; ++pass) {
        float2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float2 s = e < NE ? vin[e] : make_float2(0.f, 0.f);
            if (npass == 2) s = pass == 0 ? make_float2(s.x, 0.f) : make_float2(0.f, s.y);
            if (WIN && e < NE) s = make_float2(s.x * wn[e], s.y * wn[e]);
            v[e] = s;
        }
        // ---- 256-point complex FFT of a + i b: radix-16, transpose through LDS, twiddle, radix-16 ----
        fft16_reg(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) zh[wbase1 + 2 * r] = v[r];
        wave_order();
        if (!pre && pass == npass - 1 && next < o_hi) load_oct<NE>(a, next, total, f, j, vin, tA_next, tB_next);  // the input registers are dead now
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
            const float4 w2 = RES_TW ? tw2r[RES_TW ? p : 0] : s_tw2[p * 16 + j];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 7) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
        }
        fft16_reg(u);  // u[r] = Z[j + 16 r]

        // ---- split Z into the two frames' spectra; |X| (processing.rs:168) * 1/N (:180); row sums (feature.rs:216) ----
        float2 zcs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) zcs[r] = make_float2(bperm(paddr, u[15 - r].x), bperm(paddr, u[15 - r].y));
        float *prA = slot, *prB = slot + kPRowX;
        float esA = 0.f, esB = 0.f;
        float pmin = 3.0e38f, pmax = 0.f;  // smallest and largest bin of the pair (both frames)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float2 zk = u[r];
            // lane 0 pairs with itself: Z[256 - 16 r] = own register (16 - r) & 15
            const float2 zc = j == 0 ? u[(16 - r) & 15] : zcs[r];
            const float sx = zk.x + zc.x, sy = zk.y - zc.y;  // 2 A[k]
            const float dx = zk.x - zc.x, dy = zk.y + zc.y;  // 2 i B[k]
            const float na = fmaf(sx, sx, sy * sy), nb = fmaf(dx, dx, dy * dy);
            const float pa = POW2 ? na : __builtin_amdgcn_sqrtf(na);  // unscaled; hscale is applied to the sums below
            const float pb = POW2 ? nb : __builtin_amdgcn_sqrtf(nb);
            prA[j + 16 * r] = pa;
            prB[j + 16 * r] = pb;
            esA += pa;
            esB += pb;
            pmin = fminf(pmin, fminf(pa, pb));
            pmax = fmaxf(pmax, fmaxf(pa, pb));
        }
        if (j == 0) {
            // bin 128 pairs with itself: 2 A[128] = 2 Re Z[128], 2 B[128] = 2 Im Z[128]
            const float2 z = u[8];
            const float na = 4.f * z.x * z.x, nb = 4.f * z.y * z.y;
            const float pa = POW2 ? na : __builtin_amdgcn_sqrtf(na);
            const float pb = POW2 ? nb : __builtin_amdgcn_sqrtf(nb);
            prA[128] = pa;
            prB[128] = pb;
            esA += pa;
            esB += pb;
            pmin = fminf(pmin, fminf(pa, pb));
            pmax = fmaxf(pmax, fmaxf(pa, pb));
        }
        if (npass == 1) {
            // tiny-bin check over the 16 lanes of the pair (see kTinyBin); an all-zero pair has nothing to protect
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                pmin = fminf(pmin, __shfl_xor(pmin, m, 64));
                pmax = fmaxf(pmax, __shfl_xor(pmax, m, 64));
            }
            redo = __any(pmin < (POW2 ? kTinyBin * kTinyBin : kTinyBin) * pmax);
            if (redo) break;  // wave-uniform: nothing of this run is kept
        }
        if (j < 3) {  // pad bins read (with zero weight) by the mel stage
            prA[129 + j] = 0.f;
            prB[129 + j] = 0.f;
        }
        float en[2] = {hscale32 * row16_sum(esA), hscale32 * row16_sum(esB)};  // E * 2^32
#pragma unroll
        for (int s = 0; s < 2; ++s) en[s] = en[s] == 0.f ? kEps * kTwo32 : en[s];  // zero_handling, feature.rs:219
        wave_order();

        // ---- banded mel reduction (feature.rs:229), zero handling (:230), ln (:105) ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (npass == 2 && s != pass) continue;  // two-pass octs: this pass carried frame `pass` only
            const float *pr = slot + s * kPRowX;
            float *frow = slot + 2 * kPRowX + 48 * s;
            float m[3];
            int off = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                m[k] = hscale32 * mel_slot1(w4 + off, pr + st[k], a.mel_q4[k]);
                m[k] = m[k] == 0.f ? kEps * kTwo32 : m[k];
                off += a.mel_q4[k];
            }
            const unsigned gf = oct * 8 + 2 * f + s;
            if (MFE) {
                if (gf < total) {
                    float *row = a.out + static_cast<unsigned long long>(gf) * M;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (fi[k] >= 0) row[fi[k]] = m[k] * (1.0f / kTwo32);  // exact: power of two
                    if (j == 0) a.out_energy[gf] = en[s] * (1.0f / kTwo32);
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) frow[16 * k + j] = ln_scaled(m[k]);
            wave_order();
            // ---- DCT-II, first n_ceps coefficients (feature.rs:120-123): lane c against the 48-entry row ----
            float acc = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 lq[6], cq[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    lq[i] = *reinterpret_cast<const float4 *>(&frow[4 * (6 * h + i)]);
                    cq[i] = c4[6 * h + i];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    acc = fmaf(lq[i].x, cq[i].x, acc);
                    acc = fmaf(lq[i].y, cq[i].y, acc);
                    acc = fmaf(lq[i].z, cq[i].z, acc);
                    acc = fmaf(lq[i].w, cq[i].w, acc);
                }
            }
            // scaling + column-0 replacement (feature.rs:126-146)
            float o = acc * a.dct_scale_k;
            if (j == 0) o = a.dc_elimination ? ln_scaled(en[s]) : acc * ((s ? tB : tA) == 0 ? a.dct_scale_00 : a.dct_scale_0);
            if (j < Cc && gf < total) a.out[static_cast<unsigned long long>(gf) * Cc + j] = o;
            if (Cc > 16) {  // 17..32 cepstra: the lane also forms coefficient 16 + j (small batches: next to the prefetch registers)
                wave_order();
                const float4 *d4 = c4 + 16 * (52 / 4);
                float acc2 = 0.f;
#pragma unroll 1
                for (int h = 0; h < 6; ++h) {
                    float4 lq[2], dq[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        lq[i] = *reinterpret_cast<const float4 *>(&frow[4 * (2 * h + i)]);
                        dq[i] = d4[2 * h + i];
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        acc2 = fmaf(lq[i].x, dq[i].x, acc2);
                        acc2 = fmaf(lq[i].y, dq[i].y, acc2);
                        acc2 = fmaf(lq[i].z, dq[i].z, acc2);
                        acc2 = fmaf(lq[i].w, dq[i].w, acc2);
                    }
                }
                if (16 + j < Cc && gf < total) a.out[static_cast<unsigned long long>(gf) * Cc + 16 + j] = acc2 * a.dct_scale_k;
            }
        }
        wave_order();
        }
        if (redo) {
            // same oct again, one frame per transform; the samples that were prefetched for `next` are fetched again later
            wave_order();
            held = next;
            alone = true;
            if (!pre) load_oct<NE>(a, oct, total, f, j, vin, tA_next, tB_next);
            continue;
        }
        oct = next;
    }
}

template <int WAVES>
hipError_t launch_x(const Mfcc256Args &a_in, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    Mfcc256Args a = a_in;
    a.nf_magic = 0;
    a.nf_shift = 0;
    {
        // floor(x / d) for x < 2^31 as umulhi(x, ceil(2^(31+l) / d)) >> (l - 1), l = ceil(log2 d) (Granlund-Montgomery);
        // the kernel's one-wrap lane fix-up needs d >= 8
        const unsigned long long tot = static_cast<unsigned long long>(a.batch) * a.n_frames, d = a.n_frames;
        if (d >= 8 && d < (1ull << 31) && tot + 8 < (1ull << 31)) {
            unsigned l = 0;
            while ((1ull << l) < d) ++l;
            const unsigned __int128 num = static_cast<unsigned __int128>(1) << (31 + l);
            a.nf_magic = static_cast<uint32_t>((num + d - 1) / d);
            a.nf_shift = l - 1;
        }
    }
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloatsX + L::kMelW + 16 * static_cast<size_t>(a.mel_wpitch) + (a.windowed ? 256 : 0) + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    if (total + 8 >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned long long octs = (total + 7) / 8;
    unsigned long long blocks = (octs + WAVES - 1) / WAVES;
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
#define SS_X(NE, P, M, W, NAME) go(ss_mfcc_c256x2<NE, P, M, W, WAVES>, NAME)
#define SS_XN(NE, TAG)                                                                                                                          \
    if (a.out_mfe) {                                                                                                                            \
        if (pow2) return win ? SS_X(NE, true, true, true, "ss_mfcc_c256x2<" TAG ",pow2,mfe,win>") : SS_X(NE, true, true, false, "ss_mfcc_c256x2<" TAG ",pow2,mfe>"); \
        return win ? SS_X(NE, false, true, true, "ss_mfcc_c256x2<" TAG ",mfe,win>") : SS_X(NE, false, true, false, "ss_mfcc_c256x2<" TAG ",mfe>"); \
    }                                                                                                                                           \
    if (pow2) return win ? SS_X(NE, true, false, true, "ss_mfcc_c256x2<" TAG ",pow2,win>") : SS_X(NE, true, false, false, "ss_mfcc_c256x2<" TAG ",pow2>"); \
    return win ? SS_X(NE, false, false, true, "ss_mfcc_c256x2<" TAG ",win>") : SS_X(NE, false, false, false, "ss_mfcc_c256x2<" TAG ">");
    if (a.flen <= 160) { SS_XN(10, "10") }
    SS_XN(16, "16")
#undef SS_XN
#undef SS_X
}

}  // namespace

hipError_t launch_mfcc_c256x2(const Mfcc256Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    return launch_x<12>(a, stream, num_cus, info);
}

}  // namespace ss

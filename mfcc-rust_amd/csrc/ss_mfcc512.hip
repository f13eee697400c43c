// ss_mfcc_c256: the headline kernel -- fused MFCC for fft_points = 512 (C = 256 packed complex points)
// on gfx950 (MI355X), wave64.  Everything between the clip samples in HBM and the [frames x n_ceps]
// block in HBM happens in registers and wave-private LDS.
//
// Mapping
//   * Work unit: a QUAD of 4 consecutive frames; 16 lanes (one DPP row) own one frame, 16 complex points
//     per lane.  One persistent workgroup per CU; its waves pull quads from an LDS counter (dynamic
//     balance inside the CU, no global atomics), and the next quad's samples are prefetched.
//   * 256-point FFT = two radix-16 register butterflies with ONE transposing exchange through LDS.  The
//     exchange is private to a wave (LDS operations of one wave execute in order), so the main loop has
//     NO workgroup barrier.  Each frame has a 2304-B slot: ds_write_b64 scatter to
//     (n1,k1) -> 34*(n1>>1) + 2*k1 + (n1&1), read back with 8 ds_read_b128 per lane; conflict-free on
//     both sides (2304 B = 9 bank rows keeps the b128 lane groups of neighbouring frames apart).  Zero padding
//     is compile-time (template NE: a 320-sample frame fills 10 of the 16 first-pass inputs).
//   * The real-FFT untangle needs Z[256-k], which sits in lane 16-j, register 15-r: fetched with
//     ds_bpermute_b32 (LDS crossbar, no memory round trip).
//   * |X|/N for bins 0..128 goes to a P row in LDS (the mel bank ends at bin (F+1)/2, feature.rs:69-70);
//     all 257 bins feed the frame energy, reduced over the DPP row.
//   * mel: banded reduction -- each lane owns up to three filters (host-sorted by tap count so the
//     lock-step loops are short: 16 / 5 / 1 taps at the defaults, read as 16 / 6 / 2); weights are per-lane rows in LDS read as
//     ds_read_b128, all fetches of a stage issue back to back before the FMAs (LDS latency under load
//     is several hundred cycles).  Not MFMA: on gfx950 v_mfma_f32_* shares the FP32 datapath with
//     the VALU (measured: a VALU wave and an f32-MFMA wave on one SIMD take the SUM of their times),
//     so a block-dense product costs 8x the sparse one (tools/ubench/mfma_valu_overlap.hip; the retired
//     MFMA build of this kernel is tools/experiments/ss_mfcc512_mfma.hip).
//   * DCT-II: with 40 filters cos(pi c (2(39-m)+1)/80) = (-1)^c cos(pi c (2m+1)/80), so lane c < n_ceps multiplies 20 terms of
//     s[m] = L[m] + L[39-m] (even c) or d[m] = L[m] - L[39-m] (odd c) with its half cosine row held in registers.  The host lays
//     the (slot, lane) cells out so that a lane owns both filters of a pair (build_fast512, "paired", round 4): s and d are
//     formed in registers and go straight to the row the products read (template RES bits 2 + 3).  Other filter counts:
//     48-entry (slot, lane)-ordered ln(mel) row against the lane's cosine row in LDS (pitch 52 floats: conflict-free).
//   * Output stores are counted stores (ss_wave.h): unconditional buffer stores whose descriptor drops what must not be written,
//     so that the wait for the next quad's prefetched samples leaves them in flight (vmcnt(1) / (4) / (17) instead of vmcnt(0)).
//   * What bounds it (round 4, DESIGN.md 4 / 5, profiles/r04/): the launch runs at the board's 1400 W power cap (1320-1360 W on random
//     samples, shader clock 2.0-2.25 of 2.4 GHz; all-zero samples: 1.0 kW at 2.4 GHz and the 59 k cycles the kernel needs), so time is
//     energy per launch over the cap: 8 .. 12 waves per CU all take 30.1 +- 0.5 us while the clock falls from 2.37 to 2.01 GHz
//     (profiles/r04/power_probe.txt, ab_cfg2_waves_per_cu.txt), and what shortens a launch is fewer instructions per frame.  Round 4:
//     623 -> 601 executed VALU and 73 -> 65 LDS instructions per quad (paired DCT layout; tight mel taps: the host places the filters
//     of slots 1 / 2 inside 6 / 2 taps and only those are read; sample loads with a scalar base and a 32-bit lane offset -- no 64-bit
//     address arithmetic or quarter-rate multiplies on the VALU; no aggregation scaffold around the work counter's atomic).  A SIMD
//     issues one VALU instruction per 2.14 cycles when two of its waves have one ready, and the launch runs at 0.52-0.54 of that
//     floor (waves 20 % in s_waitcnt, 27 % stalled at issue, 11 % on the LDS queue; LDS array half busy, no bank conflicts).
//     Also written for instruction count: twiddle magnitudes folded into butterfly FMAs (ss_fft_reg.h), pass-2 twiddles in
//     registers (RES bit 1), ln on values pre-scaled by 2^32, conflict-free LDS accesses throughout.
//   * Builds (template OUTK / FRONT): MFCC; mfe's (features, energy); power_spectrum rows; each optionally with a
//     frame window and fused pre-emphasis on load.
//   * HBM traffic: samples once (the 50 % frame overlap is served by L1/L2), n_ceps floats per frame out.
// Compiled with -fno-slp-vectorize: v_pk_fma_f32 takes the issue slots of two scalar FMAs on gfx950
// (tools/ubench/valu_issue.hip: 5.1 against 2 x 2.5 cycles), so packing buys nothing and costs registers.
//
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

#include <algorithm>
#include <cstdlib>

// Timing-attribution builds (lab build only: tools/ablate.sh passes -DSS_LAB=1 -DSS_ABLATE=<bits>): each bit removes one
// stage; results are then wrong by design.  The product build compiles the switch out (SS_ABLATE is the constant 0 there,
// whatever the command line says).
//   1 partner fetch (ds_bpermute)   2 mel + ln + DCT   4 LDS exchange   8 square roots   16 sample loads in the loop
//   32 DCT only   64 second radix-16 pass
//   128 all sample loads from clip 0 (L2-resident: removes the HBM misses)
#if !SS_LAB
#undef SS_ABLATE
#endif
#ifndef SS_ABLATE
#define SS_ABLATE 0
#endif
// LDS stores, partner fetches and sample loads leave in small groups from inside the butterflies and the twiddle loop instead
// of as bursts behind them -- a wave issues in order, so a burst of 8..16 memory instructions holds back its own VALU work
// while the LDS / vector-memory queue drains (-1.0 us of 30.8 on one box against the round-1 bursts).

// SS_PROF2 (lab builds): every wave sums the shader-clock ticks it spends in each phase of its iterations (s_memtime at the
// phase boundaries, pinned by scheduling barriers; the wait for a stamp also drains the phase's LDS operations) and writes 16
// 64-bit words into the stamp buffer at the end: [0] iterations, [1..10] phase sums, [11] main-loop lifetime.  The buffer of
// ss_debug_stamp_buffer then needs 16 words per wave (tools/prof2.py); the ordinary stamps are off in such a build.
#if !SS_LAB
#undef SS_PROF2
#endif
#ifndef SS_PROF2
#define SS_PROF2 0
#endif
#if SS_PROF2
#define SS_PH(k)                                                     \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                           \
        pacc[k] += tn_ - tprev;                                      \
        tprev = tn_;                                                 \
    } while (0)
#else
#define SS_PH(k) do { } while (0)
#endif

namespace ss {

namespace {

using namespace wv;

constexpr int kZStride = 288;           // float2 per frame exchange slot (2304 B = 9 bank rows)
constexpr int kPRow = 144;              // floats per P row: bins 0..128, 3 zero pad bins, padding
constexpr int kPOff = 4 * kZStride * 2;  // P rows sit behind the exchange slots: their zero pad bins persist
constexpr int kWaveFloats = 4 * kZStride * 2 + 4 * kPRow;  // four exchange slots (log-mel rows reuse them) + four P rows
// Issue priorities of the phases (s_setprio; the reasoning is in ss_mel2048.hip): the two butterflies run at the lowest priority,
// so that a wave that is about to claim, exchange through LDS, fetch partners or read tables gets those requests out in front of
// the pure VALU streams of the waves beside it.  cfg2, same box (profiles/r03/ab_cfg2_cfg5_priorities*.txt): 30.8 us without
// priorities, 29.4 with 3 / 1 / 1 / 2 / 2; every assignment with the butterflies lowest is within 0.3 us of that.
// SS_PRIOS2 (lab builds): five decimal digits -- loop top (claim, samples), exchange reads + twiddles, untangle, mel, ln + DCT + store.
#if SS_LAB && defined(SS_PRIOS2)
#define SS_P2_TOP ((SS_PRIOS2 / 10000) % 10)
#define SS_P2_EX ((SS_PRIOS2 / 1000) % 10)
#define SS_P2_UN ((SS_PRIOS2 / 100) % 10)
#define SS_P2_MEL ((SS_PRIOS2 / 10) % 10)
#define SS_P2_DCT (SS_PRIOS2 % 10)
#define SS_P2_F1 ((SS_PRIOS2 / 1000000) % 10)
#define SS_P2_F2 ((SS_PRIOS2 / 100000) % 10)
#else
#define SS_P2_F1 0
#define SS_P2_F2 0
#define SS_P2_TOP 3
#define SS_P2_EX 1
#define SS_P2_UN 1
#define SS_P2_MEL 2
#define SS_P2_DCT 2
#endif
#if SS_LAB && defined(SS_NOPRIO2)
#define SS_PRIOL(x) do { } while (0)
#else
#define SS_PRIOL(x) __builtin_amdgcn_s_setprio(x)
#endif
namespace L = fast512_layout;


// Issues the loads of one quad; returns this lane's frame index within its clip.
template <int NE, bool EXACT, bool PRE, bool CENTER = false>
__device__ __forceinline__ unsigned load_quad(const Fast512Args &a, unsigned quad, unsigned total, int f, int j, float2 (&vin)[NE],
                                              float2 (&pin)[PRE ? NE : 1])
{
    const unsigned q4 = quad * 4;                                // uniform
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
    if (CENTER) {
        // librosa center=True: frame t is centred on sample t*step.  Frames inside the clip load like contract frames from
        // their (even) start; the few at the clip edges mirror (np.pad 'reflect') or zero their out-of-range samples.
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        const int s0 = static_cast<int>(t * a.step) - static_cast<int>(a.flen / 2), ns = static_cast<int>(a.n_samples);
        const bool inside = s0 >= 0 && s0 + static_cast<int>(a.flen) <= ns;
        if (__all(inside)) {
            const float2 *srcc = reinterpret_cast<const float2 *>(xc + s0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int n = j + 16 * e;
                // zero pad beyond flen; an odd frame length ends in a half pair
                const int rem = static_cast<int>(a.flen) - 2 * n;
                vin[e] = rem >= 2 ? srcc[n] : make_float2(rem == 1 ? reinterpret_cast<const float *>(srcc)[2 * n] : 0.f, 0.f);
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
                        if (ok) sv[h] = xc[pos];
                    }
                }
                vin[e] = make_float2(sv[0], sv[1]);
            }
        }
        return t;
    }
    // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
    if (SS_ABLATE & 128) clip = 0;
    const float2 *src = reinterpret_cast<const float2 *>(a.x + static_cast<unsigned long long>(clip) * a.ld + t * a.step);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int n = j + 16 * e;
        if (EXACT) {
            vin[e] = src[n];
        } else {
            // zero pad beyond flen (processing.rs:147-156); an odd frame length ends in a half pair
            const int rem = static_cast<int>(a.flen) - 2 * n;
            vin[e] = rem >= 2 ? src[n] : make_float2(rem == 1 ? reinterpret_cast<const float *>(src)[2 * n] : 0.f, 0.f);
        }
        if (PRE) {
            // pre-emphasis taps x[(i - shift) mod L] of the sample pair (processing.rs:31-53, np.roll semantics over the clip)
            const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
            const unsigned i0 = t * a.step + 2 * n, sh = a.preemph_shift % a.n_samples;
            const unsigned j0 = i0 >= sh ? i0 - sh : i0 + a.n_samples - sh;
            const unsigned j1 = i0 + 1 >= sh ? i0 + 1 - sh : i0 + 1 + a.n_samples - sh;
            if (EXACT || 2 * n < static_cast<int>(a.flen)) pin[e] = make_float2(xc[j0], xc[j1]);
            else pin[e] = make_float2(0.f, 0.f);
        }
    }
    return t;
}

// Contract framing only: where this lane's frame of `quad` starts (frame t of its clip begins at sample t * step), and t.
// `quad` is uniform: the clip / frame split of the quad's first frame, the clip's address and the frame's offset in it are scalar
// work, and what a lane adds is a 32-bit byte offset (its frame within the quad, its sample pair, one conditional step into the
// next clip) -- the loads take the SGPR-base + VGPR-offset form and no 64-bit address is formed on the VALU.
struct QuadSrc {
    const char *base;  // uniform
    unsigned off;      // this lane's byte offset
};
__device__ __forceinline__ QuadSrc quad_src(const Fast512Args &a, unsigned quad, unsigned total, int f, int j, unsigned &t_out)
{
    const unsigned q4 = quad * 4;
    const unsigned fl = min(static_cast<unsigned>(f), total - 1 - q4);  // lanes past the last frame redo it
    if (a.nf_magic) {
        const unsigned clip = __umulhi(q4, a.nf_magic) >> a.nf_shift;
        const unsigned t0 = q4 - clip * a.n_frames;
        const unsigned traw = t0 + fl;
        const bool wrap = traw >= a.n_frames;
        t_out = min(traw, traw - a.n_frames);  // (unsigned: the difference is huge unless the frame belongs to the next clip)
        // a step into the next clip: + ld samples, - n_frames * step of them
        const unsigned into_next = (static_cast<unsigned>(a.ld) - a.n_frames * a.step) * 4u;
        QuadSrc r;
        r.base = reinterpret_cast<const char *>(a.x + static_cast<unsigned long long>(clip) * a.ld + static_cast<unsigned long long>(t0) * a.step);
        r.off = __umul24(fl, a.step * 4u) + static_cast<unsigned>(j) * 8u + (wrap ? into_next : 0u);
        return r;
    }
    // clips of fewer than four frames (no reciprocal: a quad may span several clips): the lane's offset from the first clip's first
    // sample, which launch_w has checked to fit 32 bits for the whole batch
    const unsigned gf = q4 + fl;
    const unsigned clip = gf / a.n_frames;
    const unsigned t = gf - clip * a.n_frames;
    t_out = t;
    QuadSrc r;
    r.base = reinterpret_cast<const char *>(a.x);
    r.off = (clip * static_cast<unsigned>(a.ld) + t * a.step) * 4u + static_cast<unsigned>(j) * 8u;
    return r;
}

// One mel slot with a compile-time tap count (multiple of 4): weights and P taps are all requested before
// the first FMA.  `w4` = this lane's weight row at the slot's offset, `p` = P row at the slot's start bin.
template <int Q4>
__device__ __forceinline__ float mel_slot_fixed(const float4 *w4, const float *p)
{
    float4 w[Q4];
    float t[4 * Q4];
#pragma unroll
    for (int i = 0; i < Q4; ++i) w[i] = w4[i];
#pragma unroll
    for (int i = 0; i < 4 * Q4; ++i) t[i] = p[i];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < Q4; ++i) {
        acc = fmaf(w[i].x, t[4 * i], acc);
        acc = fmaf(w[i].y, t[4 * i + 1], acc);
        acc = fmaf(w[i].z, t[4 * i + 2], acc);
        acc = fmaf(w[i].w, t[4 * i + 3], acc);
    }
    return acc;
}

// One mel slot with a compile-time tap count that need not be a multiple of 4 (TIGHT builds): whole float4s of weights, T taps
template <int T>
__device__ __forceinline__ float mel_slot_taps(const float4 *w4, const float *p)
{
    constexpr int Q = (T + 3) / 4;
    float4 w[Q];
    float t[T];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        w[i] = w4[i];
        // (whole 16-byte reads: left alone, the compiler narrows the partly used float4s and merges two of them into one
        // ds_read2_b64, whose four words per lane collide between lanes at the rows' 28-float pitch -- 8 conflict cycles per quad)
        asm volatile("" : "+v"(w[i].x), "+v"(w[i].y), "+v"(w[i].z), "+v"(w[i].w));
    }
#pragma unroll
    for (int i = 0; i < T; ++i) t[i] = p[i];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const float wi = (i & 3) == 0 ? w[i / 4].x : (i & 3) == 1 ? w[i / 4].y : (i & 3) == 2 ? w[i / 4].z : w[i / 4].w;
        acc = fmaf(wi, t[i], acc);
    }
    return acc;
}

// Same with a run-time tap count (configurations other than the default bank shape): chunks of 4, 2 and 1 float4 of
// weights, so that a slot costs a few LDS round trips instead of one per four taps.
__device__ __forceinline__ float mel_slot_loop(const float4 *w4, const float *p, int q4)
{
    float acc = 0.f;
    int i = 0;
    for (; i + 4 <= q4; i += 4) acc += mel_slot_fixed<4>(w4 + i, p + 4 * i);
    if (i + 2 <= q4) {
        acc += mel_slot_fixed<2>(w4 + i, p + 4 * i);
        i += 2;
    }
    if (i < q4) acc += mel_slot_fixed<1>(w4 + i, p + 4 * i);
    return acc;
}

// MULTI builds (ss_mfcc_batches_device): the launch's quad range is the concatenation of up to kMaxLaunchBatches independent batches'
// quad ranges, each batch with its own input and output block (BatchTable, ss_device.h; Seg / seg_of, ss_wave.h).
template <int NE, bool EXACT, bool POW2, int WAVES, bool BANK421, int NQ, int RES = 0, int OUTK = 0, int FRONT = 0, bool FULLP = false,
          bool CENTER = false, bool MULTI = false>
__global__ __launch_bounds__(WAVES * 64) void ss_mfcc_c256(const Fast512Args a_in, const MultiArg<MULTI> mt)
{
    // Everything in front of a wave's first sample loads is start-up latency of the launch (nothing can be computed before
    // the samples are here), so the kernel arguments that lead to those loads are fetched by ONE batch of scalar loads at the
    // very top (pinned: left alone, the compiler fetches them where they are first used -- three dependent scalar-memory
    // round trips, the last one behind the table waves' vector loads).
    Fast512Args a = a_in;
    // (the pointers themselves are not pinned: behind an asm statement they would no longer be known to point to global
    // memory and their loads would become flat_load; they sit in the same kernarg lines as the pinned scalars)
    asm volatile("" : "+s"(a.ld));
    asm volatile("" : "+s"(a.n_samples), "+s"(a.batch), "+s"(a.flen), "+s"(a.step), "+s"(a.n_frames), "+s"(a.mel_wpitch), "+s"(a.win_floats));
    asm volatile("" : "+s"(a.nf_magic), "+s"(a.nf_shift), "+s"(a.q_base), "+s"(a.q_rem));
    constexpr bool PREFETCH = true;
    // more than 12 waves per CU (4 per SIMD, <= 128 VGPRs): the P / ln / sum-difference rows live inside the frame's own
    // exchange slot (9216 B per wave) and no table stays in registers
    constexpr bool ALIAS = WAVES > 12;
    constexpr bool TABREG = WAVES <= 12;
    constexpr int WF = ALIAS ? 4 * kZStride * 2 : kWaveFloats;
    constexpr bool MFE = OUTK == 1, PWR = OUTK == 2;  // output: 0 MFCC, 1 mfe's (features, energy), 2 the power_spectrum rows
    constexpr bool WIN = (FRONT & 1) != 0, PRE = (FRONT & 2) != 0;  // optional frame window / fused pre-emphasis  // a 4-waves-per-SIMD build has no registers for the prefetch / resident twiddles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    // diagnostic stamps (tools/dbg_times.py) go to memory where they are taken: kept in registers until the end they cost the
    // production path two spilled VGPRs (1.5 MB of scratch writes per cfg2 launch)
    auto stamp = [&](int k, unsigned long long v) {
        if (!SS_PROF2 && a.dbg && lane == 0) a.dbg[6ull * (blockIdx.x * WAVES + wave) + k] = v;
    };
    if (!SS_PROF2 && a.dbg) {
        stamp(0, __builtin_amdgcn_s_memrealtime());
        stamp(5, __builtin_amdgcn_s_memtime());  // shader-clock ticks: replaced by the cycles lived at the end
        stamp(4, 0ull);
    }
    const int f = lane >> 4;  // frame within the quad
    const int j = lane & 15;  // column of the frame's 16 x 16 point matrix owned by this lane of the DPP row

    // ---- LDS carve: per-wave regions, then the shared read-only table block, then the quad counter ----
    float *wbase = reinterpret_cast<float *>(smem) + wave * WF;
    float2 *zh = reinterpret_cast<float2 *>(wbase) + f * kZStride;        // this frame's exchange slot
    // P row: bins 0..128 + zero pad bins behind the exchange slots; FULLP: all 257 bins (+ pad) inside the frame's own
    // exchange slot, with the ln(mel) row behind it
    float *prow = (FULLP || ALIAS) ? wbase + f * (2 * kZStride) : wbase + kPOff + f * kPRow;
    float *frow = FULLP ? prow + 264 : ALIAS ? prow + 144 : wbase + f * 48;  // ln(mel) in (slot, lane) order (after the exchange)
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * WF;
    const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const float *s_cos = s_tab + L::kCos;
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const float *s_melw = s_tab + L::kMelW;
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kMelW + 16 * a.mel_wpitch);  // WIN: window sample pairs
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 16 * a.mel_wpitch + (WIN ? a.win_floats : 0));

    // quad range of this workgroup (contiguous, balanced to within one quad)
    const unsigned total = a.batch * a.n_frames;
    const unsigned q_lo = blockIdx.x * a.q_base + min(blockIdx.x, a.q_rem);
    const unsigned q_hi = q_lo + a.q_base + (blockIdx.x < a.q_rem ? 1u : 0u);

    // The table block (global layout == LDS layout) is fetched by the workgroup's first two waves only, before they ask
    // for their samples: behind the sample loads of the other waves (HBM misses that fill the CU's miss queue) a table load
    // takes 3-4 us instead of 1, and the barrier below waits for the slowest one.
    constexpr int kTabWaves = 2;
    const int n4 = (L::kMelW + 16 * a.mel_wpitch + (WIN ? a.win_floats : 0)) / 4;
    constexpr int kBatch = 6;  // loads in flight per lane (768 float4s cover every 512-point table block with <= 80 floats per mel row)
    float4 tv[kBatch];
    if (wave < kTabWaves) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) tv[k] = reinterpret_cast<const float4 *>(a.tab)[min(tid + k * (kTabWaves * 64), n4 - 1)];
    }
    // first quad of this wave; its loads are in flight across the barrier (a wave without a quad loads the block's last
    // one: no branch around the loads).  The table waves ask for their samples right behind their table loads, before they
    // wait for the tables: vector loads return in order, so the tables still come first.
    unsigned quad = __builtin_amdgcn_readfirstlane(q_lo + wave);  // uniform: kept scalar
    float2 vin[NE];
    float2 pin[PRE ? NE : 1];
    unsigned t_next = 0;  // frame index within the clip of the quad whose samples are in vin
    Seg ns{};                // MULTI: the batch of the quad whose samples are being fetched ...
    Fast512Args an = a_in;   // ... and the argument block with that batch's input
    if constexpr (MULTI) {
        an = a;
        ns = seg_of(mt.m, min(quad, q_hi - 1));
        an.x = ns.x;
        t_next = load_quad<NE, EXACT, PRE, CENTER>(an, min(quad, q_hi - 1) - ns.u0, ns.total, f, j, vin, pin);
    } else {
        t_next = load_quad<NE, EXACT, PRE, CENTER>(a, min(quad, q_hi - 1), total, f, j, vin, pin);
    }
    if (wave < kTabWaves) {
        // (pinned: the compiler otherwise sinks each load next to its store, one memory round trip per float4)
#pragma unroll
        for (int k = 0; k < kBatch; ++k) asm volatile("" : "+v"(tv[k].x), "+v"(tv[k].y), "+v"(tv[k].z), "+v"(tv[k].w));
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int idx = tid + k * (kTabWaves * 64);
            if (idx < n4) reinterpret_cast<float4 *>(s_tab)[idx] = tv[k];
        }
        // table blocks of more than 768 float4s (mel rows wider than 80 floats): the rest, behind the samples
        for (int base = kBatch * kTabWaves * 64; base < n4; base += kBatch * kTabWaves * 64) {
#pragma unroll
            for (int k = 0; k < kBatch; ++k) tv[k] = reinterpret_cast<const float4 *>(a.tab)[min(base + tid + k * (kTabWaves * 64), n4 - 1)];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) asm volatile("" : "+v"(tv[k].x), "+v"(tv[k].y), "+v"(tv[k].z), "+v"(tv[k].w));
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int idx = base + tid + k * (kTabWaves * 64);
                if (idx < n4) reinterpret_cast<float4 *>(s_tab)[idx] = tv[k];
            }
        }
        // table waves: when the tables were in LDS, as 100 MHz ticks since this wave's start stamp, flagged 2 in bits 40..41 (the
        // other waves flag their cycle count 1 there; an absolute stamp would carry either flag by accident, depending on uptime)
        if (!SS_PROF2 && a.dbg && lane == 0) {
            unsigned long long *d = a.dbg + 6ull * (blockIdx.x * WAVES + wave);
            d[5] = ((__builtin_amdgcn_s_memrealtime() - d[0]) & ((1ull << 40) - 1)) | (2ull << 40);
        }
    }
    {
        if (tid == 0) *s_next = q_lo + WAVES;
        if (!FULLP && !ALIAS && j < 3) prow[129 + j] = 0.f;  // pad bins read (with zero weight) by the mel stage; never written again
    }

    const int paddr = ((lane & 48) | ((16 - j) & 15)) << 2;  // lane holding Z[256 - k]
    const int wbase1 = 34 * (j >> 1) + (j & 1);              // exchange write base (float2 units)
    const int Cc = static_cast<int>(a.n_ceps);
    // |X| = (1/2)|...|: the 1/2 of the untangle is folded into the scale (1/4 for the squared form)
    const float hscale32 = (POW2 ? 0.25f * a.scale : 0.5f * a.scale) * kTwo32;
    wg_barrier_lds();  // the tables are in LDS; the wave's first samples stay in flight across it
    // first P bin of this lane's three filters, as float indices into LDS; opaque so that the full address stays in a
    // register (the compiler otherwise re-adds the P-row offset in front of every ds_read2)
    const int pbase = wave * WF + ((FULLP || ALIAS) ? f * (2 * kZStride) : kPOff + f * kPRow);
    int st0 = pbase + s_start[j];
    int st1 = pbase + s_start[16 + j];
    int st2 = pbase + s_start[32 + j];
    asm volatile("" : "+v"(st0), "+v"(st1), "+v"(st2));
    const float *smem_f = reinterpret_cast<const float *>(smem);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    constexpr bool SYM = (RES & 4) != 0;  // 40 filters: DCT against sum/difference rows with the half cosine row in registers
    constexpr bool PAIRED = (RES & 8) != 0;  // SYM with the host's paired cell layout: s and d are formed in registers
    constexpr bool TIGHT = (RES & 16) != 0;  // PAIRED with the host's tight tap placement: slots 1 / 2 read 6 / 2 taps
    const int fidx0 = (MFE || SYM) ? s_filt[j] : 0, fidx1 = (MFE || SYM) ? s_filt[16 + j] : 0, fidx2 = (MFE || SYM) ? s_filt[32 + j] : 0;
    float4 ch[SYM ? 5 : 1];
    if (SYM && TABREG) {
        const float4 *h4 = reinterpret_cast<const float4 *>(s_tab + L::kCosH + j * 20);
#pragma unroll
        for (int i = 0; i < 5; ++i) ch[i] = h4[i];
    }
    // SYM: where this lane's three ln(mel) values go in the natural-order row (slots without a filter go to the pad entries)
    float *fr0 = frow + (fidx0 >= 0 ? fidx0 : 47), *fr1 = frow + (fidx1 >= 0 ? fidx1 : 47), *fr2 = frow + (fidx2 >= 0 ? fidx2 : 47);
    // s[20] = L[m] + L[39-m], d[20] = L[m] - L[39-m] of this frame (PAIRED: rows 48 floats apart -- room for the writes' dump
    // words, and the four rows' ds_read_b128 stay in different banks)
    float *sd = ALIAS ? prow + 192 : wbase + 192 + f * (PAIRED ? 48 : 40);
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + j * a.mel_wpitch);
    const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + j * 52);
    float2 twn[8];  // exp(-2 pi i (j + 16 r) / 512): resident when the register budget allows (<= 3 waves per SIMD)
    if (TABREG) {
#pragma unroll
        for (int r = 0; r < 8; ++r) twn[r] = s_twn[r * 16 + j];
    }
    // RES bit 0: this lane's cosine row stays in registers; bit 1: the 15 pass-2 twiddles do
    float4 cr[(RES & 1) ? NQ : 1];
    float4 tw2r[(RES & 2) ? 8 : 1];
    if (RES & 1) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) cr[i] = c4[i];
    }
    if (RES & 2) {
#pragma unroll
        for (int p = 0; p < 8; ++p) tw2r[p] = s_tw2[p * 16 + j];
    }
    // column scale of this lane (feature.rs:126-146): column 0 has its own (and none when the frame energy replaces it)
    const float sc_lane = j == 0 ? (a.dc_elimination ? 0.f : a.dct_scale_0) : a.dct_scale_k;
    if (!SS_PROF2 && a.dbg) stamp(1, __builtin_amdgcn_s_memrealtime());
#if SS_PROF2
    unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif
    unsigned n_done = 0;
    // The loop waits for a quad's prefetched samples with its predecessor's output store still in flight (vmcnt(1): the store is
    // the one younger operation).  The first quad's samples come from the prologue, with no store behind them -- and where the
    // two paths meet the compiler would have to assume the worse one, vmcnt(0), for every iteration.  One store that the range
    // check drops gives the prologue path the same shape (as many as the build stores per quad: 1 MFCC, 4 mfe, 17 power rows).
    {
        constexpr int kLoopStores = OUTK == 0 ? 1 : (OUTK == 1 ? 4 : 17);
        const __amdgpu_buffer_rsrc_t none = out_rsrc(a.out, 0u);
#pragma unroll
        for (int k = 0; k < kLoopStores; ++k) buf_store(0.f, none, 64 * k);  // (distinct, non-adjacent addresses: identical or adjacent stores would be merged)
    }

    SS_PRIOL(SS_P2_TOP);
    while (quad < q_hi) {
        const Seg cs = ns;  // MULTI: the batch of this iteration's quad (its samples were fetched from it)
        (void)cs;
        // claim the next quad now so that its samples can be prefetched during this one
        // (the claim is issued here and read behind the first butterfly, where the prefetch needs it: read at once, the LDS
        // atomic's round trip was exposed at the top of every iteration)
        unsigned next_v = 0;
        if (lane == 0) next_v = atomicAdd(s_next, 1u);
        ++n_done;

        SS_PH(1);  // claim
        if (!PREFETCH && n_done > 1) t_next = load_quad<NE, EXACT, PRE, CENTER>(a, quad, total, f, j, vin, pin);
        const unsigned t_cur = t_next;
#if SS_PROF2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SS_PH(2);  // the prefetched samples are here
        SS_PRIOL(SS_P2_F1);
        float2 v[16];
#if SS_LAB
        if (!SS_PROF2 && a.dbg && n_done == 1) {
            // diagnostic runs (lab builds: a conditional store inside the loop costs the product path its counted waits): when
            // this wave's first samples have arrived
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(4, __builtin_amdgcn_s_memrealtime());
        }
#endif
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float2 s = e < NE ? vin[e] : make_float2(0.f, 0.f);  // zero pad, processing.rs:147-156
            if (PRE && e < NE) s = make_float2(fmaf(-a.preemph, pin[e].x, s.x), fmaf(-a.preemph, pin[e].y, s.y));
            if (WIN && e < NE) {
                const float2 w = s_win[j + 16 * e];
                s = make_float2(s.x * w.x, s.y * w.y);
            }
            v[e] = s;
        }

        // ---- 256-point complex FFT: radix-16, transpose through LDS, twiddle, radix-16 ----
        if (!(SS_ABLATE & 4)) {
            // the exchange stores leave group by group while the butterfly is still computing (no 16-store burst into the LDS queue)
            fft16_emit<(NE >= 8 ? NE : 16)>(
                v, [&](int r, float2 val) { zh[wbase1 + 2 * r] = val; }, [] { __builtin_amdgcn_sched_barrier(0); });
        } else {
            fft16_reg(v);
        }
        wave_order();
        const unsigned next = __builtin_amdgcn_readfirstlane(next_v);
        SS_PRIOL(SS_P2_EX);
        SS_PH(3);  // pass 1 + exchange stores
        // the input registers are dead now: the next quad's samples load into them (no copies), three quarters of an
        // iteration ahead of their use
        // SPREAD: the ten sample loads of the next quad go out one per twiddle step instead of as a burst (a wave issues in
        // order: behind a burst of vector-memory instructions its own VALU work waits); no branch surrounds them -- the last
        // iteration of a wave fetches the block's last quad again and drops it
        constexpr bool SPREAD = PREFETCH && EXACT && !PRE && !CENTER && !(SS_ABLATE & 16);
        static_assert(!MULTI || (SPREAD && OUTK == 0), "the multi-batch build exists for the spread-prefetch MFCC builds");
        QuadSrc nsrc{nullptr, 0u};
        if constexpr (MULTI) {
            const unsigned nq = min(next, q_hi - 1);
            if (nq >= ns.u1) {  // (uniform, rare: the claimed quad starts the next batch)
                ns = seg_of(mt.m, nq);
                an.x = ns.x;
            }
            nsrc = quad_src(an, nq - ns.u0, ns.total, f, j, t_next);
        } else {
            if (SPREAD) nsrc = quad_src(a, min(next, q_hi - 1), total, f, j, t_next);
        }
        if (!SPREAD && PREFETCH && next < q_hi && !(SS_ABLATE & 16)) t_next = load_quad<NE, EXACT, PRE, CENTER>(a, next, total, f, j, vin, pin);
        float2 u[16];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (SS_ABLATE & 4) {
                u[2 * p] = v[2 * p];
                u[2 * p + 1] = v[2 * p + 1];
                continue;
            }
            const float4 t4 = *reinterpret_cast<const float4 *>(&zh[34 * p + 2 * j]);
            u[2 * p] = make_float2(t4.x, t4.y);
            u[2 * p + 1] = make_float2(t4.z, t4.w);
        }
        wave_order();
#pragma unroll
        for (int p = 0; p < 8; ++p) {  // two twiddles per ds_read_b128
            const float4 w2 = (RES & 2) ? tw2r[p] : s_tw2[p * 16 + j];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 7) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
            if (SPREAD) {
                if (p < NE) vin[p] = *reinterpret_cast<const float2 *>(nsrc.base + nsrc.off + 128 * p);
                if (p == 7) {
#pragma unroll
                    for (int e = 8; e < NE; ++e) vin[e] = *reinterpret_cast<const float2 *>(nsrc.base + nsrc.off + 128 * e);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        SS_PRIOL(SS_P2_F2);
        SS_PH(4);  // exchange reads + twiddles (+ prefetch issue)
        // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
        // the partner of bin j + 16 r is register 15 - r of lane 16 - j: fetched with ds_bpermute
        float2 zcs[8];
        if (!(SS_ABLATE & 65)) {
            // the fetches of the upper registers go out as soon as the butterfly has produced them, group by group
            float2 uo[16];  // (the butterfly still reads its input registers while the first groups' outputs appear)
            fft16_emit(
                u,
                [&](int k, float2 val) {
                    uo[k] = val;
                    if (k >= 8) zcs[15 - k] = make_float2(bperm(paddr, val.x), bperm(paddr, val.y));
                },
                [] { __builtin_amdgcn_sched_barrier(0); });
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k] = uo[k];
        } else {  // stage-removal builds only
            if (!(SS_ABLATE & 64)) fft16_reg(u);  // u[r] = Z[j + 16 r]
#pragma unroll
            for (int r = 0; r < 8; ++r) zcs[r] = (SS_ABLATE & 1) ? u[15 - r] : make_float2(bperm(paddr, u[15 - r].x), bperm(paddr, u[15 - r].y));
        }
        SS_PRIOL(SS_P2_UN);
        SS_PH(5);  // pass 2 + partner fetches
        float esum = 0.f;
        // power_spectrum output (processing.rs:179-181): the scaled |X| of all 257 bins of the frame, 64 contiguous bytes
        // per register and frame on either side of the spectrum
        // (counted stores, ss_wave.h: the descriptor covers the quad's valid frames)
        const float hs_pw = hscale32 * (1.0f / kTwo32);
        __amdgpu_buffer_rsrc_t pw_rsrc = out_rsrc(nullptr, 0u);
        if (PWR) {
            const unsigned quad_s = __builtin_amdgcn_readfirstlane(quad);
            pw_rsrc = out_rsrc(a.out + static_cast<unsigned long long>(quad_s) * 4ull * 257ull, min(4u, total - quad_s * 4) * 257u * 4u);
        }
        const int pw_off = f * 257 * 4;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float2 zk = u[r];
            // lane 0 pairs with itself: Z[256 - 16 r] = own register (16 - r) & 15
            const float2 zc = j == 0 ? u[(16 - r) & 15] : zcs[r];
            const float2 w = TABREG ? twn[r] : s_twn[r * 16 + j];
            const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
            const float2 d = make_float2(zk.x - zc.x, zk.y + zc.y);
            // 2 X[k] = s - i w d, 2 conj X[256-k] = s + i w d = 2 s - 2 X[k]: six FMAs instead of a product and four adds
            const float xa_r = fmaf(w.y, d.x, fmaf(w.x, d.y, s.x));
            const float xa_i = fmaf(w.y, d.y, fmaf(-w.x, d.x, s.y));
            const float xb_r = fmaf(2.f, s.x, -xa_r), xb_i = fmaf(2.f, s.y, -xa_i);
            const float na = xa_r * xa_r + xa_i * xa_i, nb = xb_r * xb_r + xb_i * xb_i;
            const float pa = (POW2 || (SS_ABLATE & 8)) ? na : __builtin_amdgcn_sqrtf(na);  // unscaled; hscale is applied to the sums below
            const float pb = (POW2 || (SS_ABLATE & 8)) ? nb : __builtin_amdgcn_sqrtf(nb);
            if (PWR) {
                buf_store(hs_pw * pa, pw_rsrc, pw_off + (j + 16 * r) * 4);
                buf_store(hs_pw * pb, pw_rsrc, pw_off + (256 - j - 16 * r) * 4);
                continue;
            }
            prow[j + 16 * r] = pa;  // only bins <= 128 can carry mel weight (the bank ends at (F+1)/2, feature.rs:69-70) ...
            if (FULLP) prow[256 - j - 16 * r] = pb;  // ... unless the bank covers the whole spectrum
            esum += pa + pb;
        }
        if (j == 0) {
            // lane 0's pair k = 0 produced X[0] and X[256]; X[128] = conj Z[128] is the one extra bin
            const float2 z = u[8];
            const float n = 4.f * (z.x * z.x + z.y * z.y);
            const float p128 = POW2 ? n : __builtin_amdgcn_sqrtf(n);
            if (!PWR) {
                prow[128] = p128;
                esum += p128;
            }
        }
        if (PWR) {
            const float2 z = u[8];  // (every lane computes it; lane 0's is X[128])
            buf_store(hs_pw * (POW2 ? 4.f * (z.x * z.x + z.y * z.y) : __builtin_amdgcn_sqrtf(4.f * (z.x * z.x + z.y * z.y))), pw_rsrc,
                      j == 0 ? pw_off + 128 * 4 : kOobOffset);
        }
        if ((FULLP || ALIAS) && j < 3) prow[(FULLP ? 257 : 129) + j] = 0.f;  // pad bins (the slot was overwritten by the exchange)
        if (PWR) {
            wave_order();
            quad = next;
            continue;
        }
        float energy = hscale32 * row16_sum(esum);      // E * 2^32
        energy = energy == 0.f ? kEps * kTwo32 : energy;  // zero_handling, feature.rs:219
        wave_order();
        SS_PRIOL(SS_P2_MEL);
        SS_PH(6);  // untangle + magnitudes + energy

        // ---- banded mel reduction (feature.rs:229), zero handling (:230), ln (:105) ----
        float m0, m1, m2;
        if (SS_ABLATE & 2) {
            const unsigned gf = quad * 4 + f;
            if (j < Cc && gf < total) a.out[static_cast<unsigned long long>(gf) * Cc + j] = energy;
            wave_order();
            quad = next;
            continue;
        }
        if (BANK421) {
            m0 = mel_slot_fixed<4>(w4, smem_f + st0);
            // TIGHT: the host placed every filter of slots 1 / 2 inside the first 6 / 2 taps (the dropped products are x * 0)
            m1 = TIGHT ? mel_slot_taps<6>(w4 + 4, smem_f + st1) : mel_slot_fixed<2>(w4 + 4, smem_f + st1);
            m2 = TIGHT ? mel_slot_taps<2>(w4 + 6, smem_f + st2) : mel_slot_fixed<1>(w4 + 6, smem_f + st2);
        } else {
            m0 = mel_slot_loop(w4, smem_f + st0, a.mel_q4[0]);
            m1 = mel_slot_loop(w4 + a.mel_q4[0], smem_f + st1, a.mel_q4[1]);
            m2 = mel_slot_loop(w4 + a.mel_q4[0] + a.mel_q4[1], smem_f + st2, a.mel_q4[2]);
        }
        if (MFE) {
            // mfe (feature.rs:200-233): the mel energies and the frame energy themselves, in filter order
            const float hs = hscale32 * (1.0f / kTwo32);
            {
                // counted stores (ss_wave.h): descriptors over the quad's valid frames; slots without a filter are dropped
                const unsigned quad_s = __builtin_amdgcn_readfirstlane(quad);
                const unsigned nvalid = min(4u, total - quad_s * 4);
                const unsigned nfl = a.n_filters;
                const __amdgpu_buffer_rsrc_t frs = out_rsrc(a.out + static_cast<unsigned long long>(quad_s) * 4ull * nfl, nvalid * nfl * 4u);
                const __amdgpu_buffer_rsrc_t ers = out_rsrc(a.out_energy + static_cast<unsigned long long>(quad_s) * 4ull, nvalid * 4u);
                const int rowb = f * static_cast<int>(nfl);
                const float e0 = m0 * hs, e1 = m1 * hs, e2 = m2 * hs;
                buf_store(e0 == 0.f ? kEps : e0, frs, fidx0 >= 0 ? (rowb + fidx0) * 4 : kOobOffset);
                buf_store(e1 == 0.f ? kEps : e1, frs, fidx1 >= 0 ? (rowb + fidx1) * 4 : kOobOffset);
                buf_store(e2 == 0.f ? kEps : e2, frs, fidx2 >= 0 ? (rowb + fidx2) * 4 : kOobOffset);
                const float en = energy * (1.0f / kTwo32);  // exact: power-of-two scaling
                buf_store(en, ers, j == 0 ? f * 4 : kOobOffset);
            }
            wave_order();
            quad = next;
            continue;
        }
        SS_PRIOL(SS_P2_DCT);
        SS_PH(7);  // mel
        m0 *= hscale32;  // mel energies * 2^32 (see ln_scaled)
        m1 *= hscale32;
        m2 *= hscale32;
        float acc = 0.f;
        if (SS_ABLATE & 32) {
            acc = ln_scaled(m0 + m1 + m2);
        } else if (SYM && PAIRED) {
            // ---- DCT-II with cos(pi c (2(39-m)+1)/80) = (-1)^c cos(pi c (2m+1)/80): an even coefficient is a 20-term product with
            // s[m] = L[m] + L[39-m], an odd one with d[m] = L[m] - L[39-m].  The host laid the cells out so that this lane's slot 0
            // holds filter 39 - j and its slot 2 (j < 8) or slot 1 (j >= 8) filter j, and lanes 2i, 2i + 1 < 8 of slot 1 the pair
            // (16 + i, 23 - i): every s and d is formed in registers and goes straight to the sum / difference row -- no ln(mel)
            // row, one LDS round trip from the logarithms to the products instead of three ----
            const float l0 = ln_scaled(m0 == 0.f ? kEps * kTwo32 : m0);  // L[39 - j]
            const float l1 = ln_scaled(m1 == 0.f ? kEps * kTwo32 : m1);  // L[16 + j/2] (even j < 8), L[23 - j/2] (odd j < 8), L[j] (j >= 8)
            const float l2 = ln_scaled(m2 == 0.f ? kEps * kTwo32 : m2);  // L[j] (j < 8)
            const float p1 = dpp<0xB1>(l1);                               // quad_perm [1,0,3,2]: the neighbouring lane's
            const float lj = j < 8 ? l2 : l1;
            sd[j] = lj + l0;
            sd[20 + j] = lj - l0;
            // the four middle pairs, from the even lanes below 8; every other lane writes the row's dump words
            const int mi = (j < 8 && !(j & 1)) ? 16 + (j >> 1) : 40 + (j & 3);
            sd[mi] = l1 + p1;
            sd[(mi < 40 ? 20 : 4) + mi] = l1 - p1;
            wave_order();
            const float4 *r4 = reinterpret_cast<const float4 *>(sd + ((j & 1) ? 20 : 0));
            float4 rq[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) rq[i] = r4[i];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                acc = fmaf(rq[i].x, ch[i].x, acc);
                acc = fmaf(rq[i].y, ch[i].y, acc);
                acc = fmaf(rq[i].z, ch[i].z, acc);
                acc = fmaf(rq[i].w, ch[i].w, acc);
            }
        } else if (SYM) {
            // ---- DCT-II with cos(pi c (2(39-m)+1)/80) = (-1)^c cos(pi c (2m+1)/80): natural-order row, then the sum and
            // difference rows once per frame; an even coefficient is a 20-term product with s, an odd one with d ----
            fr0[0] = ln_scaled(m0 == 0.f ? kEps * kTwo32 : m0);
            fr1[0] = ln_scaled(m1 == 0.f ? kEps * kTwo32 : m1);
            fr2[0] = ln_scaled(m2 == 0.f ? kEps * kTwo32 : m2);
            wave_order();
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int m = j + 16 * h2;
                if (m < 20) {
                    const float lo = frow[m], hi = frow[39 - m];
                    sd[m] = lo + hi;
                    sd[20 + m] = lo - hi;
                }
            }
            wave_order();
            const float4 *r4 = reinterpret_cast<const float4 *>(sd + ((j & 1) ? 20 : 0));
            float4 rq[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) rq[i] = r4[i];
            if (!TABREG) {
                const float4 *h4 = reinterpret_cast<const float4 *>(s_tab + L::kCosH + j * 20);
#pragma unroll
                for (int i = 0; i < 5; ++i) ch[i] = h4[i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                acc = fmaf(rq[i].x, ch[i].x, acc);
                acc = fmaf(rq[i].y, ch[i].y, acc);
                acc = fmaf(rq[i].z, ch[i].z, acc);
                acc = fmaf(rq[i].w, ch[i].w, acc);
            }
        } else {
        frow[j] = ln_scaled(m0 == 0.f ? kEps * kTwo32 : m0);
        frow[16 + j] = ln_scaled(m1 == 0.f ? kEps * kTwo32 : m1);
        frow[32 + j] = ln_scaled(m2 == 0.f ? kEps * kTwo32 : m2);
        wave_order();

        // ---- DCT-II, first n_ceps coefficients (feature.rs:120-123): lane c against the 48-entry row ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // NQ float4s of the (slot, lane)-ordered row carry filters; two batches of fetches
            constexpr int HB = NQ / 2;
            float4 lq[HB], cq[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                lq[i] = *reinterpret_cast<const float4 *>(&frow[4 * (HB * h + i)]);
                cq[i] = (RES & 1) ? cr[HB * h + i] : c4[HB * h + i];
            }
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                acc = fmaf(lq[i].x, cq[i].x, acc);
                acc = fmaf(lq[i].y, cq[i].y, acc);
                acc = fmaf(lq[i].z, cq[i].z, acc);
                acc = fmaf(lq[i].w, cq[i].w, acc);
            }
        }
        }
        SS_PH(8);  // ln + DCT
        // ---- scaling + column-0 replacement (feature.rs:126-146) and the store ----
        {
            const unsigned gf = quad * 4 + f;
            // (sc_lane: this lane's column scale, a loop invariant; the two special cases of column 0 sit behind a uniform branch each,
            // so that the default path pays one product and one select)
            float o = acc * sc_lane;
            if (a.dc_elimination) {
                const float le = ln_scaled(energy);
                o = j == 0 ? le : o;
            } else if (t_cur == 0 && j == 0) {
                o = acc * a.dct_scale_00;
            }
            // unconditional, counted store (ss_wave.h): the descriptor covers the quad's valid frames, lanes j >= n_ceps are
            // dropped by its range check -- the next quad's samples are waited for with this store still in flight
            (void)gf;
            unsigned quad_s = __builtin_amdgcn_readfirstlane(quad);  // uniform, but born from the wave number: a VGPR to the compiler
            unsigned q_total = total;
            float *q_out = a.out;
            if constexpr (MULTI) {  // this quad's batch: its own output block, the quad's index and the frame count within it
                quad_s -= cs.u0;
                q_total = cs.total;
                q_out = cs.out;
            }
            const unsigned nvalid = min(4u, q_total - quad_s * 4);
            const __amdgpu_buffer_rsrc_t orow = out_rsrc(q_out + static_cast<unsigned long long>(quad_s) * 4ull * Cc, nvalid * Cc * 4u);
            buf_store(o, orow, j < Cc ? (f * Cc + j) * 4 : kOobOffset);
        }
        wave_order();
        SS_PH(9);  // store
        SS_PRIOL(SS_P2_TOP);
        quad = next;
    }
#if SS_PROF2
    if (a.dbg && lane == 0) {
        unsigned long long *o = a.dbg + 16ull * (blockIdx.x * WAVES + wave);
        pacc[11] = __builtin_amdgcn_s_memtime() - tstart;
        pacc[0] = n_done;
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = pacc[k];
    }
#endif
    if (!SS_PROF2 && a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + 6ull * (blockIdx.x * WAVES + wave);
        d[2] = __builtin_amdgcn_s_memrealtime();
        if (wave >= kTabWaves) d[5] = (__builtin_amdgcn_s_memtime() - d[5]) | (1ull << 40);  // the other waves: cycles lived
        d[3] = (static_cast<unsigned long long>(n_done) << 32) | __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // XCC_ID
    }
}

constexpr int NE13RES = 4;  // resident-table set of the 13-input default-bank build (symmetric DCT; the twiddles would spill)

template <int WAVES>
hipError_t launch_w(const Fast512Args &a_in, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    Fast512Args a = a_in;
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
            // quad_src (the SPREAD builds' sample addresses): a lane's 32-bit byte offset from the quad's uniform base reaches
            // 3 frames + one step into the next clip + its 16 sample pairs + the sixteen 128-byte strides of the loads, and the
            // frame-in-quad product is a 24-bit multiply.  Row strides / hops beyond that go to the next kernel
            // (launch_frames falls through on this value), which forms 64-bit addresses.
            const unsigned long long span = static_cast<unsigned long long>(a.n_frames) * a.step;
            // (centred frames never take quad_src, and their n_frames * step may exceed the clip: not checked)
            if (!a.center && (a.ld < span || static_cast<unsigned long long>(a.step) * 4ull >= (1ull << 24) ||
                              3ull * a.step * 4ull + (a.ld - span) * 4ull + 16ull * 8ull + 16ull * 128ull >= (1ull << 32)))
                return hipErrorInvalidValue;
        } else if (static_cast<unsigned long long>(a.batch) * a.ld * 4ull >= (1ull << 32)) {
            return hipErrorInvalidValue;  // without the reciprocal the kernel addresses a lane's frame by a 32-bit offset from the batch's first sample
        }
    }
    const size_t lds = (static_cast<size_t>(WAVES) * (WAVES > 12 ? 4 * kZStride * 2 : kWaveFloats) + L::kMelW + 16 * a.mel_wpitch + (a.win_floats > 0 ? a.win_floats : 0)) * sizeof(float) + 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    const unsigned long long quads = (total + 3) / 4;
    // one workgroup per CU; fewer when there is not at least one quad per wave
    unsigned long long blocks = (quads + WAVES - 1) / WAVES;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    a.q_base = static_cast<uint32_t>(quads / grid);
    a.q_rem = static_cast<uint32_t>(quads % grid);
    auto go = [&](auto kern, const char *name) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds));
            if (e != hipSuccess) return e;
        }
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(WAVES * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a, MultiArg<false>{});
        return hipGetLastError();
    };
    const bool pow2 = a.spectrum_exponent == 2;
    const bool b421 = a.mel_q4[0] == 4 && a.mel_q4[1] == 2 && a.mel_q4[2] == 1;
    if (a.fullp || a.center) {
        // librosa-compatible variants: P rows of all 257 bins (banks that cover the whole spectrum) and / or centred frames;
        // MFCC output, optional frame window, no fused pre-emphasis
        if constexpr (WAVES != 12) {
            return hipErrorInvalidValue;
        } else {
            if (a.out_mfe || a.preemph != 0.0f) return hipErrorInvalidValue;
            const bool win = a.win_floats > 0;
#define SS_LV(P2, FR, FP, CE, NAME) return go(ss_mfcc_c256<16, false, P2, WAVES, false, 12, 0, 0, FR, FP, CE>, NAME)
            if (a.fullp && !a.center) {
                if (pow2) { if (win) SS_LV(true, 1, true, false, "ss_mfcc_c256<16,pow2,win,fullp>"); SS_LV(true, 0, true, false, "ss_mfcc_c256<16,pow2,fullp>"); }
                if (win) SS_LV(false, 1, true, false, "ss_mfcc_c256<16,win,fullp>");
                SS_LV(false, 0, true, false, "ss_mfcc_c256<16,fullp>");
            }
            if (!a.fullp) {
                if (pow2) { if (win) SS_LV(true, 1, false, true, "ss_mfcc_c256<16,pow2,win,center>"); SS_LV(true, 0, false, true, "ss_mfcc_c256<16,pow2,center>"); }
                if (win) SS_LV(false, 1, false, true, "ss_mfcc_c256<16,win,center>");
                SS_LV(false, 0, false, true, "ss_mfcc_c256<16,center>");
            }
            if (pow2) { if (win) SS_LV(true, 1, true, true, "ss_mfcc_c256<16,pow2,win,fullp,center>"); SS_LV(true, 0, true, true, "ss_mfcc_c256<16,pow2,fullp,center>"); }
            if (win) SS_LV(false, 1, true, true, "ss_mfcc_c256<16,win,fullp,center>");
            SS_LV(false, 0, true, true, "ss_mfcc_c256<16,fullp,center>");
#undef SS_LV
        }
    }
    // register-resident tables (bit 0 cosines, bit 1 twiddles, bit 2 symmetric DCT).  2: twiddles resident (140 VGPRs, 0.7 us
    // faster than 0; 3 spills); 6: + symmetric DCT (165 VGPRs, another 1.2 %)
    int res = 6;
#if SS_LAB
    static const char *res_env = std::getenv("SS_RES");  // A/B knob (lab build)
    if (res_env) res = std::atoi(res_env);
#endif
    if (a.flen == 320 && !pow2 && b421 && a.n_filters <= 40) {
        const int front = (a.win_floats > 0 ? 1 : 0) | (a.preemph != 0.0f ? 2 : 0);
        if (a.out_mfe == 2) {
            if (front || WAVES > 12) return hipErrorInvalidValue;
            return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2, 2>, "ss_mfcc_c256<10,exact,power>");
        }
        if (front) {
            if (WAVES > 12) return hipErrorInvalidValue;
            if (a.out_mfe) {
                if (front == 1) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2, 1, 1>, "ss_mfcc_c256<10,exact,bank421,mfe,win>");
                if (front == 2) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 0, 1, 2>, "ss_mfcc_c256<10,exact,bank421,mfe,pre>");
                return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 0, 1, 3>, "ss_mfcc_c256<10,exact,bank421,mfe,win,pre>");
            }
            if (front == 1) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2, 0, 1>, "ss_mfcc_c256<10,exact,bank421,win>");
            if (front == 2) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2, 0, 2>, "ss_mfcc_c256<10,exact,bank421,pre>");
            return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2, 0, 3>, "ss_mfcc_c256<10,exact,bank421,win,pre>");
        }
        if (a.out_mfe) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2, 1>, "ss_mfcc_c256<10,exact,bank421,mfe>");
        if (res == 6 && a.n_filters == 40 && a.paired == 2 && WAVES <= 12)
            return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 30>, "ss_mfcc_c256<10,exact,bank421,sym>");
        if (res == 6 && a.n_filters == 40 && a.paired && WAVES <= 12)
            return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 14>, "ss_mfcc_c256<10,exact,bank421,sym,taps84>");
        if (res == 6 && a.n_filters == 40)
            return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, WAVES <= 12 ? 6 : 4>, WAVES <= 12 ? "ss_mfcc_c256<10,exact,bank421,symrow>" : "ss_mfcc_c256<10,exact,bank421,sym,w16>");
        if (res == 6) res = 2;  // the symmetric DCT is written for exactly 40 filters
        if (WAVES <= 12 && res == 1) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 1>, "ss_mfcc_c256<10,exact,bank421,res1>");
        if (WAVES <= 12 && res == 2) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 2>, "ss_mfcc_c256<10,exact,bank421,res2>");
        if (WAVES <= 12 && res == 3) return go(ss_mfcc_c256<10, true, false, WAVES, true, 10, 3>, "ss_mfcc_c256<10,exact,bank421,res3>");
        return go(ss_mfcc_c256<10, true, false, WAVES, true, 10>, "ss_mfcc_c256<10,exact,bank421>");
    }
    if (a.flen == 320) {
        return pow2 ? go(ss_mfcc_c256<10, true, true, WAVES, false, 12>, "ss_mfcc_c256<10,exact,pow2>")
                    : go(ss_mfcc_c256<10, true, false, WAVES, false, 12>, "ss_mfcc_c256<10,exact>");
    }
    if (a.flen <= 320) {
        return pow2 ? go(ss_mfcc_c256<10, false, true, WAVES, false, 12>, "ss_mfcc_c256<10,pow2>")
                    : go(ss_mfcc_c256<10, false, false, WAVES, false, 12>, "ss_mfcc_c256<10>");
    }
    if (a.flen <= 416) {  // 25 ms at 16 kHz (400 samples): 13 of the 16 first-pass inputs
        // the bank does not depend on the frame length: the default bank keeps its fixed tap counts and symmetric DCT
        if (!pow2 && b421 && a.n_filters == 40 && WAVES <= 12 && !a.out_mfe && a.win_floats == 0 && a.preemph == 0.0f)
            return go(ss_mfcc_c256<13, false, false, WAVES, true, 10, NE13RES>, "ss_mfcc_c256<13,bank421,sym>");
        return pow2 ? go(ss_mfcc_c256<13, false, true, WAVES, false, 12>, "ss_mfcc_c256<13,pow2>")
                    : go(ss_mfcc_c256<13, false, false, WAVES, false, 12>, "ss_mfcc_c256<13>");
    }
    return pow2 ? go(ss_mfcc_c256<16, false, true, WAVES, false, 12>, "ss_mfcc_c256<16,pow2>")
                : go(ss_mfcc_c256<16, false, false, WAVES, false, 12>, "ss_mfcc_c256<16>");
}

}  // namespace

hipError_t launch_mfcc_c256_multi(const Fast512Args &a_in, int n_batches, const float *const *d_x, float *const *d_out, const size_t *clips,
                                  hipStream_t stream, int num_cus, LaunchInfo *info)
{
    constexpr int WAVES = 12;
    Fast512Args a = a_in;
    // the builds that exist: MFCC output of the default frame shape and bank (what launch_w's first branch serves without a
    // window, pre-emphasis or another output), 40 filters in the paired tight-tap layout -- the headline build
    const bool b421 = a.mel_q4[0] == 4 && a.mel_q4[1] == 2 && a.mel_q4[2] == 1;
    if (n_batches < 1 || n_batches > kMaxLaunchBatches || a.fullp || a.center || a.out_mfe || a.win_floats > 0 || a.preemph != 0.0f ||
        a.flen != 320 || a.spectrum_exponent == 2 || !b421 || a.n_filters != 40 || a.paired != 2 || a.n_frames < 4)
        return hipErrorInvalidValue;
    Fast512Multi m{};
    unsigned long long quads = 0, max_total = 0;
    for (int b = 0; b < kMaxLaunchBatches; ++b) {
        m.uend[b] = 0xffffffffu;
        if (b >= n_batches) continue;
        const unsigned long long tot = static_cast<unsigned long long>(clips[b]) * a.n_frames;
        if (tot == 0 || tot + 4 >= (1ull << 31)) return hipErrorInvalidValue;  // (empty batches are dropped by the caller)
        quads += (tot + 3) / 4;
        if (quads >= 0xffffffffull) return hipErrorInvalidValue;
        max_total = std::max(max_total, tot);
        m.x[b] = d_x[b];
        m.out[b] = d_out[b];
        m.uend[b] = static_cast<uint32_t>(quads);
        m.total[b] = static_cast<uint32_t>(tot);
    }
    {
        // the same reciprocal and address-range conditions as launch_w (every batch has the same clip shape)
        const unsigned long long d = a.n_frames;
        unsigned l = 0;
        while ((1ull << l) < d) ++l;
        const unsigned __int128 num = static_cast<unsigned __int128>(1) << (31 + l);
        a.nf_magic = static_cast<uint32_t>((num + d - 1) / d);
        a.nf_shift = l - 1;
        const unsigned long long span = static_cast<unsigned long long>(a.n_frames) * a.step;
        if (a.ld < span || static_cast<unsigned long long>(a.step) * 4ull >= (1ull << 24) ||
            3ull * a.step * 4ull + (a.ld - span) * 4ull + 16ull * 8ull + 16ull * 128ull >= (1ull << 32))
            return hipErrorInvalidValue;
    }
    a.x = d_x[0];
    a.out = d_out[0];
    a.batch = static_cast<uint32_t>(clips[0]);
    const size_t lds = (static_cast<size_t>(WAVES) * kWaveFloats + L::kMelW + 16 * a.mel_wpitch) * sizeof(float) + 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    unsigned long long blocks = (quads + WAVES - 1) / WAVES;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    a.q_base = static_cast<uint32_t>(quads / grid);
    a.q_rem = static_cast<uint32_t>(quads % grid);
    auto kern = ss_mfcc_c256<10, true, false, WAVES, true, 10, 30, 0, 0, false, false, true>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    if (info) *info = LaunchInfo{"ss_mfcc_c256m<10,exact,bank421,sym>", grid, static_cast<unsigned>(WAVES * 64), lds};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a, MultiArg<true>{m});
    return hipGetLastError();
}

bool mfcc_c256_has_mfe(const Fast512Args &a)
{
    return a.flen == 320 && a.spectrum_exponent != 2 && a.mel_q4[0] == 4 && a.mel_q4[1] == 2 && a.mel_q4[2] == 1 && a.n_filters <= 40;
}

hipError_t launch_mfcc_c256(const Fast512Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    // mfe output, frame window and pre-emphasis exist for the default-bank build only (the librosa-style builds take a window)
    if (!(a.fullp || a.center) && (a.out_mfe || a.win_floats > 0 || a.preemph != 0.0f) && !mfcc_c256_has_mfe(a)) return hipErrorInvalidValue;
    // 12 waves per CU (3 per SIMD, <= 168 VGPRs, 138 KB of LDS): measured equal to 14 and 16 and 8 % faster than 8
#if SS_LAB
    static const char *w = std::getenv("SS_WAVES");  // A/B knob for occupancy experiments (lab build)
    if (w && std::atoi(w) == 8 && !a.fullp && !a.center) return launch_w<8>(a, stream, num_cus, info);
    if (w && std::atoi(w) == 16 && !a.fullp && !a.center) return launch_w<16>(a, stream, num_cus, info);
    if (w && std::atoi(w) == 10 && !a.fullp && !a.center) return launch_w<10>(a, stream, num_cus, info);
    if (w && std::atoi(w) == 11 && !a.fullp && !a.center) return launch_w<11>(a, stream, num_cus, info);
    if (w && std::atoi(w) == 9 && !a.fullp && !a.center) return launch_w<9>(a, stream, num_cus, info);
#endif
    return launch_w<12>(a, stream, num_cus, info);
}

}  // namespace ss

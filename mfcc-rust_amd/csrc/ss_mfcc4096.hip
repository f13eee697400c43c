// ss_mfcc_c2048: fused MFCC for fft_points = 4096 (C = 2048 packed complex points) on gfx950 -- BASELINE config 5
// (44.1 kHz, hop 1024, 256 mel filters, 40 cepstra).  One wave owns one frame: 64 lanes x 32 complex points.
//
//   * 2048-point FFT = 32 x 64: a radix-32 register butterfly over n2 (n = n1 + 64 n2, n1 = lane), ONE transposing
//     exchange through wave-private LDS, then the 64-point transform over n1 split as n1 = a + 2b: lane (k1, a)
//     does a radix-32 butterfly over b, and the last radix-2 over a pairs lanes L and L+32 with
//     v_permlane32_swap (both halves' values arrive in one instruction; no LDS):
//       Z[k1 + 32 c + 1024 d] = G0'[c] + (-1)^d G1'[c],  Ga'[c] = W2048^(a (k1 + 32 c)) FFT32_b(A[a+2b][k1] W1024^(b k1))[c].
//     The exchange is two independent 32 x 32 transposes (even / odd n1), run in two register halves exactly like the
//     2048-point mel kernel (ss_mel2048.hip): 8704 B of LDS per wave, conflict-free ds_write_b64 / ds_read_b128.
//   * real-FFT untangle: lane (k1, d) register c holds bin k = k1 + 32c + 1024d; its partner 2048-k sits in lane
//     (32-k1, 1-d), register 31-c (32-c for the k1 = 0 lanes, which expose their upper registers shifted by one);
//     fetched with ds_bpermute_b32.  Every lane handles its 16 lower registers; k = 0, 512/1536 and 1024 are specials.
//   * |X|/N: bins 0..1024 go to the P row (the mel bank ends at (F+1)/2, feature.rs:69-70), all 2049 feed the frame
//     energy (feature.rs:216-219), reduced over the wave.
//   * banded mel (4 filters per lane, host-sorted by tap count), zero handling, ln, then the DCT-II for the first
//     n_ceps coefficients using the m <-> M-1-m symmetry of the cosine (half the table, half the multiplies).
//   * dynamic frame scheduling from an LDS counter; no workgroup barrier in the main loop.
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

#include <cstdlib>
#include <type_traits>

// timing-attribution builds (lab build only, tools/ablate.sh; results wrong by design): 1 no DCT, 2 no mel + DCT, 4 no partner
// fetch, 8 no exchange, 16 every frame reads clip 0 (L2-resident samples), 32 no sample loads.  The product build compiles the
// switch out.
#if !SS_LAB
#undef SS_ABL5
#endif
#ifndef SS_ABL5
#define SS_ABL5 0
#endif
// SS_PROF5 (lab builds, with SS_DEBUG_ROWS=<file>): every wave sums the shader-clock ticks it spends in each phase of its
// iterations (s_memtime at the phase boundaries; the wait for the stamp also drains the phase's LDS operations) and writes 16
// 64-bit words at the end: [0] iterations, [1..] phase sums.  tools/prof5.py prints the table.
#if !SS_LAB
#undef SS_PROF5
#endif
#ifndef SS_PROF5
#define SS_PROF5 0
#endif
#if SS_PROF5
#define SS_PH(k)                                                   \
    do {                                                           \
        __builtin_amdgcn_sched_barrier(0);                         \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                         \
        pacc[k] += tn_ - tprev;                                    \
        tprev = tn_;                                               \
    } while (0)
#else
#define SS_PH(k) do { } while (0)
#endif

namespace ss {

namespace {

using namespace wv;

// Issue priorities of the phases (s_setprio; see ss_mel2048.hip): the butterflies lowest, whatever requests samples, exchanges
// through LDS or reads tables in front of them.  cfg5, same box (profiles/r03/ab_cfg2_cfg5_priorities*.txt): 64.1 us without, 61.2
// with 3 / 1 / 1 / 2 / 3.  SS_PRIOS5 (lab builds): five decimal digits -- loop top (sample request),
// exchange + twiddles, radix-2 stage, untangle, mel + DCT + store.  Only the twelve-wave build sets priorities.
#if SS_LAB && defined(SS_PRIOS5)
#define SS_P5_TOP ((SS_PRIOS5 / 10000) % 10)
#define SS_P5_EX ((SS_PRIOS5 / 1000) % 10)
#define SS_P5_R2 ((SS_PRIOS5 / 100) % 10)
#define SS_P5_UN ((SS_PRIOS5 / 10) % 10)
#define SS_P5_MEL (SS_PRIOS5 % 10)
#else
#define SS_P5_TOP 3
#define SS_P5_EX 1
#define SS_P5_R2 1
#define SS_P5_UN 2
#define SS_P5_MEL 3
#endif
#define SS_PRIOL(x) do { if (LEAN && !kNoPrio) __builtin_amdgcn_s_setprio(x); } while (0)
#if SS_LAB && defined(SS_NOPRIO5)
constexpr bool kNoPrio = true;
#else
constexpr bool kNoPrio = false;
#endif
#if SS_LAB && defined(SS_PF5)
constexpr bool kNoPf = false;  // A/B (lab builds, -DSS_PF5=1): the next frame's samples requested in front of the DCT stage
#else
constexpr bool kNoPf = true;   // measured and not kept (profiles/r04/ab_cfg5_prefetch.txt): 62.05 against 62.36 us, inside the noise
#endif
namespace L = mfcc4096_layout;
constexpr int kClsStride = 16 * 34 + 8;    // float2 per class slice: +8 keeps the two classes of a write group 16 banks apart
constexpr int kExFloats = (kClsStride + 16 * 34) * 2;  // exchange region (two classes x half the columns, 8768 B); P row + ln(mel) row reuse it
constexpr bool kDbgStages = false;        // true: SS_DEBUG_ROWS also dumps frame 0's registers after each FFT stage (tools/dbg4096.py)
constexpr int kFRowOff = L::kPRow;        // ln(mel) row [256] behind the P row (both inside the exchange region)
constexpr int kSRowOff = kFRowOff + 256;  // s and d rows of the symmetric DCT (4 segments of 64 + 4 floats), behind the ln(mel) row
constexpr int kSegPitch = 68;
constexpr int kMelPitch8321 = 60;  // floats per lane row of mel weights for the 8 / 3 / 2 / 1 bank (14 float4s + 1: odd pitch in 16-byte units)


// MULTI (ss_mfcc_batches_device): the launch's frames are the concatenation of up to kMaxLaunchBatches batches' frames, each batch
// with its own input and output block (BatchTable, ss_device.h; Seg / seg_of, ss_wave.h); the default-shape MFCC build only.
template <bool EXACT, bool POW2, int WAVES, bool MFE = false, bool WIN = false, bool PRE = false, bool FIXMEL = false, bool MULTI = false>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void ss_mfcc_c2048(const Mfcc4096Args a, const MultiArg<MULTI> mt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const LifeStamp life = life_begin(WAVES > 8 ? a.stamps : nullptr);  // (diagnostic, twelve-wave build: null in every ordinary launch)
    // LEAN: twelve waves per CU (three per SIMD) at <= 168 VGPRs.  Every table read comes in batches of at most eight float4s
    // (a third wave on the SIMD hides the round trips the 8-wave build has to avoid) and what depends on the lane number only
    // is derived again in every iteration instead of living in registers: twelve exchange regions + the tables are 163 584 of
    // the 163 840 bytes of LDS.
    constexpr bool LEAN = WAVES > 8;
    float *wbase = reinterpret_cast<float *>(smem) + wave * kExFloats;
    float2 *ex = reinterpret_cast<float2 *>(wbase);
    float *prow = wbase;             // P[0..1024] + zero pad bins, after the exchange
    float *frow = wbase + kFRowOff;  // ln(mel) in filter order
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kExFloats;
    const float4 *s_t1 = reinterpret_cast<const float4 *>(s_tab + L::kT1);
    const float2 *s_t2 = reinterpret_cast<const float2 *>(s_tab + L::kT2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const float *s_cos = s_tab + L::kCos;
    const int Cc = static_cast<int>(a.n_ceps);
    const int melw0 = L::kCos + a.cos_floats;
    const float *s_melw = s_tab + melw0;
    // WIN: the frame window (4096 floats, zero beyond flen) sits behind the table block
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + melw0 + 64 * a.mel_wpitch);
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + melw0 + 64 * a.mel_wpitch + (WIN ? 4096 : 0));

    const unsigned total = MULTI ? a.batch : a.batch * a.n_frames;  // (MULTI: the launcher hands over the launch's frame count)
    const unsigned f_lo = static_cast<unsigned>(static_cast<unsigned long long>(total) * blockIdx.x / gridDim.x);
    const unsigned f_hi = static_cast<unsigned>(static_cast<unsigned long long>(total) * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (melw0 + 64 * a.mel_wpitch) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (WIN) {
            float *wdst = s_tab + melw0 + 64 * a.mel_wpitch;
            for (int i = tid; i < 4096; i += WAVES * 64) wdst[i] = i < static_cast<int>(a.flen) ? a.window[i] : 0.f;
        }
        if (tid == 0) *s_next = f_lo + WAVES;
    }
    __syncthreads();
    const float hscale32 = (POW2 ? 0.25f * a.scale : 0.5f * a.scale) * kTwo32;
    const int M = FIXMEL ? 256 : static_cast<int>(a.n_filters), Mh = M / 2;  // (FIXMEL: the launcher checked n_filters == 256 -- the DCT stage's lane guards fold)
    constexpr bool pre = PRE;  // fused pre-emphasis: builds of their own (the taps cost registers the plain builds do not have)
    const unsigned psh = PRE ? a.preemph_shift % a.n_samples : 0u;

    // (uniform, but born from the wave number: without the readfirstlane it is a VGPR to the compiler and the clip / frame split and
    // the 64-bit source address of every iteration are computed on the VALU -- ~25 instructions with five quarter-rate multiplies)
    unsigned frame = __builtin_amdgcn_readfirstlane(f_lo + wave);
    // PF (lab builds with -DSS_PF5=1; the twelve-wave build of the default cfg5 shape): the next frame's samples are requested in front of the DCT stage --
    // the one stretch of the iteration in which the transform's 64 registers are free -- so their round trip runs under the
    // stage's 32 table reads and 64 FMAs instead of being waited for at the top of the next iteration (12 % of a wave's time in
    // the round-4 phase profile).  The stage's one output store is a counted store (ss_wave.h): the samples are then waited for
    // with it still in flight.  Result: no change (three waves per SIMD already cover a wave's wait for its samples).
    constexpr bool PF = LEAN && FIXMEL && EXACT && !MFE && !WIN && !PRE && !SS_PROF5 && SS_ABL5 == 0 && !kNoPf;
    static_assert(!MULTI || (LEAN && FIXMEL && EXACT && !MFE && !WIN && !PRE), "the batch-table build exists for the default-shape MFCC build");
    float2 vpf[PF ? 32 : 1];
    Seg ns{};  // MULTI: the batch of the frame whose samples are being fetched (PF) / of the current frame
    if constexpr (MULTI && !PF) ns = seg_of(mt.m, min(frame, f_hi - 1));
    if (PF) {
        unsigned fr0 = min(frame, f_hi - 1);  // (a wave without a frame loads the block's last one: no branch around the loads)
        const float *x0 = a.x;
        if constexpr (MULTI) {
            ns = seg_of(mt.m, fr0);
            fr0 -= ns.u0;
            x0 = ns.x;
        }
        const unsigned clip0 = fr0 / a.n_frames;
        const float2 *src0 = reinterpret_cast<const float2 *>(x0 + static_cast<unsigned long long>(clip0) * a.ld + (fr0 - clip0 * a.n_frames) * a.step) + (threadIdx.x & 63);
#pragma unroll
        for (int e = 0; e < 32; ++e) vpf[e] = src0[64 * e];
        buf_store(0.f, out_rsrc(a.out, 0u), 0);  // both ways into the loop have one store behind the samples (see ss_mfcc512.hip)
    }
#if SS_PROF5
    unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif
    SS_PRIOL(SS_P5_TOP);
    while (frame < f_hi) {
        // the claim of the next frame is issued here and read where it is needed (the end of the iteration): the LDS atomic's
        // round trip hides behind the transform
        // What depends on the lane number only.  8 waves per CU: loop invariants, the compiler keeps them in registers across
        // the iterations.  LEAN: the lane number is made opaque here, so that they are derived again in every iteration
        // (~40 integer instructions per frame) instead of occupying ~40 of the 168 registers for the whole kernel.
        int lane_it = static_cast<int>(threadIdx.x) & 63;
        if (LEAN) asm volatile("" : "+v"(lane_it));
        const int lane = lane_it & 63;  // (the mask tells the compiler the range again: 24-bit multiplies, no sign extensions)
        const int k1 = lane & 31, d = lane >> 5;   // reader view: column k1, half a = d
        const int cls = lane & 1, bw = lane >> 1;  // writer view: n1 = lane = cls + 2 bw
        // (FIXMEL: the 8 / 3 / 2 / 1 bank's rows are 15 float4s apart -- checked by the launcher -- so the lane's row is two shifts,
        // not a quarter-rate multiply each time the compiler derives it again)
        const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + lane * (FIXMEL ? kMelPitch8321 : static_cast<int>(a.mel_wpitch)));
        const int paddr = ((((32 - k1) & 31) | ((1 - d) << 5))) << 2;  // lane holding Z[2048 - k]
        float2 *exw = ex + cls * kClsStride + 34 * (bw >> 1) + (bw & 1);  // writer base (float2 units)
        const float2 *exr = ex + d * kClsStride + 2 * (k1 & 15);        // reader base
        const bool k1z = k1 == 0;
        unsigned next_v = 0;
        if (lane == 0) next_v = atomicAdd(s_next, 1u);
#if SS_PROF5
        pacc[0] += 1;
#endif
        SS_PH(1);  // claim

        if constexpr (MULTI && !PF) {
            if (frame >= ns.u1) ns = seg_of(mt.m, frame);  // (uniform, rare: the claimed frame starts the next batch)
        }
        const Seg cs = ns;  // MULTI: the batch of this iteration's frame (PF: its samples were fetched from it)
        (void)cs;
        const unsigned frame_b = MULTI ? frame - cs.u0 : frame;  // the frame's index within its batch
        const unsigned clip = frame_b / a.n_frames;
        const unsigned t = frame_b - clip * a.n_frames;
        // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
        // (SS_ABL5 & 16: every frame reads clip 0 -- L2-resident samples; & 32: no sample loads at all)
        // (uniform base + this lane's 32-bit byte offset: the loads take the SGPR-base form, no 64-bit address is formed on the VALU)
        const char *src_b = reinterpret_cast<const char *>((MULTI ? cs.x : a.x) + static_cast<unsigned long long>((SS_ABL5 & 16) ? 0u : clip) * a.ld + t * a.step);
        const unsigned src_o = static_cast<unsigned>(lane) * 8u;
        const float2 *src = reinterpret_cast<const float2 *>(src_b + src_o);
        unsigned src_o4[4] = {src_o, src_o + 4096u, src_o + 8192u, src_o + 12288u};
        asm volatile("" : "+v"(src_o4[1]), "+v"(src_o4[2]), "+v"(src_o4[3]));  // (kept as 32-bit offsets: left alone they become 64-bit VALU address sums)
        float2 v[32];
        if (PF) {
#pragma unroll
            for (int e = 0; e < 32; ++e) v[e] = vpf[e];
        }
#pragma unroll
        for (int e = 0; e < (PF ? 0 : 32); ++e) {
            if (SS_ABL5 & 32) {
                v[e] = make_float2(1e-3f * static_cast<float>(lane + e), 2e-3f * static_cast<float>(frame & 255u));
                continue;
            }
            if (EXACT) v[e] = *reinterpret_cast<const float2 *>(src_b + src_o4[e / 8] + 512u * (e % 8));  // (a 32-bit lane offset per 4096 bytes: the rest fits the instruction)
            else {  // zero pad, processing.rs:147-156; an odd frame length ends in a half pair
                const int rem = static_cast<int>(a.flen) - 2 * (lane + 64 * e);
                v[e] = rem >= 2 ? src[64 * e] : make_float2(rem == 1 ? reinterpret_cast<const float *>(src)[128 * e] : 0.f, 0.f);
            }
            if (pre) {  // fused pre-emphasis (processing.rs:31-53) of the samples that exist
                const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
                const int pos = static_cast<int>(t * a.step) + 2 * (lane + 64 * e), rem = static_cast<int>(a.flen) - 2 * (lane + 64 * e);
                if (rem >= 1) v[e].x = fmaf(-a.preemph, preemph_tap(xc, pos, psh, a.n_samples), v[e].x);
                if (rem >= 2) v[e].y = fmaf(-a.preemph, preemph_tap(xc, pos + 1, psh, a.n_samples), v[e].y);
            }
        }
        if (WIN) {
            // optional frame window (mfcc_window switch): sample pairs from the copy in LDS
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const float2 w = s_win[lane + 64 * e];
                v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
            }
        }
#if SS_PROF5
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SS_PH(2);  // sample loads arrived
        SS_PRIOL(0);
        // ---- pass 1: radix-32 over n2 ----
        fft_reg<32>(v);
        SS_PRIOL(SS_P5_EX);
        SS_PH(3);  // pass 1

        if (kDbgStages && a.dbg && frame == 0) {
#pragma unroll
            for (int e = 0; e < 32; ++e) reinterpret_cast<float2 *>(a.dbg + 1284 + 0 * 4096)[lane * 32 + e] = v[e];
        }
        // ---- transpose (two 32 x 32 problems: even and odd n1), in two register halves ----
        float2 u[32];
        if (SS_ABL5 & 8) {
#pragma unroll
            for (int k = 0; k < 32; ++k) u[k] = v[k];
        } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) exw[2 * k] = v[k];
        wave_order();
        if (LEAN) {
            // Every lane fills u in ONE of the two half-wave phases below.  Left half-defined, the "undefined" halves are carried
            // around the loop and spilled (132 bytes of scratch at 168 VGPRs); an empty asm statement per register defines them
            // at no cost.  (Sixty-four moves did the same for 4 % more instructions; running the first phase's reads on all
            // lanes -- lanes k1 >= 16 reading the column of lane k1 - 16, a broadcast -- measured 1.7 % slower.)
#pragma unroll
            for (int k = 0; k < 32; ++k) u[k] = make_float2(defined_garbage(), defined_garbage());
        }
        if (k1 < 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
        }
        wave_order();
#pragma unroll
        for (int k = 0; k < 16; ++k) exw[2 * k] = v[16 + k];
        wave_order();
        if (k1 >= 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
        }
        wave_order();
        }
        if (kDbgStages && a.dbg && frame == 0) {
#pragma unroll
            for (int e = 0; e < 32; ++e) reinterpret_cast<float2 *>(a.dbg + 1284 + 1 * 4096)[lane * 32 + e] = u[e];
        }
        // The pass-2 twiddles are all requested here, right behind the exchange's reads and in front of the first product: read
        // where they are used, they came two at a time, each pair one exposed LDS round trip (nobody hides it with two waves per
        // SIMD).  (Requested before the exchange they are live across it and 19 loop invariants spill.)
        // (LEAN: two batches of eight)
        SS_PH(4);  // exchange
        // ---- twiddle W1024^(b k1), radix-32 over b, twiddle W2048^(a (k1 + 32 c)) ----
        constexpr int kTwBatch = LEAN ? 8 : 16;
#pragma unroll
        for (int pb = 0; pb < 16; pb += kTwBatch) {
            float4 tw1[kTwBatch];
#pragma unroll
            for (int p = 0; p < kTwBatch; ++p) tw1[p] = s_t1[(pb + p) * 32 + k1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < kTwBatch; ++p) {
                const float4 w2 = tw1[p];
                u[2 * (pb + p) + 1] = cmul(u[2 * (pb + p) + 1], make_float2(w2.x, w2.y));
                if (pb + p < 15) u[2 * (pb + p) + 2] = cmul(u[2 * (pb + p) + 2], make_float2(w2.z, w2.w));
            }
        }
        SS_PH(5);  // twiddles
        SS_PRIOL(0);
        fft_reg<32>(u);  // u[c] = G_a[c], a = lane >> 5
        SS_PRIOL(SS_P5_R2);
        SS_PH(6);  // pass 2
        if (kDbgStages && a.dbg && frame == 0) {
#pragma unroll
            for (int e = 0; e < 32; ++e) reinterpret_cast<float2 *>(a.dbg + 1284 + 2 * 4096)[lane * 32 + e] = u[e];
        }
        // ---- radix-2 over a.  One v_permlane32_swap of registers (i, 16 + i) hands the lower half-wave both halves'
        // column i and the upper half-wave both halves' column 16 + i; each lane then forms BOTH outputs
        //   Z[k1 + 32 c + 1024 d] = G0[c] + (-1)^d W2048^(k1 + 32 c) G1[c],  c = i + 16 h, h = lane >> 5
        // r0[i] (d = 0) and r1[i] (d = 1): half the swaps, half the twiddle products, no copies ----
        // (all sixteen twiddles are requested in front of the swaps: read inside the loop they came one iteration ahead of their
        // use -- less than an LDS round trip -- and every iteration waited: 3.4 k of a wave's 20 k cycles per frame)
        float2 t2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t2[i] = s_t2[i * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        float2 r0[16], r1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float px = u[i].x, qx = u[16 + i].x, py = u[i].y, qy = u[16 + i].y;
            swap_halves(px, qx);
            swap_halves(py, qy);
            const float2 wq = cmul(make_float2(qx, qy), t2[i]);
            r0[i] = make_float2(px + wq.x, py + wq.y);
            r1[i] = make_float2(px - wq.x, py - wq.y);
        }
        if (kDbgStages && a.dbg && frame == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                reinterpret_cast<float2 *>(a.dbg + 1284 + 3 * 4096)[lane * 32 + e] = r0[e];
                reinterpret_cast<float2 *>(a.dbg + 1284 + 3 * 4096)[lane * 32 + 16 + e] = r1[e];
            }
        }

        SS_PRIOL(SS_P5_UN);
        SS_PH(7);  // radix-2 across the half-waves (permlane32 swaps + twiddles)
        // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
        // Lane (k1, h) register r0[i] holds bin k = k1 + 32 i + 512 h (< 1024); its partner 2048 - k is r1[15 - i] of lane
        // (32 - k1, 1 - h).  The k1 = 0 lanes pair with r1[16 - i] of the other k1 = 0 lane (they expose r1 shifted by one);
        // their i = 0 pairs are in-lane: lane 0 has k = 0 (X[0] and X[2048] come from Z[0] alone), lane 32 has
        // (512, 1536) = (r0[0], r1[0]).  Z[1024] = lane 0's r1[0] is the one bin left over.
        float esum = 0.f;
        float *pdst = prow + k1 + 512 * d;
        // All 16 untangle twiddles and all 32 partner values are requested here, before the stage's first LDS write: a read
        // may not move above an earlier write of the same wave, so twiddles fetched inside the loop below came one LDS round
        // trip per bin pair (16 dependent round trips per frame, a quarter of the wave's time with two waves per SIMD).
        // first P bin (low 16 bits) and filter index (high 16 bits, -1: none) of this lane's four slots, one register each
        // (requested here, with the untangle's table reads, for the mel stage behind it)
        int sf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) sf[s] = s_start[s * 64 + lane];
        auto st = [&](int s) { return sf[s] & 0xffff; };
        auto fi = [&](int s) { return FIXMEL ? static_cast<int>(static_cast<unsigned>(sf[s]) >> 16) : sf[s] >> 16; };  // (FIXMEL: never negative)
        // (LEAN: two batches of eight bin pairs, each with its own twiddle and partner fetches)
        constexpr int kUb = LEAN ? 8 : 16;
        if (lane < 3 && !LEAN) prow[1025 + lane] = 0.f;  // pad bins read (with zero weight) by the mel stage
#pragma unroll
        for (int hb = 0; hb < 16 / kUb; ++hb) {
            float2 twn[kUb];
#pragma unroll
            for (int q = 0; q < kUb; ++q) twn[q] = s_twn[(kUb * hb + q) * 64 + lane];
            float2 zcs[kUb];
#pragma unroll
            for (int q = 0; q < kUb; ++q) {
                const int i = kUb * hb + q;
                const float2 sv = k1z ? r1[(16 - i) & 15] : r1[15 - i];
                zcs[q] = (SS_ABL5 & 4) ? sv : make_float2(bperm(paddr, sv.x), bperm(paddr, sv.y));
            }
            __builtin_amdgcn_sched_barrier(0);  // all 32 fetches are in flight before the first bin pair is formed (the scheduler
                                                // otherwise sinks each next to its use: one exposed round trip per pair)
#pragma unroll
            for (int q = 0; q < kUb; ++q) {
                const int i = kUb * hb + q;
                const float2 zk = r0[i];
                float2 zc = zcs[q];
                if (i == 0) zc = k1z ? (d ? r1[0] : zk) : zc;
                const float2 w = twn[q];
                const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                const float2 dd = make_float2(zk.x - zc.x, zk.y + zc.y);
                // 2 X[k] = s - i w dd, 2 conj X[2048-k] = s + i w dd = 2 s - 2 X[k]
                const float xa_r = fmaf(w.y, dd.x, fmaf(w.x, dd.y, s.x));
                const float xa_i = fmaf(w.y, dd.y, fmaf(-w.x, dd.x, s.y));
                const float xb_r = fmaf(2.f, s.x, -xa_r), xb_i = fmaf(2.f, s.y, -xa_i);
                const float na = xa_r * xa_r + xa_i * xa_i, nb = xb_r * xb_r + xb_i * xb_i;
                const float pa = POW2 ? na : __builtin_amdgcn_sqrtf(na);
                const float pb = POW2 ? nb : __builtin_amdgcn_sqrtf(nb);
                pdst[32 * i] = pa;  // bins 0..1023 carry mel weight (the bank ends at (F+1)/2, feature.rs:69-70), as does 1024 below
                esum += pa + pb;

            }
        }
        if (lane < 3 && LEAN) prow[1025 + lane] = 0.f;  // pad bins read (with zero weight) by the mel stage
        if (lane == 0) {
            const float2 z = r1[0];  // X[1024] = conj Z[1024]
            const float n = 4.f * (z.x * z.x + z.y * z.y);
            const float p1024 = POW2 ? n : __builtin_amdgcn_sqrtf(n);
            prow[1024] = p1024;
            esum += p1024;
        }
        float energy = hscale32 * wave_sum_dpp(esum);      // E * 2^32 (see ln_scaled_h)
        energy = energy == 0.f ? kEps * kTwo32 : energy;  // zero_handling, feature.rs:219
        wave_order();
        SS_PRIOL(SS_P5_MEL);
        SS_PH(8);  // untangle + magnitudes + energy

        if (SS_ABL5 & 2) {
            if (lane < Cc) a.out[static_cast<unsigned long long>(frame) * Cc + lane] = energy;
            wave_order();
            frame = __builtin_amdgcn_readfirstlane(next_v);
            continue;
        }
        // ---- banded mel reduction (feature.rs:229), zero handling (:230), ln (:105) -> row in filter order ----
        {
            // the default bank shape of cfg5 (256 filters up to fs/2: 8 / 3 / 2 / 1 float4s per slot) has every tap count at
            // compile time: one LDS wait for the whole stage; other shapes take the run-time loops
            float mfix[4] = {0.f, 0.f, 0.f, 0.f};
            constexpr bool fixed8321 = FIXMEL;  // the launcher checked a.mel_q4 == {8, 3, 2, 1}
            if (fixed8321)
                mel4_fixed<8, 3, 2, 1>(w4, reinterpret_cast<const float4 *>(prow + st(0)), reinterpret_cast<const float4 *>(prow + st(1)),
                                       reinterpret_cast<const float4 *>(prow + st(2)), reinterpret_cast<const float4 *>(prow + st(3)), mfix);
            int off = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float m = hscale32 * (fixed8321 ? mfix[s] : mel_slot4(w4 + off, reinterpret_cast<const float4 *>(prow + st(s)), a.mel_q4[s]));
                m = m == 0.f ? kEps * kTwo32 : m;
                if (FIXMEL || fi(s) >= 0) {  // fewer than 256 filters: some (slot, lane) pairs own none (FIXMEL: 256 filters, checked by the launcher)
                    if (MFE) a.out[static_cast<unsigned long long>(frame) * a.n_filters + fi(s)] = m * (1.0f / kTwo32);  // exact: power of two
                    else frow[fi(s)] = ln_scaled(m);
                }
                off += a.mel_q4[s];
            }
        }
        if (MFE) {  // mfe (feature.rs:200-233): mel energies (written above) and the frame energy
            if (lane == 0) a.out_energy[frame] = energy * (1.0f / kTwo32);
            wave_order();
            frame = __builtin_amdgcn_readfirstlane(next_v);
            continue;
        }
        wave_order();
        SS_PH(9);  // mel + ln
#if SS_LAB
        if (!SS_PROF5 && a.dbg && frame == 0) {  // SS_DEBUG_ROWS (lab builds): frame 0's P row and ln(mel) row
            for (int i = lane; i < 1028; i += 64) a.dbg[i] = prow[i];
            for (int i = lane; i < 256; i += 64) a.dbg[1028 + i] = frow[i];
        }
#endif

        if (SS_ABL5 & 1) {
            if (lane < Cc) a.out[static_cast<unsigned long long>(frame) * Cc + lane] = frow[lane] + energy;
            wave_order();
            frame = __builtin_amdgcn_readfirstlane(next_v);
            continue;
        }
        // ---- DCT-II (feature.rs:120-123), folded twice.  cos(pi c (2(M-1-m)+1) / 2M) = (-1)^c cos(pi c (2m+1) / 2M): with
        // s[m] = L[m] + L[M-1-m] and d[m] = L[m] - L[M-1-m] an even coefficient is an M/2-term product with s, an odd one
        // with d.  For c = 2c' the same identity holds once more inside s (cos(pi c (2(M/2-1-m)+1) / 2M) =
        // (-1)^c' cos(pi c (2m+1) / 2M)): M/4 terms with s[m] + s[M/2-1-m] or s[m] - s[M/2-1-m].  Lane roles (host table,
        // ss_host.cpp build_4096): lanes 0 .. ne-1 the even coefficients, then two lanes per odd coefficient (d[0..63] and
        // d[64..127]); every lane has 64 terms -- 16 + 16 ds_read_b128 and 64 FMAs, against 64 and 128 with one lane per
        // coefficient.  Rows as four 64-value segments (s+, s-, d low, d high) one float4 apart: the four addresses a read
        // touches lie in different banks; the cosine rows are per lane at an odd float4 pitch. ----
        // (The stage's lane-dependent addresses are loop invariants: the compiler keeps them across the transform and spills
        // three of them to scratch -- 1.5 MB of extra writes per cfg5 launch.  Re-deriving them per frame from an opaque copy
        // of the lane number removes the spills and measured 1.1 us slower, so they stay.)
        if (FIXMEL || a.dct_fold2) {  // (the FIXMEL build is only launched with the twice-folded table: no generic DCT code in it)
            if (PF) {
                unsigned nf = min(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(next_v)), f_hi - 1);  // past the end: the last frame again, dropped
                const float *xn = a.x;
                if constexpr (MULTI) {
                    if (nf >= ns.u1) ns = seg_of(mt.m, nf);  // (uniform, rare: the claimed frame starts the next batch)
                    nf -= ns.u0;
                    xn = ns.x;
                }
                const unsigned nclip = nf / a.n_frames;
                const float2 *nsrc = reinterpret_cast<const float2 *>(xn + static_cast<unsigned long long>(nclip) * a.ld + (nf - nclip * a.n_frames) * a.step) + lane;
#pragma unroll
                for (int e = 0; e < 32; ++e) vpf[e] = nsrc[64 * e];
            }
            float *seg = wbase + kSRowOff;
            {
                const int m2 = lane + 64;
                const float l0 = lane < Mh ? frow[lane] : 0.f, l1 = lane < Mh ? frow[M - 1 - lane] : 0.f;
                const float l2 = m2 < Mh ? frow[m2] : 0.f, l3 = m2 < Mh ? frow[M - 1 - m2] : 0.f;
                const bool q = lane < (M >> 2);
                const float sa = q ? l0 + l1 : 0.f, sb = q ? frow[Mh - 1 - lane] + frow[Mh + lane] : 0.f;
                seg[lane] = sa + sb;
                seg[kSegPitch + lane] = sa - sb;
                seg[2 * kSegPitch + lane] = l0 - l1;
                seg[3 * kSegPitch + lane] = l2 - l3;
            }
            wave_order();
            const int ne = (Cc + 1) >> 1, nep = (ne + 1) & ~1;
            const int lp = lane - nep;
            const bool even = lane < nep;
            const int sk = even ? (min(lane, ne - 1) & 1) : 2 + (lp & 1);
            const float4 *r4 = reinterpret_cast<const float4 *>(seg + sk * kSegPitch);
            const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + lane * L::kCosLanePitch);
            float acc = 0.f;
            constexpr int kChunk = LEAN ? 8 : 16;  // float4 pairs fetched per LDS wait (16: one wait, 128 registers)
#pragma unroll
            for (int c0 = 0; c0 < 16; c0 += kChunk) {
                float4 rq[kChunk], cq[kChunk];
#pragma unroll
                for (int i = 0; i < kChunk; ++i) {
                    rq[i] = r4[c0 + i];
                    cq[i] = c4[c0 + i];
                }
#pragma unroll
                for (int i = 0; i < kChunk; ++i) {
                    acc = fmaf(rq[i].x, cq[i].x, acc);
                    acc = fmaf(rq[i].y, cq[i].y, acc);
                    acc = fmaf(rq[i].z, cq[i].z, acc);
                    acc = fmaf(rq[i].w, cq[i].w, acc);
                }
                if (kChunk < 16) __builtin_amdgcn_sched_barrier(0);
            }
            if (!even) acc += dpp<0xB1>(acc);  // quad_perm [1,0,3,2]: the coefficient's other half (nep is even)
            const bool whole = lane < ne || (!even && !(lp & 1) && lp < Cc - 1);  // the lanes that hold a whole coefficient
            if (PF) {
                // scaling + column-0 replacement (feature.rs:126-146); an unconditional, counted store (ss_wave.h)
                float o = acc * a.dct_scale_k;
                if (lane == 0) o = a.dc_elimination ? ln_scaled(energy) : acc * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                const unsigned frame_s = __builtin_amdgcn_readfirstlane(frame_b);
                buf_store(o, out_rsrc((MULTI ? cs.out : a.out) + static_cast<unsigned long long>(frame_s) * Cc, static_cast<unsigned>(Cc) * 4u),
                          whole ? (even ? 2 * lane : lp + 1) * 4 : kOobOffset);
            } else if (whole) {
                // scaling + column-0 replacement (feature.rs:126-146)
                float o = acc * a.dct_scale_k;
                if (lane == 0) o = a.dc_elimination ? ln_scaled(energy) : acc * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                (MULTI ? cs.out : a.out)[static_cast<unsigned long long>(frame_b) * Cc + (even ? 2 * lane : lp + 1)] = o;
            }
            wave_order();
            SS_PH(10);  // DCT + store
            SS_PRIOL(SS_P5_TOP);
            frame = __builtin_amdgcn_readfirstlane(next_v);
            continue;
        }
        if constexpr (FIXMEL) continue;  // (unreachable: the twice-folded path above always leaves the iteration)
        // Other shapes (n_filters not a multiple of 4, more than 43 coefficients): one fold, cosine rows [c][kCosPitch].
        // ---- DCT-II (feature.rs:120-123) with cos(pi c (2(M-1-m)+1) / 2M) = (-1)^c cos(pi c (2m+1) / 2M), M = 256:
        // the wave first forms s[m] = L[m] + L[255-m] and d[m] = L[m] - L[255-m] once (2 + 2 values per lane); an even
        // coefficient is then a 128-term product with s, an odd one with d -- half the FMAs and half the LDS reads.
        // The rows are kept as four 64-value segments (s low, s high, d low, d high) one float4 apart, so that the four
        // addresses one read of the product stage touches lie in different banks. ----
        {
            float *seg = wbase + kSRowOff;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int m = lane + 64 * h2;
                const bool in = m < Mh;  // M < 256: the rows are zero beyond M/2 (as are the cosine rows)
                const bool mid = (M & 1) && m == Mh;  // odd filter count: the middle filter pairs with itself
                const float lo = in || mid ? frow[m] : 0.f, hi = in ? frow[M - 1 - m] : 0.f;
                seg[h2 * kSegPitch + lane] = lo + hi;
                seg[(2 + h2) * kSegPitch + lane] = mid ? 0.f : lo - hi;
            }
        }
        wave_order();
        {
            // terms [first, first + 4 NQ) of coefficient c's 128-term product (first a multiple of 4 NQ <= 64)
            auto dct_part = [&](int c, int first, auto nq) {
                constexpr int NQ = decltype(nq)::value;
                const float4 *r4 = reinterpret_cast<const float4 *>(wbase + kSRowOff + ((c & 1) * 2 + (first >> 6)) * kSegPitch + (first & 63));
                const float4 *c4 = reinterpret_cast<const float4 *>(s_cos + c * L::kCosPitch + first);
                float4 rq[NQ], cq[NQ];
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    rq[i] = r4[i];
                    cq[i] = c4[i];
                }
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    acc = fmaf(rq[i].x, cq[i].x, acc);
                    acc = fmaf(rq[i].y, cq[i].y, acc);
                    acc = fmaf(rq[i].z, cq[i].z, acc);
                    acc = fmaf(rq[i].w, cq[i].w, acc);
                }
                return acc;
            };
            // All 64 lanes work: coefficient c < 32 is split over lanes c (terms 0..63) and c + 32 (terms 64..127); the
            // coefficients from 32 on take a second, shorter pass -- up to 8 of them in eighths over lanes (c & 7) + 8 part,
            // more in halves like the first pass.  (One lane per coefficient left 24 lanes idle and cost 128 FMAs and 64
            // ds_read_b128 per lane; this is 80 + 40 for 40 coefficients.)
            float acc = dct_part(min(lane & 31, Cc - 1), 64 * (lane >> 5), std::integral_constant<int, 16>{});
            acc += __shfl_xor(acc, 32, 64);
            if (Cc > 32) {
                float acc2;
                if (Cc <= 40) {
                    acc2 = dct_part(min(32 + (lane & 7), Cc - 1), 16 * (lane >> 3), std::integral_constant<int, 4>{});
                    acc2 += dpp<0x128>(acc2);  // row_ror:8: lane ^ 8
                    acc2 += __shfl_xor(acc2, 16, 64);
                    acc2 += __shfl_xor(acc2, 32, 64);
                } else {
                    acc2 = dct_part(min(32 + (lane & 31), Cc - 1), 64 * (lane >> 5), std::integral_constant<int, 16>{});
                    acc2 += __shfl_xor(acc2, 32, 64);
                }
                if (lane >= 32) acc = acc2;
            }
            if (lane < Cc) {
                // scaling + column-0 replacement (feature.rs:126-146)
                float o = acc * a.dct_scale_k;
                if (lane == 0) o = a.dc_elimination ? ln_scaled(energy) : acc * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                a.out[static_cast<unsigned long long>(frame) * Cc + lane] = o;
            }
        }
        wave_order();
        frame = __builtin_amdgcn_readfirstlane(next_v);
    }
    if (WAVES > 8) life_end(a.stamps, life, blockIdx.x * WAVES + wave);
#if SS_PROF5
    if (a.dbg && (threadIdx.x & 63) == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dbg) + 16ull * (blockIdx.x * WAVES + wave);
        pacc[11] = __builtin_amdgcn_s_memtime() - tstart;  // main loop lifetime
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = pacc[k];
    }
#endif
}

// ss_mel_c2048: the mel-spectrogram path (frame_analysis + |X wnorm|^2 + mel einsum, functions.rs:125-170, feature.rs:151-174)
// at fft_points = 4096 on the same FFT mapping: one wave owns one row (a 4096-sample window), the work unit is a (clip, row).
// Row r covers the 4096 samples that end at chunk r + n_pad: zero initial state, zero tail, rows past the real ones all zero
// (D3); windows inside the clip load at constant offsets, clip edges through one masked range per lane.  The table block is
// mfcc4096_layout without cosine rows (the mel rows start at kCos) and with the Vorbis window behind the mel rows.
template <int WAVES, bool STFT>
__global__ __launch_bounds__(WAVES * 64) void ss_mel_c2048(const Mel2048Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int k1 = lane & 31, d = lane >> 5;  // reader view: column k1, half a = d
    const int cls = lane & 1, bw = lane >> 1; // writer view: n1 = lane = cls + 2 bw

    float *wbase = reinterpret_cast<float *>(smem) + wave * kExFloats;
    float2 *ex = reinterpret_cast<float2 *>(wbase);
    float *prow = wbase;  // P[0..1024] + zero pad bins, after the exchange
    float *s_tab = reinterpret_cast<float *>(smem) + WAVES * kExFloats;
    const float4 *s_t1 = reinterpret_cast<const float4 *>(s_tab + L::kT1);
    const float2 *s_t2 = reinterpret_cast<const float2 *>(s_tab + L::kT2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + L::kTwn);
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const float *s_melw = s_tab + L::kCos;
    const float2 *s_win = reinterpret_cast<const float2 *>(s_tab + L::kCos + 64 * a.mel_wpitch);
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kCos + 64 * a.mel_wpitch + 4096);

    const int R = static_cast<int>(a.rows), Rreal = static_cast<int>(a.real_rows), M = static_cast<int>(a.n_filters);
    const unsigned total = a.batch * a.rows;
    const unsigned f_lo = static_cast<unsigned>(static_cast<unsigned long long>(total) * blockIdx.x / gridDim.x);
    const unsigned f_hi = static_cast<unsigned>(static_cast<unsigned long long>(total) * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kCos + 64 * a.mel_wpitch + 4096) / 4;
        for (int i = tid; i < n4; i += WAVES * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = f_lo + WAVES;
    }
    __syncthreads();
    int st[4], fi[4];  // first P bin and filter index (-1: none) of this lane's four slots (one packed table word each)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pk = s_start[s * 64 + lane];
        st[s] = pk & 0xffff;
        fi[s] = pk >> 16;
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(s_melw + lane * a.mel_wpitch);
    const int paddr = ((((32 - k1) & 31) | ((1 - d) << 5))) << 2;  // lane holding Z[2048 - k]
    float2 *exw = ex + cls * kClsStride + 34 * (bw >> 1) + (bw & 1);  // writer base (float2 units)
    const float2 *exr = ex + d * kClsStride + 2 * (k1 & 15);        // reader base
    const float hs = 0.25f * a.scale * a.scale;                      // |X wnorm|^2 = (wnorm^2 / 4) |2X|^2
    const bool k1z = k1 == 0;

    unsigned row = __builtin_amdgcn_readfirstlane(f_lo + wave);  // uniform: kept scalar
    while (row < f_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);

        const unsigned clip = row / a.rows;
        const int r = static_cast<int>(row - clip * a.rows);
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        // functions.rs:137-151: the window covers the last 4096 samples ending at chunk r + n_pad
        const int start = static_cast<int>(r + a.n_pad + 1) * static_cast<int>(a.hop) - 4096;
        const float2 *src = reinterpret_cast<const float2 *>(xc + start) + lane;
        float2 v[32];
        if (r < Rreal && start >= 0 && start + 4096 <= static_cast<int>(a.n_samples)) {  // uniform: one row per wave
#pragma unroll
            for (int e = 0; e < 32; ++e) v[e] = src[64 * e];
        } else {
            // clip edges and inactive rows: start and n_samples are even, the valid sample pairs form one range per lane
            // (the address of a masked load may lie outside the clip; it is never dereferenced)
            const int base = start + 2 * lane;
            const int n = static_cast<int>(a.n_samples);
            if (((start | n) & 1) == 0) {
                const int e_lo = base >= 0 ? 0 : (127 - base) >> 7;
                int e_hi = base >= n ? 0 : min(32, (n - base + 127) >> 7);
                if (r >= Rreal) e_hi = 0;
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    float2 s = make_float2(0.f, 0.f);
                    if (e >= e_lo && e < e_hi) s = src[64 * e];
                    v[e] = s;
                }
            } else {  // odd hop or clip length: a pair may straddle the clip edge, bounds per sample
                const bool act = r < Rreal;
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const int p0 = base + 128 * e;
                    v[e] = make_float2(act && p0 >= 0 && p0 < n ? xc[p0] : 0.f, act && p0 + 1 >= 0 && p0 + 1 < n ? xc[p0 + 1] : 0.f);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const float2 w = s_win[lane + 64 * e];
            v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
        }
        // ---- 2048-point complex FFT as in ss_mfcc_c2048 ----
        fft_reg<32>(v);
        float2 u[32];
#pragma unroll
        for (int k = 0; k < 16; ++k) exw[2 * k] = v[k];
        wave_order();
        if (k1 < 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
        }
        wave_order();
#pragma unroll
        for (int k = 0; k < 16; ++k) exw[2 * k] = v[16 + k];
        wave_order();
        if (k1 >= 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&exr[34 * p]);
                u[2 * p] = make_float2(t4.x, t4.y);
                u[2 * p + 1] = make_float2(t4.z, t4.w);
            }
        }
        wave_order();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float4 w2 = s_t1[p * 32 + k1];
            u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
            if (p < 15) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
        }
        fft_reg<32>(u);
        // (all sixteen twiddles are requested in front of the swaps: read inside the loop they came one iteration ahead of their
        // use -- less than an LDS round trip -- and every iteration waited: 3.4 k of a wave's 20 k cycles per frame)
        float2 t2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t2[i] = s_t2[i * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        float2 r0[16], r1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float px = u[i].x, qx = u[16 + i].x, py = u[i].y, qy = u[16 + i].y;
            swap_halves(px, qx);
            swap_halves(py, qy);
            const float2 wq = cmul(make_float2(qx, qy), t2[i]);
            r0[i] = make_float2(px + wq.x, py + wq.y);
            r1[i] = make_float2(px - wq.x, py - wq.y);
        }
        // ---- untangle Z -> X; (|X| wnorm)^2 of bins 0..1024 (functions.rs:166-169 + feature.rs:164) ----
        if (!STFT && lane < 3) prow[1025 + lane] = 0.f;  // pad bins read (with zero weight) by the mel stage
        float *pdst = prow + k1 + 512 * d;
        // stft build (functions.rs:86-123, :166-169): X[k] * wnorm for all 2049 bins of the row, interleaved re / im
        float2 *srow = STFT ? reinterpret_cast<float2 *>(a.out) + static_cast<unsigned long long>(row) * 2049ull : nullptr;
        const int kb = k1 + 512 * d;
        const float cs = 0.5f * a.scale;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float2 zcs[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = 8 * hb + q;
                const float2 sv = k1z ? r1[(16 - i) & 15] : r1[15 - i];
                zcs[q] = make_float2(bperm(paddr, sv.x), bperm(paddr, sv.y));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = 8 * hb + q;
                const float2 zk = r0[i];
                float2 zc = zcs[q];
                if (i == 0) zc = k1z ? (d ? r1[0] : zk) : zc;
                const float2 w = s_twn[i * 64 + lane];
                const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                const float2 dd = make_float2(zk.x - zc.x, zk.y + zc.y);
                const float xr = fmaf(w.y, dd.x, fmaf(w.x, dd.y, s.x));  // 2 X[k] = s - i w dd
                const float xi = fmaf(w.y, dd.y, fmaf(-w.x, dd.x, s.y));
                if (STFT) {
                    srow[kb + 32 * i] = make_float2(cs * xr, cs * xi);
                    // 2 conj X[2048 - k] = 2 s - 2 X[k]
                    srow[2048 - (kb + 32 * i)] = make_float2(cs * fmaf(2.f, s.x, -xr), -cs * fmaf(2.f, s.y, -xi));
                } else {
                    pdst[32 * i] = hs * fmaf(xr, xr, xi * xi);
                }
            }
        }
        if (lane == 0) {
            const float2 z = r1[0];  // X[1024] = conj Z[1024]
            if (STFT) srow[1024] = make_float2(a.scale * z.x, -a.scale * z.y);
            else prow[1024] = hs * 4.f * fmaf(z.x, z.x, z.y * z.y);
        }
        wave_order();
        if (STFT) {
            row = next;
            continue;
        }
        // ---- banded mel reduction (feature.rs:173) -> out[clip][m][r] ----
        {
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
        row = next;
    }
}

template <int WAVES>
hipError_t launch_h(const Mfcc4096Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = (static_cast<size_t>(WAVES) * kExFloats + (a.window ? 4096 : 0) + L::kCos + static_cast<size_t>(a.cos_floats) +
                        64 * static_cast<size_t>(a.mel_wpitch) + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    if (total >= 0xffffffffull) return hipErrorInvalidValue;
    unsigned long long blocks = (total + WAVES - 1) / WAVES;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(WAVES * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a, MultiArg<false>{});
        return hipGetLastError();
    };
    const bool pow2 = a.spectrum_exponent == 2, exact = a.flen == 4096 && a.preemph == 0.f;
    if constexpr (WAVES == 12) {
        // only the default cfg5 shape (exact frames, magnitude spectrum, 8/3/2/1 bank, twice-folded DCT, no window / mfe) has a
        // 12-wave build
        const bool lean_ok = exact && !pow2 && !a.window && !a.out_mfe && a.dct_fold2 && a.mel_q4[0] == 8 && a.mel_q4[1] == 3 &&
                             a.mel_q4[2] == 2 && a.mel_q4[3] == 1 && a.mel_wpitch == kMelPitch8321 && a.n_filters == 256;
        if (!lean_ok) return hipErrorInvalidValue;
        return go(ss_mfcc_c2048<true, false, 12, false, false, false, true>, "ss_mfcc_c2048<exact,mel8321,w12>");
    } else {
    if (a.preemph != 0.f) {  // fused pre-emphasis builds (run-time frame length)
#define SS_HP(P2, MF, WN, NAME) go(ss_mfcc_c2048<false, P2, WAVES, MF, WN, true>, NAME)
        if (a.window) {
            if (pow2) return a.out_mfe ? SS_HP(true, true, true, "ss_mfcc_c2048<pow2,mfe,win,pre>") : SS_HP(true, false, true, "ss_mfcc_c2048<pow2,win,pre>");
            return a.out_mfe ? SS_HP(false, true, true, "ss_mfcc_c2048<mfe,win,pre>") : SS_HP(false, false, true, "ss_mfcc_c2048<win,pre>");
        }
        if (pow2) return a.out_mfe ? SS_HP(true, true, false, "ss_mfcc_c2048<pow2,mfe,pre>") : SS_HP(true, false, false, "ss_mfcc_c2048<pow2,pre>");
        return a.out_mfe ? SS_HP(false, true, false, "ss_mfcc_c2048<mfe,pre>") : SS_HP(false, false, false, "ss_mfcc_c2048<pre>");
#undef SS_HP
    }
    if (a.window) {  // windowed builds: MFCC and mfe
        if (pow2) {
            if (a.out_mfe) return exact ? go(ss_mfcc_c2048<true, true, WAVES, true, true>, "ss_mfcc_c2048<exact,pow2,mfe,win>") : go(ss_mfcc_c2048<false, true, WAVES, true, true>, "ss_mfcc_c2048<pow2,mfe,win>");
            return exact ? go(ss_mfcc_c2048<true, true, WAVES, false, true>, "ss_mfcc_c2048<exact,pow2,win>") : go(ss_mfcc_c2048<false, true, WAVES, false, true>, "ss_mfcc_c2048<pow2,win>");
        }
        if (a.out_mfe) return exact ? go(ss_mfcc_c2048<true, false, WAVES, true, true>, "ss_mfcc_c2048<exact,mfe,win>") : go(ss_mfcc_c2048<false, false, WAVES, true, true>, "ss_mfcc_c2048<mfe,win>");
        return exact ? go(ss_mfcc_c2048<true, false, WAVES, false, true>, "ss_mfcc_c2048<exact,win>") : go(ss_mfcc_c2048<false, false, WAVES, false, true>, "ss_mfcc_c2048<win>");
    }
    if (a.out_mfe) {
        if (exact) return pow2 ? go(ss_mfcc_c2048<true, true, WAVES, true>, "ss_mfcc_c2048<exact,pow2,mfe>") : go(ss_mfcc_c2048<true, false, WAVES, true>, "ss_mfcc_c2048<exact,mfe>");
        return pow2 ? go(ss_mfcc_c2048<false, true, WAVES, true>, "ss_mfcc_c2048<pow2,mfe>") : go(ss_mfcc_c2048<false, false, WAVES, true>, "ss_mfcc_c2048<mfe>");
    }
    const bool m8321 = a.mel_q4[0] == 8 && a.mel_q4[1] == 3 && a.mel_q4[2] == 2 && a.mel_q4[3] == 1 && a.mel_wpitch == kMelPitch8321 && a.n_filters == 256;
    if (exact && m8321 && !pow2 && a.dct_fold2) return go(ss_mfcc_c2048<true, false, WAVES, false, false, false, true>, "ss_mfcc_c2048<exact,mel8321>");
    if (exact) return pow2 ? go(ss_mfcc_c2048<true, true, WAVES>, "ss_mfcc_c2048<exact,pow2>") : go(ss_mfcc_c2048<true, false, WAVES>, "ss_mfcc_c2048<exact>");
    return pow2 ? go(ss_mfcc_c2048<false, true, WAVES>, "ss_mfcc_c2048<pow2>") : go(ss_mfcc_c2048<false, false, WAVES>, "ss_mfcc_c2048");
    }
}

}  // namespace

hipError_t launch_mfcc_c2048_multi(const Mfcc4096Args &a_in, int n_batches, const float *const *d_x, float *const *d_out, const size_t *clips,
                                   hipStream_t stream, int num_cus, LaunchInfo *info)
{
    constexpr int WAVES = 12;
    Mfcc4096Args a = a_in;
    // the build that exists: the default cfg5 shape (see launch_h<12>)
    const bool lean_ok = a.flen == 4096 && a.preemph == 0.f && a.spectrum_exponent != 2 && !a.window && !a.out_mfe && a.dct_fold2 && a.mel_q4[0] == 8 &&
                         a.mel_q4[1] == 3 && a.mel_q4[2] == 2 && a.mel_q4[3] == 1 && a.mel_wpitch == kMelPitch8321 && a.n_filters == 256;
    if (n_batches < 1 || n_batches > kMaxLaunchBatches || !lean_ok) return hipErrorInvalidValue;
    const size_t lds = (static_cast<size_t>(WAVES) * kExFloats + L::kCos + static_cast<size_t>(a.cos_floats) + 64 * static_cast<size_t>(a.mel_wpitch) + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    BatchTable m{};
    unsigned long long total = 0;
    for (int b = 0; b < kMaxLaunchBatches; ++b) {
        m.uend[b] = 0xffffffffu;
        if (b >= n_batches) continue;
        const unsigned long long tot = static_cast<unsigned long long>(clips[b]) * a.n_frames;
        if (tot == 0) return hipErrorInvalidValue;  // (empty batches are dropped by the caller)
        total += tot;
        if (total >= 0xffffffffull) return hipErrorInvalidValue;
        m.x[b] = d_x[b];
        m.out[b] = d_out[b];
        m.uend[b] = static_cast<uint32_t>(total);
        m.total[b] = static_cast<uint32_t>(tot);
    }
    a.x = d_x[0];
    a.out = d_out[0];
    a.batch = static_cast<uint32_t>(total);  // MULTI: the launch's frame count (the kernel takes the batches from the table)
    unsigned long long blocks = (total + WAVES - 1) / WAVES;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    auto kern = ss_mfcc_c2048<true, false, 12, false, false, false, true, true>;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    if (info) *info = LaunchInfo{"ss_mfcc_c2048m<exact,mel8321,w12>", grid, static_cast<unsigned>(WAVES * 64), lds};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a, MultiArg<true>{m});
    return hipGetLastError();
}

hipError_t launch_mfcc_c2048(const Mfcc4096Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    // the default cfg5 shape has a build with three waves per SIMD (12 per CU, <= 168 VGPRs): 63.4 us against 72.1 for the
    // 8-wave build on one box; everything else (and that shape where 12 regions do not fit the LDS) takes the 8-wave builds
    {
        const hipError_t e = launch_h<12>(a, stream, num_cus, info);
        if (e != hipErrorInvalidValue) return e;
    }
    return launch_h<8>(a, stream, num_cus, info);
}

hipError_t launch_mel_c2048(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    constexpr int WAVES = 8;
    const size_t lds = (static_cast<size_t>(WAVES) * kExFloats + L::kCos + 64 * static_cast<size_t>(a.mel_wpitch) + 4096 + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.rows;
    if (total == 0) return hipSuccess;
    if (total >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned cap = static_cast<unsigned>(num_cus > 0 ? num_cus : 256);
    const unsigned long long blocks = (total + WAVES - 1) / WAVES;
    const unsigned grid = static_cast<unsigned>(blocks < cap ? blocks : cap);
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(WAVES * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
        return hipGetLastError();
    };
    return a.out_stft ? go(ss_mel_c2048<WAVES, true>, "ss_mel_c2048<stft>") : go(ss_mel_c2048<WAVES, false>, "ss_mel_c2048");
}

}  // namespace ss

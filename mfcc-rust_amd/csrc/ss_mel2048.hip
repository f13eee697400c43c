// ss_mel_c1024: fused mel spectrogram for fft_points = 2048 (C = 1024 packed complex points) on gfx950 --
// the STFT branch of the reference: frame_analysis / stft2 (functions.rs:86-170) -> |X|^2 (feature.rs:164) ->
// mel bank (feature.rs:173), output [clip][n_mels][rows].
//
// Same structure as the 512-point MFCC kernel (ss_mfcc512.hip), one size up:
//   * 32 lanes own a frame, 32 complex points per lane; a wave carries 2 frames = two consecutive output rows of
//     one clip.  One persistent 8-wave workgroup per CU; waves pull (clip, row pair) units from an LDS counter.
//   * window (Vorbis, config.rs:151-160) applied on load; the window covers the last W samples ending at chunk
//     r + n_pad (zero outside the clip: zero initial state per clip), functions.rs:137-151.
//   * 1024-point FFT = two radix-32 register butterflies with ONE transposing exchange through wave-private LDS, run in
//     two register halves (ds_write_b64 scatter to 34*(n1>>1) + 2*k1' + (n1&1), 16 ds_read_b128 back; both
//     conflict-free) so that input and output registers of the transpose never coexist in full.
//   * untangle with ds_bpermute_b32 (partner = lane 32-j, register 31-r); only bins 0..512 are produced: the mel
//     bank ends at bin (F+1)/2 (feature.rs:69-70) and this path has no frame energy.
//   * (|X| wnorm)^2 (functions.rs:166-169, feature.rs:164) -> P row in LDS -> banded mel reduction, 4 filters per lane.
//   * output [clip][m][r]: each lane stores its four mel values straight to out[clip][m][r]; the wave's two rows are adjacent
//     words, so the stores are 8-byte pieces of lines whose other rows come from other waves of the same CU within a few
//     microseconds and merge in L2 only partly (HBM writes ~1.4x the output = traffic 1.09x the algorithmic bytes).  The CU-wide
//     whole-line tile that removed this (writes 1.00x, same duration) lives in tools/experiments/ss_mel2048_tile.hip since round 4:
//     it does not fit beside twelve waves, and no BASELINE shape selected it any more.  No workgroup barrier anywhere in the main
//     loop (a barrier-synchronised transposing tile measured 20 % slower in round 1).
//   * stft output [clip][row][1025] complex64: see the note on write-dominated streams in DESIGN.md (the 1024-clip shape's
//     269 MB output sits on the edge of the 256 MiB Infinity Cache).
// Rows >= real_rows (the trailing n_pad rows the reference never writes, functions.rs:121) come out as exact zeros.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"
#include "ss_wave.h"

#include <cstdlib>

namespace ss {

namespace {

using namespace wv;

// Issue priorities of the twelve-wave kernel's phases (s_setprio; the SIMD's arbiter takes the highest priority first, the oldest
// wave among equals).  The butterflies are pure VALU streams and run at the lowest priority: a wave that is about to request
// samples, exchange through LDS or read tables gets its requests out in front of them, and their round trips pass while the
// butterflies of the other waves fill the SIMD.  Measured on cfg3 (same box, profiles/r03/ab_cfg3_priorities.txt): 47.2 us with
// no priorities, 45.6 with 3 / 1 / 0 (loop top / every other phase / butterflies), 46.7 with the loop top alone raised,
// 46.8 with the butterflies raised instead.  SS_PRIOS (lab builds): five decimal digits, priority at the loop top (sample
// request, window), exchange, twiddles, untangle, mel + stores.
#if SS_LAB && defined(SS_PRIOS)
#define SS_P_TOP ((SS_PRIOS / 10000) % 10)
#define SS_P_EX ((SS_PRIOS / 1000) % 10)
#define SS_P_TW ((SS_PRIOS / 100) % 10)
#define SS_P_UN ((SS_PRIOS / 10) % 10)
#define SS_P_MEL (SS_PRIOS % 10)
#else
#define SS_P_TOP 3
#define SS_P_EX 1
#define SS_P_TW 1
#define SS_P_UN 1
#define SS_P_MEL 1
#endif
#define SS_P_FFT 0
#define SS_PRIOL(x) __builtin_amdgcn_s_setprio(x)
// Measured and retired (rounds 4 / 5; records profiles/r04, profiles/r05, code tools/experiments/ss_mel2048_lab_r05.diff): fair shares
// between the three waves of a SIMD by rotating a priority step (equal shares at the slowest wave's speed, +9 %); no unit for the
// all-zero row pairs (6.6 % fewer instructions, +0.7 - 1.5 us: those units were free filler under the slow waves' last ones); a
// cross-workgroup pool for the launch's last eighth of units (44.2 against 44.1 us: the end of a launch is one UNIT long, not one
// slow CU long); work items of four rows with 16-byte output pieces (49.3 against 46.8 us: coarser items spread worse over twelve
// waves of different speeds); non-temporal sample loads / output stores.
namespace L = mel2048_layout;
constexpr int kExSlots = 2 * 16 * 34;        // float2 in the wave's exchange region: two frames x half the columns (8704 B)
constexpr int kWaveFloatsM = kExSlots * 2;  // one exchange region; the two P rows (2 x 520 floats) reuse it after the exchange

// Row pairs of a clip a mel build spends a unit on: every pair, the trailing all-zero ones (functions.rs:121) included.
__host__ __device__ inline unsigned mel_work_pairs(unsigned rows, unsigned /*real_rows*/) { return (rows + 1) / 2; }

template <int kWavesM, bool STFT, bool FULLP = false>
__global__ __launch_bounds__(kWavesM * 64) void ss_mel_c1024(const Mel2048Args a)
{
    constexpr bool PREFETCH_M = kWavesM <= 8;  // the next unit's samples are requested while the current one is in its second pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int half = lane >> 5;  // frame within the wave
    const int j = lane & 31;     // lane within the frame

    // ---- LDS carve: per-wave regions | table block | unit counter ----
    float *wbase = reinterpret_cast<float *>(smem) + wave * kWaveFloatsM;
    float2 *ex = reinterpret_cast<float2 *>(wbase);                 // exchange region (one frame at a time)
    // P[0..512] + zero pad bins, after the exchange; all 1025 bins when the bank reaches past (F+1)/2 (two rows of 1028 still
    // fit the region)
    float *prow = wbase + half * (FULLP ? 1088 : L::kPRow);
    float *s_tab = reinterpret_cast<float *>(smem) + kWavesM * kWaveFloatsM;
    const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2 + j * L::kTw2Pitch);  // this lane's 16 twiddle pairs
    const float4 *s_twn4 = reinterpret_cast<const float4 *>(s_tab + L::kTwn + j * L::kTwnPitch);  // this lane's 16 untangle twiddles, two per read
    const float4 *s_win4 = reinterpret_cast<const float4 *>(s_tab + L::kWin + j * L::kWinPitch);  // this lane's 32 window pairs, two per read
    const int *s_start = reinterpret_cast<const int *>(s_tab + L::kStart);
    const int *s_filt = reinterpret_cast<const int *>(s_tab + L::kFilt);
    const float *s_melw = s_tab + L::kMelW;
    // behind the table block: 4 words copied with it, then the unit counter
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 32 * a.mel_wpitch + 4);

    {
        const int n4 = (L::kMelW + 32 * a.mel_wpitch + 4) / 4;
        for (int i = tid; i < n4; i += kWavesM * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) {
            const unsigned long long units0 = static_cast<unsigned long long>(a.batch) * (STFT ? (a.rows + 1) / 2 : mel_work_pairs(a.rows, a.real_rows));
            *s_next = static_cast<unsigned>(units0 * blockIdx.x / gridDim.x) + kWavesM;
        }
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
    const float hs = 0.25f * a.scale * a.scale;              // |X wnorm|^2 = (wnorm^2 / 4) |2X|^2
    const int R = static_cast<int>(a.rows), Rreal = static_cast<int>(a.real_rows);
    const int M = static_cast<int>(a.n_filters);

    // work unit: two consecutive rows of one clip; the workgroup owns a contiguous range of units and its waves
    // pull them from an LDS counter
    // (mel builds: only the pairs that hold a real row are units; the wave that owns a clip's last one writes the zeros of the rows
    // behind it -- see ss_mel_c1024_w12)
    const unsigned pairs = STFT ? (a.rows + 1) / 2 : mel_work_pairs(a.rows, a.real_rows);
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * pairs;
    const unsigned u_lo = static_cast<unsigned>(units * blockIdx.x / gridDim.x);
    const unsigned u_hi = static_cast<unsigned>(units * (blockIdx.x + 1) / gridDim.x);
    // (clip, row) of this half-wave within a unit, and the loads of its window: functions.rs:137-151, the window covers the
    // last W samples ending at chunk r + n_pad
    auto load_unit = [&](unsigned un, float2 (&vv)[32]) {
        const unsigned clip = un / pairs;
        const int r = static_cast<int>(un - clip * pairs) * 2 + half;
        const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
        const bool active = r < Rreal;
        const int start = static_cast<int>(r + a.n_pad + 1) * static_cast<int>(a.hop) - 2048;
        const bool inside = active && start >= 0 && start + 2048 <= static_cast<int>(a.n_samples);
        const float2 *src = reinterpret_cast<const float2 *>(xc + start) + j;
        if (__all(inside)) {
            // both windows of the pair inside the clip: 8-byte loads at constant offsets from one base
#pragma unroll
            for (int e = 0; e < 32; ++e) vv[e] = src[32 * e];
        } else {
            // clip edges (zero initial state, zero padding of the last chunk, D3) and inactive rows.  start and
            // n_samples are even here, so a sample pair is inside or outside as a whole, and because the pair index
            // grows with e the valid ones form one range [e_lo, e_hi) per lane: loads outside it are masked off
            // (the address may lie before the clip; it is never dereferenced) and read as zero.
            const int base = start + 2 * j;
            const int n = static_cast<int>(a.n_samples);
            if (((start | n) & 1) == 0) {
                int e_lo = base >= 0 ? 0 : (63 - base) >> 6;
                int e_hi = base >= n ? 0 : min(32, (n - base + 63) >> 6);
                if (!active) e_hi = 0;
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    float2 s = make_float2(0.f, 0.f);
                    if (e >= e_lo && e < e_hi) s = src[32 * e];
                    vv[e] = s;
                }
            } else {  // odd hop or clip length: a pair may straddle the clip edge, bounds per sample
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const int p0 = base + 64 * e;
                    vv[e] = make_float2(active && p0 >= 0 && p0 < n ? xc[p0] : 0.f, active && p0 + 1 >= 0 && p0 + 1 < n ? xc[p0 + 1] : 0.f);
                }
            }
        }
    };

    unsigned unit = __builtin_amdgcn_readfirstlane(u_lo + wave);  // uniform: kept scalar
    float2 v[32];
    if (unit < u_hi) load_unit(unit, v);
    if (STFT) {
        // the first unit's samples have no stores behind them: as many dropped stores as a unit issues, so that both ways into
        // the loop look alike to the compiler's wait counting (see ss_mfcc512.hip)
        const __amdgpu_buffer_rsrc_t none = out_rsrc(a.out, 0u);
#pragma unroll
        for (int k = 0; k < 33; ++k) buf_store(make_float2(0.f, 0.f), none, 64 * k);
    }
    while (unit < u_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);
        {
            {
                const unsigned clip = unit / pairs;
                const int r = static_cast<int>(unit - clip * pairs) * 2 + half;
                const bool in_rows = r < R;
#pragma unroll
                for (int e = 0; e < 32; e += 2) {
                    const float4 w = s_win4[e >> 1];
                    v[e] = make_float2(v[e].x * w.x, v[e].y * w.y);
                    v[e + 1] = make_float2(v[e + 1].x * w.z, v[e + 1].y * w.w);
                }
                // ---- 1024-point complex FFT: radix-32, transpose through LDS (one frame at a time), twiddle, radix-32 ----
                fft_reg<32>(v);
                // The transpose runs in two register halves (columns k1 < 16, then k1 >= 16) so that only 16 of v's 32
                // registers are live while u is being filled: both frames' half-columns fit one 8704-B region.
                float2 u[32];
                float2 *exf = ex + half * (16 * 34);  // this frame's slice: [n1 pair 16][k1' 16][parity 2] + 2 pad per pair
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
                // The pass-2 twiddles are all requested here, right behind the exchange's reads and in front of the first product:
                // read where they are used they came a pair at a time, each one exposed LDS round trip (with two waves per SIMD
                // nobody hides it).
                // the window registers are dead now: the next unit's samples load into them while this one is finished
                if (PREFETCH_M && next < u_hi) load_unit(next, v);
                // (Requesting the sixteen twiddle pairs in one or two batches in front of the products, or reading the unit claim
                // late, measured within +-0.3 us here -- unlike in the 4096-point kernel -- and perturbs this kernel's register
                // allocation, which is at the SGPR limit: left as the compiler schedules it.)
#pragma unroll
                for (int p = 0; p < 16; ++p) {  // two twiddles per ds_read_b128: W^(j(2p+1)), W^(j(2p+2))
                    const float4 w2 = s_tw2[p];
                    u[2 * p + 1] = cmul(u[2 * p + 1], make_float2(w2.x, w2.y));
                    if (p < 15) u[2 * p + 2] = cmul(u[2 * p + 2], make_float2(w2.z, w2.w));
                }
                fft_reg<32>(u);  // u[r] = Z[j + 32 r]

                // ---- untangle: k = j + 32 r, r < 16 (and its mirror 1024 - k for the stft output), and k = 512 ----
                // stft output (functions.rs:86-123, :166-169): X[k] * wnorm for all 1025 bins of the row, interleaved re, im;
                // lanes of a half-wave write 256 contiguous bytes per register on both sides of the spectrum
                // (counted stores, ss_wave.h: one descriptor over the pair's rows that exist -- a clip's last pair may have one --
                // so that the next unit's prefetched samples are waited for with these 33 stores still in flight)
                __amdgpu_buffer_rsrc_t srs = out_rsrc(nullptr, 0u);
                if (STFT) {
                    const unsigned unit_s = __builtin_amdgcn_readfirstlane(unit);
                    const unsigned clip_s = unit_s / pairs;
                    const unsigned r0 = (unit_s - clip_s * pairs) * 2;
                    srs = out_rsrc(a.out + (static_cast<unsigned long long>(clip_s) * R + r0) * 2050ull, min(2u, static_cast<unsigned>(R) - r0) * 8200u);
                }
                const int srow_off = half * 8200;
                const float cs = 0.5f * a.scale;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {  // two batches of 8: all partner fetches of a batch go out before its arithmetic
                    float2 zcs[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) zcs[q] = make_float2(bperm(paddr, u[31 - (8 * hb + q)].x), bperm(paddr, u[31 - (8 * hb + q)].y));
                    float4 tw4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) tw4[i] = s_twn4[4 * hb + i];
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int q = 8 * hb + qq;
                        const float2 zk = u[q];
                        // lane 0 pairs with itself: Z[1024 - 32 q] = own register (32 - q) & 31
                        const float2 zc = j == 0 ? u[(32 - q) & 31] : zcs[qq];
                        const float2 w = (qq & 1) ? make_float2(tw4[qq >> 1].z, tw4[qq >> 1].w) : make_float2(tw4[qq >> 1].x, tw4[qq >> 1].y);
                        const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                        const float2 d = make_float2(zk.x - zc.x, zk.y + zc.y);
                        // 2 X[k] = s - i w d: two chained FMAs per component
                        const float xr = fmaf(w.y, d.x, fmaf(w.x, d.y, s.x));
                        const float xi = fmaf(w.y, d.y, fmaf(-w.x, d.x, s.y));
                        if (STFT) {
                            buf_store(make_float2(cs * xr, cs * xi), srs, srow_off + (j + 32 * q) * 8);
                            // 2 conj X[1024 - k] = 2 s - 2 X[k]
                            buf_store(make_float2(cs * fmaf(2.f, s.x, -xr), -cs * fmaf(2.f, s.y, -xi)), srs, srow_off + (1024 - j - 32 * q) * 8);
                        } else {
                            prow[j + 32 * q] = hs * (xr * xr + xi * xi);   // (|X| wnorm)^2, functions.rs:166-169 + feature.rs:164
                            if (FULLP) {  // bins 513..1024 as well
                                const float yr = fmaf(2.f, s.x, -xr), yi = fmaf(2.f, s.y, -xi);
                                prow[1024 - j - 32 * q] = hs * (yr * yr + yi * yi);
                            }
                        }
                    }
                }
                if (!STFT && j == 0) {
                    const float2 z = u[16];  // X[512] = conj Z[512]
                    prow[512] = hs * 4.f * (z.x * z.x + z.y * z.y);
                }
                if (STFT) {
                    buf_store(make_float2(a.scale * u[16].x, -a.scale * u[16].y), srs, j == 0 ? srow_off + 512 * 8 : kOobOffset);  // X[512] = conj Z[512]
                    wave_order();
                    if (!PREFETCH_M && next < u_hi) load_unit(next, v);
                    unit = next;
                    continue;
                }
                if (j < 3) prow[(FULLP ? 1025 : 513) + j] = 0.f;  // pad bins read (with zero weight) by the mel stage
                wave_order();
                // ---- banded mel reduction (feature.rs:173), four filters per lane; the two rows of the wave are
                //      adjacent words of out[clip][m][.] ----
                if (in_rows) {
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
            }
        }
        if (!PREFETCH_M && next < u_hi) load_unit(next, v);
        unit = next;
    }
}

// ss_mel_c1024_w12: the same mel-spectrogram path with THREE waves per SIMD (12 per CU, <= 168 VGPRs) for the reference bank
// shape (P rows of bins 0..512), direct stores.  What makes 168 registers enough: no unit is prefetched across the second pass
// (the third wave on the SIMD hides the load latency instead), every table read comes in batches of at most eight float4s,
// what depends on the lane number only is derived again in every iteration (the lane number is made opaque at the top of the
// loop) instead of living in ~40 registers for the whole kernel, and u is defined on every lane before the two half-wave
// phases of the exchange fill it (left half-defined, the "undefined" halves are carried around the loop and spilled).
// Twelve exchange regions + the tables are 134 KB of LDS, so the CU-wide whole-line tile of the 8-wave build (55 KB) does not
// fit beside them: the rows leave as 8-byte pieces of lines (HBM writes 1.4x the output, traffic 1.09x the algorithmic bytes).
// MULTI (ss_mel_spectrogram_batches_device): the launch's units are the concatenation of up to kMaxLaunchBatches blocks' row
// pairs, each block with its own input and output (BatchTable, ss_device.h; Seg / seg_of, ss_wave.h).
template <bool FIXMEL, bool STFT = false, bool MULTI = false>
__global__ __launch_bounds__(12 * 64, 3) void ss_mel_c1024_w12(const Mel2048Args a, const MultiArg<MULTI> mt)
{
    static_assert(!(MULTI && STFT), "the batch-table build is a mel-output build");
    const LifeStamp life = life_begin(a.stamps);  // (diagnostic: null in every ordinary launch)
    constexpr int kWavesM = 12;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    float *s_tab = reinterpret_cast<float *>(smem) + kWavesM * kWaveFloatsM;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 32 * a.mel_wpitch + 4);
    // Row pairs per clip that are units: all of them (see mel_work_pairs for the measured alternative)
    const unsigned pairs = STFT ? (a.rows + 1) / 2 : mel_work_pairs(a.rows, a.real_rows);
    unsigned long long units = static_cast<unsigned long long>(a.batch) * pairs;
    if constexpr (MULTI) units = a.batch;  // (the launcher hands over the launch's unit count: the sum over the blocks)
    // Work distribution: the workgroup owns a contiguous range of units (neighbouring units share three quarters of their samples:
    // L1 / L2 locality), its waves pull them from an LDS counter.
    const unsigned u_lo = static_cast<unsigned>(units * blockIdx.x / gridDim.x);
    const unsigned u_hi = static_cast<unsigned>(units * (blockIdx.x + 1) / gridDim.x);
    {
        const int n4 = (L::kMelW + 32 * a.mel_wpitch + 4) / 4;
        for (int i = tid; i < n4; i += kWavesM * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = u_lo + kWavesM;
    }
    __syncthreads();
    const float hs = 0.25f * a.scale * a.scale;  // |X wnorm|^2 = (wnorm^2 / 4) |2X|^2
    const int R = static_cast<int>(a.rows), Rreal = static_cast<int>(a.real_rows);
    const int M = static_cast<int>(a.n_filters);

#if SS_LAB && defined(SS_PROF3)
    // unit timeline (lab builds, tools/prof3.py): lane 0 of every wave stamps the 100 MHz clock at the top of a unit, once its samples
    // have arrived and at its end, into 32 words per wave BEHIND the output block (the tool allocates them); word 0 = units | XCC << 32
    unsigned long long *p3 = reinterpret_cast<unsigned long long *>(a.out + static_cast<unsigned long long>(a.batch) * a.n_filters * a.rows) +
                             32ull * (blockIdx.x * kWavesM + wave);
    unsigned p3n = 0;
#define SS_P3(k) do { if ((threadIdx.x & 63) == 0 && p3n < 10) p3[1 + 3 * p3n + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SS_P3(k) do { } while (0)
#endif
    unsigned item = u_lo + wave;
    Seg cs{};  // MULTI: the block of the current unit
    if constexpr (MULTI) cs = seg_of(mt.m, min(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(item)), u_hi - 1));
    SS_PRIOL(SS_P_TOP);
    while (item < u_hi) {
        // the claim of the next item is issued here and read at the end of the iteration
        unsigned next_v = 0;
        if ((static_cast<int>(threadIdx.x) & 63) == 0) next_v = atomicAdd(s_next, 1u);
        unsigned unit = item;
        const float *x_b = a.x;  // the block this unit belongs to: its input, its output and the unit's index within it
        float *out_b = a.out;
        if constexpr (MULTI) {
            const unsigned item_s = __builtin_amdgcn_readfirstlane(item);
            if (item_s >= cs.u1) cs = seg_of(mt.m, item_s);  // (uniform, rare: the claimed unit starts the next block)
            unit = item_s - cs.u0;
            x_b = cs.x;
            out_b = cs.out;
        }
        int lane_it = static_cast<int>(threadIdx.x) & 63;
        asm volatile("" : "+v"(lane_it));  // see above: nothing derived from the lane number is hoisted out of the loop
        const int lane = lane_it & 63;     // (the mask tells the compiler the range again: 24-bit multiplies, no sign extensions)
        const int half = lane >> 5;  // frame within the wave
        const int j = lane & 31;     // lane within the frame
        float *wbase = reinterpret_cast<float *>(smem) + wave * kWaveFloatsM;
        float *prow = wbase + half * L::kPRow;
        const float4 *s_tw2 = reinterpret_cast<const float4 *>(s_tab + L::kTw2 + j * L::kTw2Pitch);
        const float4 *s_twn4 = reinterpret_cast<const float4 *>(s_tab + L::kTwn + j * L::kTwnPitch);
        const float4 *s_win4 = reinterpret_cast<const float4 *>(s_tab + L::kWin + j * L::kWinPitch);
        const unsigned clip = unit / pairs;
        const int r = static_cast<int>(unit - clip * pairs) * 2 + half;
        const bool in_rows = r < R;
        SS_P3(0);
        // ---- the window of this half-wave's row (functions.rs:137-151: the last W samples ending at chunk r + n_pad) ----
        float2 v[32];
        {
            const float *xc = x_b + static_cast<unsigned long long>(clip) * a.ld;
            const bool active = r < Rreal;
            const int start = static_cast<int>(r + a.n_pad + 1) * static_cast<int>(a.hop) - 2048;
            const bool inside = active && start >= 0 && start + 2048 <= static_cast<int>(a.n_samples);
            const float2 *src = reinterpret_cast<const float2 *>(xc + start) + j;
            if (__all(inside)) {
                // (uniform base -- the unit's first row -- + this lane's 32-bit byte offset: loads in the SGPR-base form)
                const unsigned unit_s = __builtin_amdgcn_readfirstlane(unit);
                const unsigned clip_s = unit_s / pairs;
                const long long start0 = static_cast<long long>((unit_s - clip_s * pairs) * 2 + a.n_pad + 1) * static_cast<long long>(a.hop) - 2048;
                const char *sb = reinterpret_cast<const char *>(x_b + static_cast<unsigned long long>(clip_s) * a.ld) + start0 * 4;
                const unsigned so = static_cast<unsigned>(half) * a.hop * 4u + static_cast<unsigned>(j) * 8u;
                unsigned so2[2] = {so, so + 4096u};  // (a 32-bit lane offset per 4096 bytes, pinned: left alone the second half's addresses become 64-bit VALU sums)
                asm volatile("" : "+v"(so2[1]));
#pragma unroll
                for (int e = 0; e < 32; ++e) v[e] = *reinterpret_cast<const float2 *>(sb + so2[e / 16] + 256u * (e % 16));
            } else {
                // clip edges (zero initial state, zero padding of the last chunk, D3) and inactive rows: see ss_mel_c1024
                const int base = start + 2 * j;
                const int n = static_cast<int>(a.n_samples);
                if (((start | n) & 1) == 0) {
                    int e_lo = base >= 0 ? 0 : (63 - base) >> 6;
                    int e_hi = base >= n ? 0 : min(32, (n - base + 63) >> 6);
                    if (!active) e_hi = 0;
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        float2 sv = make_float2(0.f, 0.f);
                        if (e >= e_lo && e < e_hi) sv = src[32 * e];
                        v[e] = sv;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int p0 = base + 64 * e;
                        v[e] = make_float2(active && p0 >= 0 && p0 < n ? xc[p0] : 0.f, active && p0 + 1 >= 0 && p0 + 1 < n ? xc[p0 + 1] : 0.f);
                    }
                }
            }
        }
#if SS_LAB && defined(SS_PROF3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SS_P3(1);
#endif
        // Vorbis window (config.rs:151-160): two batches of eight reads, each in front of its products
#pragma unroll
        for (int eb = 0; eb < 16; eb += 8) {
            float4 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = s_win4[eb + i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 2 * (eb + i);
                v[e] = make_float2(v[e].x * w[i].x, v[e].y * w[i].y);
                v[e + 1] = make_float2(v[e + 1].x * w[i].z, v[e + 1].y * w[i].w);
            }
        }
        // ---- 1024-point complex FFT: radix-32, transpose through LDS in two register halves, twiddle, radix-32 ----
        SS_PRIOL(SS_P_FFT);
        fft_reg<32>(v);
        SS_PRIOL(SS_P_EX);
        float2 u[32];
        {
            float2 *exf = reinterpret_cast<float2 *>(wbase) + half * (16 * 34);
            const int wbh = 34 * (j >> 1) + (j & 1);
            const int jl = j & 15;
#pragma unroll
            for (int k = 0; k < 16; ++k) exf[wbh + 2 * k] = v[k];
            wave_order();
            // every lane fills u in ONE of the two half-wave phases: "defined" here (an empty asm per register, no instruction)
#pragma unroll
            for (int k = 0; k < 32; ++k) u[k] = make_float2(defined_garbage(), defined_garbage());
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
        }
        SS_PRIOL(SS_P_TW);
#pragma unroll
        for (int pb = 0; pb < 16; pb += 8) {  // pass-2 twiddles, two per ds_read_b128, eight reads per batch
            float4 tw2[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) tw2[p] = s_tw2[pb + p];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                u[2 * (pb + p) + 1] = cmul(u[2 * (pb + p) + 1], make_float2(tw2[p].x, tw2[p].y));
                if (pb + p < 15) u[2 * (pb + p) + 2] = cmul(u[2 * (pb + p) + 2], make_float2(tw2[p].z, tw2[p].w));
            }
        }
        SS_PRIOL(SS_P_FFT);
        fft_reg<32>(u);  // u[q] = Z[j + 32 q]
        SS_PRIOL(SS_P_UN);

        // ---- untangle bins k = j + 32 q, q < 16, and k = 512; (|X| wnorm)^2 -> P row (functions.rs:166-169, feature.rs:164) ----
        const int paddr = ((lane & 32) | ((32 - j) & 31)) << 2;  // lane holding Z[1024 - k]
        // first P bin and filter index of this lane's four slots: requested here, used by the mel stage below
        int st[4], fi[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            st[s] = STFT ? 0 : reinterpret_cast<const int *>(s_tab + L::kStart)[s * 32 + j];
            fi[s] = STFT ? 0 : reinterpret_cast<const int *>(s_tab + L::kFilt)[s * 32 + j];
        }
        // stft output (functions.rs:86-123, :166-169): X[k] * wnorm for all 1025 bins of the row, interleaved re, im, through a
        // descriptor over the pair's rows that exist (counted stores, ss_wave.h); lanes of a half-wave write 256 contiguous bytes
        // per register on both sides of the spectrum
        __amdgpu_buffer_rsrc_t srs = out_rsrc(nullptr, 0u);
        if (STFT) {
            const unsigned unit_s = __builtin_amdgcn_readfirstlane(unit);
            const unsigned clip_s = unit_s / pairs;
            const unsigned r0 = (unit_s - clip_s * pairs) * 2;
            srs = out_rsrc(a.out + (static_cast<unsigned long long>(clip_s) * R + r0) * 2050ull, min(2u, static_cast<unsigned>(R) - r0) * 8200u);
        }
        const int srow_off = half * 8200;
        const float cs = 0.5f * a.scale;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {  // two batches of 8: all partner fetches of a batch go out before its arithmetic
            float2 zcs[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) zcs[q] = make_float2(bperm(paddr, u[31 - (8 * hb + q)].x), bperm(paddr, u[31 - (8 * hb + q)].y));
            float4 tw4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) tw4[i] = s_twn4[4 * hb + i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const int q = 8 * hb + qq;
                const float2 zk = u[q];
                // lane 0 pairs with itself: Z[1024 - 32 q] = own register (32 - q) & 31
                const float2 zc = j == 0 ? u[(32 - q) & 31] : zcs[qq];
                const float2 w = (qq & 1) ? make_float2(tw4[qq >> 1].z, tw4[qq >> 1].w) : make_float2(tw4[qq >> 1].x, tw4[qq >> 1].y);
                const float2 sm = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                const float2 df = make_float2(zk.x - zc.x, zk.y + zc.y);
                // 2 X[k] = s - i w d: two chained FMAs per component
                const float xr = fmaf(w.y, df.x, fmaf(w.x, df.y, sm.x));
                const float xi = fmaf(w.y, df.y, fmaf(-w.x, df.x, sm.y));
                if (STFT) {
                    buf_store(make_float2(cs * xr, cs * xi), srs, srow_off + (j + 32 * q) * 8);
                    // 2 conj X[1024 - k] = 2 s - 2 X[k]
                    buf_store(make_float2(cs * fmaf(2.f, sm.x, -xr), -cs * fmaf(2.f, sm.y, -xi)), srs, srow_off + (1024 - j - 32 * q) * 8);
                } else {
                    prow[j + 32 * q] = hs * (xr * xr + xi * xi);
                }
            }
        }
        if (STFT) {
            buf_store(make_float2(a.scale * u[16].x, -a.scale * u[16].y), srs, j == 0 ? srow_off + 512 * 8 : kOobOffset);  // X[512] = conj Z[512]
            wave_order();
            SS_PRIOL(SS_P_TOP);
        } else {
            if (j == 0) {
                const float2 z = u[16];  // X[512] = conj Z[512]
                prow[512] = hs * 4.f * (z.x * z.x + z.y * z.y);
            }
            if (j < 3) prow[513 + j] = 0.f;  // pad bins read (with zero weight) by the mel stage
            wave_order();
            SS_PRIOL(SS_P_MEL);
            // ---- banded mel reduction (feature.rs:173), four filters per lane; the two rows of the wave are adjacent words of
            //      out[clip][m][.] ----
            {
                const float4 *w4 = reinterpret_cast<const float4 *>(s_tab + L::kMelW + j * a.mel_wpitch);
                float mv[4];
                if constexpr (FIXMEL) {
                    mel4_fixed<6, 3, 2, 1>(w4, reinterpret_cast<const float4 *>(prow + st[0]), reinterpret_cast<const float4 *>(prow + st[1]),
                                           reinterpret_cast<const float4 *>(prow + st[2]), reinterpret_cast<const float4 *>(prow + st[3]), mv);
                } else {
                    int off = 0;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        mv[s] = mel_slot4(w4 + off, reinterpret_cast<const float4 *>(prow + st[s]), a.mel_q4[s]);
                        off += a.mel_q4[s];
                    }
                }
                float *dst = out_b + static_cast<unsigned long long>(clip) * M * R + r;
                if (in_rows) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (fi[s] >= 0) dst[static_cast<unsigned long long>(fi[s]) * R] = mv[s];
                }
            }
            wave_order();
            SS_PRIOL(SS_P_TOP);
        }
#if SS_LAB && defined(SS_PROF3)
        SS_P3(2);
        ++p3n;
#endif
        item = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(next_v));
    }
    life_end(a.stamps, life, blockIdx.x * kWavesM + (threadIdx.x >> 6));
#if SS_LAB && defined(SS_PROF3)
    if ((threadIdx.x & 63) == 0) p3[0] = p3n | (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20)) << 32);
#endif
}

// ---- launchers: LDS budget, grid (one persistent workgroup per CU, fewer when there is not a unit per wave), the launch itself ----
size_t mel_lds_bytes(int waves, int wpitch)
{
    return (static_cast<size_t>(waves) * kWaveFloatsM + L::kMelW + 8 + 32 * static_cast<size_t>(wpitch)) * sizeof(float);
}
unsigned mel_grid(unsigned long long units, int waves, int num_cus)
{
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256), blocks = (units + waves - 1) / waves;
    return static_cast<unsigned>(blocks < cap ? blocks : cap);
}
// Eight waves per CU or twelve?  A unit (two rows) takes a wave 1.29 x as long with three waves on its SIMD as with two (cfg3, one
// box: 8.0 us against 6.2), and a CU's units go round in ceil(units / waves) rounds: twelve waves win unless the CU's share of
// units fits eight waves much better (cfg3: 64 units per CU, 8 rounds of 8 against 5.3 -> 6 of 12: 46.5 us against 49.7).
bool twelve_waves_win(unsigned long long units, int num_cus)
{
    const unsigned long long cus = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256), per_cu = (units + cus - 1) / cus;
    return 1.29 * static_cast<double>((per_cu + 11) / 12) < static_cast<double>((per_cu + 7) / 8);
}
template <typename Kern, typename... Args>
hipError_t mel_go(Kern kern, const char *name, unsigned grid, int waves, size_t lds, hipStream_t stream, LaunchInfo *info, const Args &...args)
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(waves * 64), lds};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, stream, args...);
    return hipGetLastError();
}

template <int kWavesM>
hipError_t launch_mel_w(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = mel_lds_bytes(kWavesM, a.mel_wpitch);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.batch == 0) return hipSuccess;
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * (a.out_stft ? (a.rows + 1) / 2 : mel_work_pairs(a.rows, a.real_rows));
    if (units >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned grid = mel_grid(units, kWavesM, num_cus);
    // (a fixed-shape mel stage costs this kernel 84 bytes of scratch per lane beside the prefetched unit: measured 77 us against
    // 50; the run-time loops stay, with their remainders fetched in one batch)
    if (a.out_stft) return mel_go(ss_mel_c1024<kWavesM, true>, "ss_mel_c1024<stft>", grid, kWavesM, lds, stream, info, a);
    if (a.fullp) return mel_go(ss_mel_c1024<kWavesM, false, true>, "ss_mel_c1024<fullp>", grid, kWavesM, lds, stream, info, a);
    return mel_go(ss_mel_c1024<kWavesM, false>, "ss_mel_c1024", grid, kWavesM, lds, stream, info, a);
}

// three waves per SIMD, direct stores (see ss_mel_c1024_w12): mel output with the reference bank shape, and stft
hipError_t launch_mel_w12(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    if (a.fullp || a.batch == 0) return hipErrorInvalidValue;
    const size_t lds = mel_lds_bytes(12, a.mel_wpitch);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * (a.out_stft ? (a.rows + 1) / 2 : mel_work_pairs(a.rows, a.real_rows));
    if (units >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned grid = mel_grid(units, 12, num_cus);
    const MultiArg<false> none{};
    if (a.out_stft) return mel_go(ss_mel_c1024_w12<false, true>, "ss_mel_c1024<w12,stft>", grid, 12, lds, stream, info, a, none);
    const bool m6321 = a.mel_q4[0] == 6 && a.mel_q4[1] == 3 && a.mel_q4[2] == 2 && a.mel_q4[3] == 1;
    return m6321 ? mel_go(ss_mel_c1024_w12<true>, "ss_mel_c1024<w12,mel6321>", grid, 12, lds, stream, info, a, none)
                 : mel_go(ss_mel_c1024_w12<false>, "ss_mel_c1024<w12>", grid, 12, lds, stream, info, a, none);
}

}  // namespace

hipError_t launch_mel_c1024_multi(const Mel2048Args &a_in, int n_batches, const float *const *d_x, float *const *d_out, const size_t *channels,
                                  hipStream_t stream, int num_cus, LaunchInfo *info)
{
    Mel2048Args a = a_in;
    // the build that exists: mel output, the reference bank shape (P rows of bins 0..512), compile-time tap counts 6 / 3 / 2 / 1
    const bool m6321 = a.mel_q4[0] == 6 && a.mel_q4[1] == 3 && a.mel_q4[2] == 2 && a.mel_q4[3] == 1;
    if (n_batches < 1 || n_batches > kMaxLaunchBatches || a.out_stft || a.fullp || !m6321) return hipErrorInvalidValue;
    const size_t lds = mel_lds_bytes(12, a.mel_wpitch);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned pairs = mel_work_pairs(a.rows, a.real_rows);
    MultiArg<true> mt{};
    unsigned long long units = 0;
    for (int b = 0; b < kMaxLaunchBatches; ++b) {
        mt.m.uend[b] = 0xffffffffu;
        if (b >= n_batches) continue;
        // Only blocks that launch_mel_c1024 would give the twelve-wave build on their own share a launch (empty ones are dropped by
        // the caller): the eight-wave build rounds a few FMAs differently in the last bit, and the results of a call must not depend
        // on how its blocks were grouped
        if (channels[b] == 0 || channels[b] > 0x7fffffffull || !twelve_waves_win(static_cast<unsigned long long>(channels[b]) * pairs, num_cus))
            return hipErrorInvalidValue;
        units += static_cast<unsigned long long>(channels[b]) * pairs;
        if (units >= 0xffffffffull) return hipErrorInvalidValue;
        mt.m.x[b] = d_x[b];
        mt.m.out[b] = d_out[b];
        mt.m.uend[b] = static_cast<uint32_t>(units);
        mt.m.total[b] = static_cast<uint32_t>(channels[b]);
    }
    a.x = d_x[0];
    a.out = d_out[0];
    a.batch = static_cast<uint32_t>(units);  // MULTI: the launch's unit count (the kernel takes the blocks from the table)
    return mel_go(ss_mel_c1024_w12<true, false, true>, "ss_mel_c1024m<w12,mel6321>", mel_grid(units, 12, num_cus), 12, lds, stream, info, a, mt);
}

hipError_t launch_mel_c1024(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * (a.out_stft ? (a.rows + 1) / 2 : mel_work_pairs(a.rows, a.real_rows));
#if SS_LAB
    // lab library: ss_debug_mel_tile(2) asks for the eight-wave builds only -- the retired whole-line tile
    // (tools/experiments/ss_mel2048_tile.hip) where the shape has one, else eight waves with direct stores
    if (dbg_mel_build() == 2) {
        const hipError_t e = launch_mel_c1024_tile(a, stream, num_cus, info);
        if (e != hipErrorInvalidValue) return e;
    }
    static const char *w = std::getenv("SS_MEL_WAVES");  // A/B knobs (lab build)
    if (w && std::atoi(w) == 12) return launch_mel_w<12>(a, stream, num_cus, info);
    static const char *sw = std::getenv("SS_STFT_WAVES");
    if (a.out_stft && sw) return std::atoi(sw) == 12 ? launch_mel_w12(a, stream, num_cus, info) : launch_mel_w<8>(a, stream, num_cus, info);
#endif
    if (a.out_stft && !a.fullp) {
        // stft writes 8200 bytes per row: the launch is a write-dominated stream.  While input + output fit the 256 MiB Infinity
        // Cache the twelve-wave build is the faster one (cfg3 shape, 768 clips: 51.7 against 60.5 us); beyond it both run at the
        // part's mixed read / write rate and the eight-wave build, which asks for the next unit's samples ahead of its stores,
        // is level or slightly ahead (1024 clips: 95.5 against 99.5 us; 4096 clips: 318 against 323).
        const unsigned long long bytes = static_cast<unsigned long long>(a.batch) * (4ull * a.n_samples + 8200ull * a.rows);
        if (bytes <= (240ull << 20)) {
            const hipError_t e = launch_mel_w12(a, stream, num_cus, info);
            if (e != hipErrorInvalidValue) return e;
        }
        return launch_mel_w<8>(a, stream, num_cus, info);
    }
    if (!a.out_stft && !a.fullp && dbg_mel_build() != 1 && dbg_mel_build() != 2 && (twelve_waves_win(units, num_cus) || dbg_mel_build() == 3)) {
        const hipError_t e = launch_mel_w12(a, stream, num_cus, info);
        if (e != hipErrorInvalidValue) return e;
    }
    return launch_mel_w<8>(a, stream, num_cus, info);
}

}  // namespace ss

// RETIRED from the product in round 4 (kept for the lab library: `make lab` links it; the product's ss_mel2048.hip has no tile
// path).  ss_mel_c1024<tile>: the eight-wave 2048-point mel-spectrogram kernel with the CU-wide whole-line output tile, its
// lock-free intra-CU hand-off protocol, bounded polls and device error word.  Same duration as direct stores (52.9 vs 52.8 us on
// cfg3, round 2) with HBM traffic 1.007x instead of 1.09x the algorithmic bytes; since round 3 the headline shape runs on the
// twelve-wave build, beside which the 55 KB tile does not fit, so no BASELINE shape selected this build any more.  The lab
// library keeps it as the carrier of the device-error-word test (ss_debug_tile_fault -> SS_ERR_DEVICE).
// This file is the round-3 ss_mel2048.hip with only the tile launcher exported (launch_mel_c1024_tile).
//
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
//   * output [clip][m][r]: with at least one clip per CU the clip's block is collected in a CU-wide LDS tile and leaves as
//     whole 128-byte lines (TILE below; HBM writes = the output, 1.00x).  Otherwise each lane stores its four mel values
//     straight to out[clip][m][r]: the wave's two rows are adjacent words, so the stores are 8-byte pieces of lines
//     whose other rows come from other waves and merge in L2 only partly (writes 1.39x the output).  No workgroup barrier
//     anywhere in the main loop in either build (a barrier-synchronised transposing tile measured 20 % slower in round 1).
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
namespace L = mel2048_layout;
constexpr int kExSlots = 2 * 16 * 34;        // float2 in the wave's exchange region: two frames x half the columns (8704 B)
constexpr int kWaveFloatsM = kExSlots * 2;  // one exchange region; the two P rows (2 x 520 floats) reuse it after the exchange


// TILE (mel output; rows <= 32 and a multiple of 4, filters <= 128 and a multiple of 8; batch >= CUs): a clip's [mel][row]
// block is collected in a CU-wide LDS tile and leaves as whole 128-byte lines.  No workgroup barrier: two LDS counters per
// tile buffer that only grow -- row pairs written into it, wave shares written out of it -- tell a wave when a clip is
// complete and when its buffer may be reused; each wave writes a fixed share (filters 8w .. 8w + 7 and 8(w + 8) ..) of every
// clip at its next unit after the clip completed, and the rest when it runs out of units.  kTileBufs buffers: the CU's waves
// run ahead of its slowest wave by at most kTileBufs - 1 clips before they wait.  Measured on cfg3 (profiles/r02): same
// duration as the direct stores (52.9 vs 52.8 us), HBM traffic 1.007x instead of 1.09x the algorithmic bytes.  What it took
// to get there: polls as relaxed atomics (a volatile LDS poll is a flat_load whose vmcnt(0) drains the prefetch), counter
// reads issued at the top of the unit, no returning atomics, 16-byte stores only (one wave flushing a whole tile: +5 us;
// slices handed out through a CAS counter: +9 us).
constexpr int kTileRows = 32, kTileMels = 128, kTilePitch = 36;  // [mel][row], rows of 144 bytes: 16-byte aligned for the flush
constexpr int kTileBufs = 3;  // clips a CU may have open at once (its waves run ahead of the slowest by up to kTileBufs - 1 clips)
constexpr int kTileFloats = kTileBufs * kTileMels * kTilePitch;

template <int kWavesM, bool STFT, bool FULLP = false, bool TILE = false, bool FIXMEL = false>
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
    // behind the table block: 4 words copied with it ([0] = the poll bound of the tile hand-offs), then the unit counter
    const unsigned *s_ctl = reinterpret_cast<const unsigned *>(s_tab + L::kMelW + 32 * a.mel_wpitch);
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + L::kMelW + 32 * a.mel_wpitch + 4);
    float *s_tile = reinterpret_cast<float *>(s_next + 4);                        // TILE: [kTileBufs][32][129]
    // both counters only ever grow, so nobody needs the value its own increment returned: clip c (the g-th user of its
    // buffer, g = (c - c_lo) / kTileBufs) may write the tile once s_fd == kWavesM g, and is complete at s_cnt == pairs (g + 1)
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_tile + kTileFloats);           // row pairs written into buffer b so far
    // polling reads of those words: relaxed atomics, not volatile -- a volatile access keeps the generic address space and
    // becomes a flat_load, whose s_waitcnt vmcnt(0) drains the wave's outstanding global loads and stores on every poll
    auto peek = [](const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    unsigned *s_fd = s_cnt + kTileBufs;  // shares of buffer b written out so far (kWavesM per clip)
    const unsigned pairs0 = (a.rows + 1) / 2;
    // TILE: the workgroup's range is made of whole clips
    const unsigned c_lo = static_cast<unsigned>(static_cast<unsigned long long>(a.batch) * blockIdx.x / gridDim.x);
    const unsigned c_hi = static_cast<unsigned>(static_cast<unsigned long long>(a.batch) * (blockIdx.x + 1) / gridDim.x);

    {
        const int n4 = (L::kMelW + 32 * a.mel_wpitch + 4) / 4;
        for (int i = tid; i < n4; i += kWavesM * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) {
            const unsigned long long units0 = static_cast<unsigned long long>(a.batch) * ((a.rows + 1) / 2);
            *s_next = (TILE ? c_lo * pairs0 : static_cast<unsigned>(units0 * blockIdx.x / gridDim.x)) + kWavesM;
            if (TILE) {
                for (int b = 0; b < kTileBufs; ++b) {
                    s_cnt[b] = 0u;
                    s_fd[b] = 0u;
                }
            }
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
    const unsigned pairs = (a.rows + 1) / 2;
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * pairs;
    const unsigned u_lo = TILE ? c_lo * pairs : static_cast<unsigned>(units * blockIdx.x / gridDim.x);
    const unsigned u_hi = TILE ? c_hi * pairs : static_cast<unsigned>(units * (blockIdx.x + 1) / gridDim.x);
    // TILE: a finished clip's tile leaves in slices of eight filters (eight whole lines, one 16-byte store per lane); wave w
    // writes slices w, w + kWavesM, ... of every clip, at its next stop after the clip's last row pair has arrived; once
    // all kWavesM shares are out the buffer is clip + kTileBufs's.  (One wave writing all 128 lines held that wave up for
    // microseconds; handing slices out through an LDS counter cost four LDS round trips per slice.)
    const int nsl = (M + 7) >> 3;
    unsigned fl_next = c_lo;  // the oldest clip whose share this wave has not written yet
    auto tile_full = [&](unsigned c) { return pairs * ((c - c_lo) / kTileBufs + 1u); };  // s_cnt of c's buffer once c is complete
    auto flush_one = [&]() {  // this wave's share of clip fl_next, which is complete
        const unsigned fb = (fl_next - c_lo) % kTileBufs;
        // rows % 4 == 0 and filters % 8 == 0 here (launcher): lane l holds rows 4 (l & 7) .. + 3 of filter 8 sl + (l >> 3)
        const float *tb = s_tile + fb * (kTileMels * kTilePitch) + (lane >> 3) * kTilePitch + (lane & 7) * 4;
        float *dstc = a.out + static_cast<unsigned long long>(fl_next) * M * R + (lane >> 3) * R + (lane & 7) * 4;
        if ((lane & 7) * 4 < R) {
            for (int sl = wave; sl < nsl; sl += kWavesM) *reinterpret_cast<float4 *>(dstc + sl * 8 * R) = *reinterpret_cast<const float4 *>(tb + sl * 8 * kTilePitch);
        }
        wave_order();
        if (lane == 0) atomicAdd(s_fd + fb, 1u);
        ++fl_next;
    };
    // A hand-off that never comes (a protocol error: it cannot happen unless a wave of this workgroup died or the counters
    // were corrupted) must neither hang nor pass for a result.  The waits are bounded; a wave that runs into the bound sets
    // the config's device error word (pinned host memory, so the host sees it without a copy: ss_api.hip turns it into
    // SS_ERR_DEVICE at the next launch / synchronisation point on the config) and ends (s_endpgm): it writes nothing more
    // into the tile, flushes nothing, takes no further unit.  Its peers then run into their own bounds and end as well.
    // (A trap measured 2 us on the whole launch; NaNs in the output could be overwritten by a later flush.)
    // (Everything about this lives on the cold side of a branch, behind ONE kernel argument: the kernel sits at the SGPR limit
    // and every extra argument or flag on the hot path measured +1 us of 49.)
    auto protocol_error = [&]() {
        if (lane == 0 && a.ctl) __hip_atomic_store(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0);  // the store has left the wave
        __builtin_amdgcn_endpgm();       // the wave ends here: no flag to test anywhere on the hot path
    };
    // polls before a hand-off counts as lost: the word behind the table block, copied into LDS with it (2^24 in normal operation:
    // ~0.5 s; ss_debug_tile_fault sets 0, so that the first wait that is not satisfied at once takes the error path).  An LDS
    // read on purpose: a global load here would wait on vmcnt, i.e. for this wave's outstanding output stores (+2 us of 49).
    auto spin_limit = [&]() { return peek(s_ctl); };
    auto flush_share = [&](unsigned upto, bool wait) {
        while (fl_next < upto) {
            const unsigned fb = (fl_next - c_lo) % kTileBufs;
            const unsigned full = tile_full(fl_next);
            if (peek(s_cnt + fb) != full) {
                if (!wait) return;
                const unsigned lim = spin_limit();
                unsigned tries = 0;
                while (peek(s_cnt + fb) != full && tries < lim) {
                    __builtin_amdgcn_s_sleep(1);
                    ++tries;
                }
                if (peek(s_cnt + fb) != full) protocol_error();
            }
            flush_one();
        }
    };
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

    unsigned unit = u_lo + wave;
    float2 v[32];
    if (unit < u_hi) load_unit(unit, v);
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
                // TILE: this wave's share of the finished clips behind this one leaves here, right after the unit's samples
                // have arrived: vmcnt retires in order and a write is acknowledged microseconds after it was issued, so
                // stores issued later in the unit (ahead of or behind the next unit's loads) made that unit wait for them
                // TILE: the two counters this unit will look at are read here, long before their values are needed, so that the
                // round trips hide behind the transform (a stale value only postpones the flush to the wave's next unit)
                unsigned seen_cnt = 0, seen_fd = 0;
                if (TILE) {
                    seen_cnt = peek(s_cnt + (fl_next - c_lo) % kTileBufs);
                    seen_fd = peek(s_fd + (clip - c_lo) % kTileBufs);
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
                if (TILE && fl_next < clip && seen_cnt == tile_full(fl_next)) flush_one();
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
                float2 *srow = nullptr;
                if (STFT && in_rows) srow = reinterpret_cast<float2 *>(a.out) + (static_cast<unsigned long long>(clip) * R + r) * 1025ull;
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
                            if (srow) {
                                srow[j + 32 * q] = make_float2(cs * xr, cs * xi);
                                // 2 conj X[1024 - k] = 2 s - 2 X[k]
                                srow[1024 - j - 32 * q] = make_float2(cs * fmaf(2.f, s.x, -xr), -cs * fmaf(2.f, s.y, -xi));
                            }
                        } else {
                            prow[j + 32 * q] = hs * (xr * xr + xi * xi);   // (|X| wnorm)^2, functions.rs:166-169 + feature.rs:164
                            if (FULLP) {  // bins 513..1024 as well
                                const float yr = fmaf(2.f, s.x, -xr), yi = fmaf(2.f, s.y, -xi);
                                prow[1024 - j - 32 * q] = hs * (yr * yr + yi * yi);
                            }
                        }
                    }
                }
                if (j == 0) {
                    const float2 z = u[16];  // X[512] = conj Z[512]
                    if (STFT) {
                        if (srow) srow[512] = make_float2(a.scale * z.x, -a.scale * z.y);
                    } else {
                        prow[512] = hs * 4.f * (z.x * z.x + z.y * z.y);
                    }
                }
                if (STFT) {
                    wave_order();
                    if (!PREFETCH_M && next < u_hi) load_unit(next, v);
                    unit = next;
                    continue;
                }
                if (j < 3) prow[(FULLP ? 1025 : 513) + j] = 0.f;  // pad bins read (with zero weight) by the mel stage
                wave_order();
                // ---- banded mel reduction (feature.rs:173), four filters per lane; the two rows of the wave are
                //      adjacent words of out[clip][m][.] ----
                if (TILE) {
                    const unsigned b = (clip - c_lo) % kTileBufs;
                    float mv[4];
                    if constexpr (FIXMEL) {
                        // cfg3's bank shape (128 filters up to 8 kHz: 6 / 3 / 2 / 1 float4s per slot): every weight and tap is
                        // requested before the first FMA -- one LDS wait for the stage
                        mel4_fixed2<6, 3, 2, 1>(w4, reinterpret_cast<const float4 *>(prow + st[0]), reinterpret_cast<const float4 *>(prow + st[1]),
                                               reinterpret_cast<const float4 *>(prow + st[2]), reinterpret_cast<const float4 *>(prow + st[3]), mv);
                    } else {
                        int off = 0;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            mv[s] = mel_slot4(w4 + off, reinterpret_cast<const float4 *>(prow + st[s]), a.mel_q4[s]);
                            off += a.mel_q4[s];
                        }
                    }
                    // the buffer is ours once clip - kTileBufs has left it
                    const unsigned freed = kWavesM * ((clip - c_lo) / kTileBufs);
                    for (unsigned tries = 0; seen_fd != freed && peek(s_fd + b) != freed; ++tries) {
                        flush_share(clip, false);  // the buffer may be waiting for this very wave's share of an older clip
                        __builtin_amdgcn_s_sleep(1);
                        if (tries > (1u << 24)) protocol_error();  // the buffer still belongs to an older clip: do not touch it
                    }
                    float *tcol = s_tile + b * (kTileMels * kTilePitch) + r;
                    if (in_rows) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (fi[s] >= 0) tcol[fi[s] * kTilePitch] = mv[s];
                    }
                    wave_order();
                    if (lane == 0) atomicAdd(s_cnt + b, 1u);
                } else if (in_rows) {
                    float *dst = a.out + static_cast<unsigned long long>(clip) * M * R + r;
                    float mfix[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (FIXMEL)
                        mel4_fixed2<6, 3, 2, 1>(w4, reinterpret_cast<const float4 *>(prow + st[0]), reinterpret_cast<const float4 *>(prow + st[1]),
                                               reinterpret_cast<const float4 *>(prow + st[2]), reinterpret_cast<const float4 *>(prow + st[3]), mfix);
                    int off = 0;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float m = FIXMEL ? mfix[s] : mel_slot4(w4 + off, reinterpret_cast<const float4 *>(prow + st[s]), a.mel_q4[s]);
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
    if (TILE) {
        // out of units: what is left of the range's last clips (bounded wait for rows other waves are still computing)
        flush_share(c_hi, true);
    }
}

}  // namespace

// The tile build only: SS_ERR-less contract of the launchers (hipErrorInvalidValue = this shape has no tile build).
hipError_t launch_mel_c1024_tile(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    constexpr int kWavesM = 8;
    size_t lds = (static_cast<size_t>(kWavesM) * kWaveFloatsM + L::kMelW + 8 + 32 * static_cast<size_t>(a.mel_wpitch)) * sizeof(float);
    lds += (kTileFloats + 2 * kTileBufs + 2) * sizeof(float);
    const bool tile = !a.out_stft && !a.fullp && a.rows <= kTileRows && a.rows % 4 == 0 && a.n_filters <= 128 && a.n_filters % 8 == 0 &&
                      lds <= 160 * 1024 && a.batch >= static_cast<uint32_t>(num_cus > 0 ? num_cus : 256);
    if (!tile) return hipErrorInvalidValue;
    const unsigned cap = static_cast<unsigned>(num_cus > 0 ? num_cus : 256);
    const unsigned long long units = static_cast<unsigned long long>(a.batch) * ((a.rows + 1) / 2);
    if (units >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned long long blocks = (units + kWavesM - 1) / kWavesM;
    const unsigned grid = static_cast<unsigned>(blocks < cap ? blocks : cap);
    auto kern = ss_mel_c1024<kWavesM, false, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    if (info) *info = LaunchInfo{"ss_mel_c1024<tile>", grid, static_cast<unsigned>(kWavesM * 64), lds};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWavesM * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace ss

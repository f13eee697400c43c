// ss_mfcc_c256_pk: fused MFCC for fft_points = 512 (C = 256 packed complex points) on gfx950,
// two frames per 16-lane group so that every arithmetic instruction is a packed v_pk_*_f32.
//
// Why: a gfx950 SIMD issues one VALU instruction per 4 cycles, packed or not (measured:
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.1), and this path is VALU-issue bound, so the lever is
// instructions per frame.  Packing (frame A, frame B) halves them with no swizzle overhead.
//
// Work unit: an OCTET of 8 consecutive frames.  One persistent 8-wave workgroup per CU owns a
// contiguous range of octets; its waves pull octets from an LDS counter.  Lane = 16 g + j: group g
// (one DPP row) owns frames 2g and 2g+1 of the octet, 16 complex points of each per lane.
//   A1  coalesced 8-byte loads (128 B per frame row); the next octet is prefetched.
//   A2  radix-16 register butterfly, zero padding folded at compile time (template NE).
//   A3  ONE transposing exchange through wave-private LDS: element (n1, k1) of a frame pair is one
//       16-byte slot (reA, reB, imA, imB) at 17 n1 + k1; 16 ds_write_b128 + 16 ds_read_b128 per lane,
//       conflict-free on both sides (row pitch 272 B on the write side, 256-B-multiple regions).
//   A4  twiddle (table in LDS) + second radix-16 butterfly: lane j holds Z[j + 16 r].
//   A5  the real-FFT untangle needs Z[256-k] = lane 16-j, register 15-r: ds_bpermute_b32.
//   A6  |X| / N (processing.rs:168,180); bins 0..128 go to the wave's P tile, which reuses the
//       exchange region; all 257 bins feed the frame energy, reduced over the DPP row (feature.rs:216-219).
//   B1  mel^T[filter][frame] = sum_bin W[filter][bin] P[frame][bin] (feature.rs:229) on the matrix pipe,
//       v_mfma_f32_16x16x4_f32 over ONLY the non-zero 16-filter x 4-bin blocks of the banded bank.
//   B2  zero handling (:230) + ln (:105);  B3  DCT-II (:120-123) as 12 more MFMAs whose B operand
//       is B1's accumulator registers;  B4  scaling + column 0 (:126-146), staged coalesced store.
// LDS operations of one wave execute in order, so the main loop has no barrier.
//
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.
#include "ss_device.h"
#include "ss_fft_pk.h"
#include "ss_internal.h"

#include <cstdlib>

namespace ss {

namespace {

constexpr float kEpsP = 1.1920929e-7f;  // f32::EPSILON, functions.rs:70
constexpr int kGroupSlots = 272;                  // 16-byte slots per frame-pair exchange region (17 x 16)
// Per-wave LDS: the exchange region (4 frame pairs at once, or 2 at a time when HALF) | later P tile, stage, ln(energy)
template <bool HALF> constexpr int kWaveBytesP = (HALF ? 2 : 4) * kGroupSlots * 16;  // 17408 B or 8704 B
// P tile (reuses the exchange region): [pair 4][half 2][k-sub 4][36 k-steps] floats; bin = 4 s + k-sub.
// One ds_read_b128 then feeds four consecutive k-steps of a lane's MFMA B operand.
constexpr int kPPair = 288, kPHalf = 144, kPSub = 36;
constexpr int kStageOff = 1280;                   // float offsets inside the wave region (beyond the 1152-float P tile)
constexpr int kElogOff = 1536;
constexpr int kTabTw2 = fast512m_layout::kTw2;
constexpr int kTabTwn = fast512m_layout::kTwn;
constexpr int kTabCt = fast512m_layout::kCt;
constexpr int kTabWt = fast512m_layout::kWt;

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CTRL>
__device__ __forceinline__ float dpp_movp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row, both halves; every lane ends with the same bits
__device__ __forceinline__ v2f row16_sum2(v2f v)
{
    v += v2f{dpp_movp<0xB1>(v.x), dpp_movp<0xB1>(v.y)};    // quad_perm [1,0,3,2]
    v += v2f{dpp_movp<0x4E>(v.x), dpp_movp<0x4E>(v.y)};    // quad_perm [2,3,0,1]
    v += v2f{dpp_movp<0x141>(v.x), dpp_movp<0x141>(v.y)};  // row_half_mirror
    v += v2f{dpp_movp<0x140>(v.x), dpp_movp<0x140>(v.y)};  // row_mirror
    return v;
}

__device__ __forceinline__ void wave_order()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float bpermf(int addr, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// ln(x) from v_log_f32 (log2) with the denormal pre-scale the library form uses; ~1 ulp of log2.
__device__ __forceinline__ float fast_ln(float x)
{
    const bool tiny = x < 1.17549435e-38f;
    const float l = __builtin_amdgcn_logf(tiny ? x * 4294967296.f : x);
    return (l - (tiny ? 32.f : 0.f)) * 0.69314718055994530942f;
}

// One 16-filter tile of the block-sparse mel product: k-step groups [q, qh), four MFMAs per group.
// `wt4` walks the grouped weight table (tile-major), `pb4` is the lane's P-tile row.
__device__ __forceinline__ f32x4 mel_tile(const float4 *&wt4, const float4 *pb4, int q, int qh)
{
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < qh) {
        float4 wa = *wt4, pb = pb4[q];
        for (; q < qh; ++q) {
            wt4 += 64;
            const bool more = q + 1 < qh;
            const float4 wn = more ? *wt4 : wa;
            const float4 pn = more ? pb4[q + 1] : pb;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, pb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, pb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, pb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, pb.w, acc, 0, 0, 0);
            wa = wn;
            pb = pn;
        }
    }
    return acc;
}

// zero handling (feature.rs:230) + ln (:105) feeding one k-step of the DCT product (:120-123)
__device__ __forceinline__ f32x4 dct_step(f32x4 o, float c, float x)
{
    x = x == 0.f ? kEpsP : x;
    return __builtin_amdgcn_mfma_f32_16x16x4f32(c, fast_ln(x), o, 0, 0, 0);
}

// Loads the first-pass inputs of frames 2g and 2g+1 of `octet`: z[n] = x[2n] + i x[2n+1], n = j + 16 e.
template <int NE, bool EXACT>
__device__ __forceinline__ void load_octet(const Fast512MArgs &a, unsigned octet, unsigned total, int g, int j,
                                           float2 (&va)[NE], float2 (&vb)[NE])
{
    unsigned fa = octet * 8 + 2 * g;
    fa = fa < total ? fa : total - 1;
    const unsigned ca = fa / a.n_frames, ta = fa - ca * a.n_frames;
    // frame B = A + 1: next frame of the same clip, first frame of the next clip, or A again past the end
    const bool last = fa + 1 >= total, wrap = ta + 1 == a.n_frames;
    const unsigned cb = (wrap && !last) ? ca + 1 : ca;
    const unsigned tb = last ? ta : (wrap ? 0u : ta + 1);
    // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
    const float2 *sa = reinterpret_cast<const float2 *>(a.x + static_cast<unsigned long long>(ca) * a.ld + ta * a.step);
    const float2 *sb = reinterpret_cast<const float2 *>(a.x + static_cast<unsigned long long>(cb) * a.ld + tb * a.step);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int n = j + 16 * e;
        if (EXACT) {
            va[e] = sa[n];
            vb[e] = sb[n];
        } else {
            const bool in = 2 * n < static_cast<int>(a.flen);
            va[e] = in ? sa[n] : make_float2(0.f, 0.f);
            vb[e] = in ? sb[n] : make_float2(0.f, 0.f);
        }
    }
}

template <int NE, bool EXACT, bool POW2, int kWavesP, bool HALF, bool PREFETCH>
__global__ __launch_bounds__(kWavesP * 64) void ss_mfcc_c256_pk(const Fast512MArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const unsigned long long t_start = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int g = lane >> 4;  // frame pair within the octet
    const int j = lane & 15;  // lane within the pair's DPP row

    // ---- LDS carve: per-wave regions, then the shared table block, then the octet counter ----
    float *wbase = reinterpret_cast<float *>(smem + wave * kWaveBytesP<HALF>);
    float4 *ex = reinterpret_cast<float4 *>(wbase) + (HALF ? (g & 1) : g) * kGroupSlots;  // this pair's exchange region
    float *ptile = wbase;                                              // [4 pairs][132 bins][A,B] after the exchange
    float *stage = wbase + kStageOff;
    float *elog = wbase + kElogOff;
    float *s_tab = reinterpret_cast<float *>(smem + kWavesP * kWaveBytesP<HALF>);
    const float2 *s_tw2 = reinterpret_cast<const float2 *>(s_tab + kTabTw2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + kTabTwn);
    const float *s_ct = s_tab + kTabCt;
    const float *s_wt = s_tab + kTabWt;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + kTabWt + a.n_mm * 64);

    const unsigned total = a.batch * a.n_frames;
    const unsigned octets = (total + 7) / 8;
    const unsigned c_lo = static_cast<unsigned>(static_cast<unsigned long long>(octets) * blockIdx.x / gridDim.x);
    const unsigned c_hi = static_cast<unsigned>(static_cast<unsigned long long>(octets) * (blockIdx.x + 1) / gridDim.x);

    // the whole table block arrives as float4s, global layout == LDS layout
    {
        const int n4 = (kTabWt + a.n_mm * 64) / 4;
        for (int i = tid; i < n4; i += kWavesP * 64) reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (tid == 0) *s_next = c_lo + kWavesP;
    }
    unsigned oct = c_lo + wave;
    float2 va[NE], vb[NE];
    if (oct < c_hi) load_octet<NE, EXACT>(a, oct, total, g, j, va, vb);

    const int paddr = ((lane & 48) | ((16 - j) & 15)) << 2;  // lane holding Z[256 - k]
    const int Cc = static_cast<int>(a.n_ceps);
    // |X| = (1/2)|...|: the 1/2 of the untangle is folded into the scale (1/4 for the squared form)
    const float hscale = POW2 ? 0.25f * a.scale : 0.5f * a.scale;
    __syncthreads();
    float ct[12];  // DCT MFMA A operands, resident
#pragma unroll
    for (int i = 0; i < 12; ++i) ct[i] = s_ct[i * 64 + lane];
    const int pw = g * kPPair + (j & 3) * kPSub + (j >> 2);  // P tile write base: bin j + 16 r -> k-sub j & 3, k-step (j >> 2) + 4 r
    const unsigned long long t_pro = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned n_done = 0;

    while (oct < c_hi) {
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);
        ++n_done;

        if (!PREFETCH && n_done > 1) load_octet<NE, EXACT>(a, oct, total, g, j, va, vb);
        cx2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {  // zero padding to fft_points, processing.rs:147-156
            if (e < NE) v[e] = cx2{v2f{va[e].x, vb[e].x}, v2f{va[e].y, vb[e].y}};
            else v[e] = cx2{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        }
        if (PREFETCH && next < c_hi && !(a.ablate & 1)) load_octet<NE, EXACT>(a, next, total, g, j, va, vb);

        // ---- 256-point complex FFT of both frames: radix-16, transpose through LDS, twiddle, radix-16 ----
        fft16_pk(v);
        if (!(a.ablate & 2)) {
            if (HALF) {
                // two frame pairs at a time through the same 8704-B region; a lane's registers are free once it
                // has written them, so the reads land in place
                if (lane < 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[17 * j + r] = make_float4(v[r].x.x, v[r].x.y, v[r].y.x, v[r].y.y);
                    wave_order();
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const float4 t4 = ex[17 * n1 + j];
                        v[n1] = cx2{v2f{t4.x, t4.y}, v2f{t4.z, t4.w}};
                    }
                }
                wave_order();
                if (lane >= 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[17 * j + r] = make_float4(v[r].x.x, v[r].x.y, v[r].y.x, v[r].y.y);
                    wave_order();
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const float4 t4 = ex[17 * n1 + j];
                        v[n1] = cx2{v2f{t4.x, t4.y}, v2f{t4.z, t4.w}};
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ex[17 * j + r] = make_float4(v[r].x.x, v[r].x.y, v[r].y.x, v[r].y.y);
                wave_order();
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const float4 t4 = ex[17 * n1 + j];
                    v[n1] = cx2{v2f{t4.x, t4.y}, v2f{t4.z, t4.w}};
                }
            }
        }
        wave_order();
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            const float2 w = s_tw2[(r - 1) * 16 + j];
            v[r] = cmul(v[r], w.x, w.y);
        }
        fft16_pk(v);  // v[r] = Z[j + 16 r]

        // the exchange region now becomes the P tile: clear k-steps 32..35 of all 32 rows (bins 129..143 stay zero;
        // bin 128 is written below)
        ptile[(lane >> 2) * kPSub + 32 + (lane & 3)] = 0.f;
        ptile[(16 + (lane >> 2)) * kPSub + 32 + (lane & 3)] = 0.f;

        // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
        v2f esum = v2f{0.f, 0.f};
        float *prow = ptile + pw;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const cx2 zk = v[r];
            // partner register 15 - r; lane 0 pairs with itself: Z[256 - 16 r] = own register (16 - r) & 15
            cx2 zc = v[15 - r];
            if (!(a.ablate & 4)) {
                zc.x = v2f{bpermf(paddr, v[15 - r].x.x), bpermf(paddr, v[15 - r].x.y)};
                zc.y = v2f{bpermf(paddr, v[15 - r].y.x), bpermf(paddr, v[15 - r].y.y)};
            }
            if (j == 0) zc = v[(16 - r) & 15];
            const float2 w = s_twn[r * 16 + j];
            const cx2 s = cx2{zk.x + zc.x, zk.y - zc.y};  // 2 E[k]
            const cx2 d = cx2{zk.x - zc.x, zk.y + zc.y};
            const cx2 wd = cmul(d, w.x, w.y);
            const v2f xa_r = s.x + wd.y, xa_i = s.y - wd.x;  // 2 X[k]
            const v2f xb_r = s.x - wd.y, xb_i = s.y + wd.x;  // 2 conj X[256-k]
            const v2f na = xa_r * xa_r + xa_i * xa_i, nb = xb_r * xb_r + xb_i * xb_i;
            v2f pa, pb;
            if (POW2) {
                pa = na * hscale;
                pb = nb * hscale;
            } else {
                pa = v2f{__builtin_amdgcn_sqrtf(na.x), __builtin_amdgcn_sqrtf(na.y)} * hscale;
                pb = v2f{__builtin_amdgcn_sqrtf(nb.x), __builtin_amdgcn_sqrtf(nb.y)} * hscale;
            }
            // only bins <= 128 can carry mel weight (the bank ends at (F+1)/2, feature.rs:69-70)
            prow[4 * r] = pa.x;
            prow[kPHalf + 4 * r] = pa.y;
            esum += pa + pb;
        }
        if (j == 0) {
            // lane 0's pair k = 0 produced X[0] and X[256]; X[128] = conj Z[128] is the one extra bin
            const cx2 z = v[8];
            const v2f n = (z.x * z.x + z.y * z.y) * 4.f;
            v2f p128;
            if (POW2) p128 = n * hscale;
            else p128 = v2f{__builtin_amdgcn_sqrtf(n.x), __builtin_amdgcn_sqrtf(n.y)} * hscale;
            prow[32] = p128.x;  // lane j = 0: k-sub 0, k-step 32
            prow[kPHalf + 32] = p128.y;
            esum += p128;
        }
        v2f energy = row16_sum2(esum);
        energy.x = energy.x == 0.f ? kEpsP : energy.x;  // zero_handling, feature.rs:219
        energy.y = energy.y == 0.f ? kEpsP : energy.y;
        if (j == 0) *reinterpret_cast<float2 *>(&elog[2 * g]) = make_float2(fast_ln(energy.x), fast_ln(energy.y));
        wave_order();

        // ---- B1: block-sparse mel product on the matrix pipe (feature.rs:229) ----
        // B operand: column n = lane & 15 is frame n & 7 (pair (n & 7) >> 1, half n & 1); row k = lane >> 4 is bin 4 s + k.
        // One float4 = the lane's operand for k-steps 4 q .. 4 q + 3; the next group is fetched while this one multiplies.
        const float4 *pb4 = reinterpret_cast<const float4 *>(ptile + ((lane & 7) >> 1) * kPPair + (lane & 1) * kPHalf + (lane >> 4) * kPSub);
        const float4 *wt4 = reinterpret_cast<const float4 *>(s_wt) + lane;
        const int mlo = (a.ablate & 8) ? 100 : 0;
        const f32x4 acc0 = mel_tile(wt4, pb4, a.ks_lo[0] + mlo, a.ks_hi[0]);
        const f32x4 acc1 = mel_tile(wt4, pb4, a.ks_lo[1] + mlo, a.ks_hi[1]);
        const f32x4 acc2 = mel_tile(wt4, pb4, a.ks_lo[2] + mlo, a.ks_hi[2]);
        // ---- B2: zero handling (feature.rs:230) + ln (:105);  B3: DCT-II (:120-123) ----
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) o = dct_step(o, ct[i], acc0[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) o = dct_step(o, ct[4 + i], acc1[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) o = dct_step(o, ct[8 + i], acc2[i]);
        // ---- B4: scaling + column-0 replacement (feature.rs:126-146), staged coalesced store ----
        {
            const int fr = lane & 15, q = lane >> 4;
            if (fr < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * q + i;
                    float val = o[i] * a.dct_scale_k;
                    if (c == 0) {
                        if (a.dc_elimination) {
                            val = elog[fr];
                        } else {
                            const unsigned gfr = min(oct * 8 + fr, total - 1);
                            const unsigned t = gfr % a.n_frames;
                            val = o[i] * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                        }
                    }
                    if (c < Cc) stage[fr * Cc + c] = val;
                }
            }
        }
        wave_order();
        {
            const unsigned first = oct * 8;
            const unsigned nfr = min(8u, total - first);
            const int nout = static_cast<int>(nfr) * Cc;
            float *dst = a.out + static_cast<unsigned long long>(first) * Cc;
            if (!(a.ablate & 16))
                for (int i = lane; i < nout; i += 64) dst[i] = stage[i];
        }
        wave_order();
        oct = next;
    }
    if (a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + 4ull * (blockIdx.x * kWavesP + wave);
        d[0] = t_start;
        d[1] = t_pro;
        d[2] = __builtin_amdgcn_s_memrealtime();
        d[3] = (static_cast<unsigned long long>(n_done) << 32) | __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // XCC_ID
    }
}

template <int kWavesP, bool HALF, bool PREFETCH>
hipError_t launch_pk(const Fast512MArgs &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = static_cast<size_t>(kWavesP) * kWaveBytesP<HALF> + static_cast<size_t>(kTabWt + a.n_mm * 64) * sizeof(float) + 16;  // n_mm = 4 * n_grp
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    const unsigned long long octets = (total + 7) / 8;
    // one workgroup per CU; fewer when there is not at least one octet per wave
    unsigned long long blocks = (octets + kWavesP - 1) / kWavesP;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(kWavesP * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kWavesP * 64), lds, stream, a);
        return hipGetLastError();
    };
    const bool pow2 = a.spectrum_exponent == 2;
    if (a.flen == 320) {
        return pow2 ? go(ss_mfcc_c256_pk<10, true, true, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<10,true,pow2>")
                    : go(ss_mfcc_c256_pk<10, true, false, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<10,true>");
    }
    if (a.flen == 512) {
        return pow2 ? go(ss_mfcc_c256_pk<16, true, true, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<16,true,pow2>")
                    : go(ss_mfcc_c256_pk<16, true, false, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<16,true>");
    }
    if (a.flen <= 320) {
        return pow2 ? go(ss_mfcc_c256_pk<10, false, true, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<10,false,pow2>")
                    : go(ss_mfcc_c256_pk<10, false, false, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<10,false>");
    }
    return pow2 ? go(ss_mfcc_c256_pk<16, false, true, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<16,false,pow2>")
                : go(ss_mfcc_c256_pk<16, false, false, kWavesP, HALF, PREFETCH>, "ss_mfcc_c256_pk<16,false>");
}

}  // namespace

hipError_t launch_mfcc_c256_pk(const Fast512MArgs &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    static const char *w = std::getenv("SS_PK_WAVES");  // A/B knob for occupancy experiments
    const int nw = w ? std::atoi(w) : 8;
    if (nw == 16) return launch_pk<16, true, false>(a, stream, num_cus, info);
    if (nw == 12) return launch_pk<12, true, false>(a, stream, num_cus, info);
    if (nw == 13) return launch_pk<12, true, true>(a, stream, num_cus, info);
    return launch_pk<8, false, true>(a, stream, num_cus, info);
}

}  // namespace ss

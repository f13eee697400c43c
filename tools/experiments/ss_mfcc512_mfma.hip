// ss_mfcc_c256_mx: fused MFCC for fft_points = 512 (C = 256 packed complex points) on gfx950.
//
// Work unit: a CHUNK of 8 consecutive frames = two quads.  A persistent 16-wave workgroup per CU owns
// a contiguous range of chunks; its waves pull chunks from an LDS counter (dynamic balance inside the
// CU; no global atomics).  Per quad (4 frames, 16 lanes = one DPP row per frame):
//   A1  10 x 8-byte coalesced loads per lane (128 B per frame row); the next quad is prefetched.
//   A2  radix-16 register butterfly, zero padding folded at compile time (template NE).
//   A3  ONE transposing exchange through wave-private LDS, two frames at a time (lanes 0-31, then
//       32-63) so the buffer is 2 x 2304 B: ds_write_b64 scatter to (n1,k1) -> 34*(n1>>1) + 2*k1 + (n1&1),
//       read back as 8 x ds_read_b128 per lane; both sides conflict-free.  LDS operations of one
//       wave execute in order, so no barrier is needed anywhere in the main loop.
//   A4  twiddle (table in LDS) + second radix-16 butterfly: lane j holds Z[j + 16 r].
//   A5  the real-FFT untangle needs Z[256-k] = lane 16-j, register 15-r: ds_bpermute_b32 (crossbar only).
//   A6  |X| / N; bins 0..128 go to the wave's P tile [8 frames x 132]; all 257 feed the frame energy,
//       reduced over the DPP row (processing.rs:168,180; feature.rs:216-219).
// After the two quads the wave runs the reference's two small contractions on the otherwise idle
// matrix pipe (v_mfma_f32_16x16x4_f32: exact f32, each MFMA is an fmaf chain):
//   B1  mel^T[filter][frame] = sum_bin W[filter][bin] P[frame][bin] (feature.rs:229) over ONLY the non-zero
//       16-filter x 4-bin blocks of the banded bank (35 of 99 at the defaults): block-sparse, not dense.
//   B2  zero handling (feature.rs:230) + ln (:105) on the 12 accumulator registers.
//   B3  DCT-II (:120-123): out^T[ceps][frame] = sum_filter cos[ceps][filter] L[filter][frame]; B1's accumulator
//       registers ARE B3's B operand (the contraction runs over B1's row index): no data movement.
//   B4  scaling + column-0 replacement (:126-146), staged through LDS, coalesced store.
// Occupancy: 8.9 KB of LDS per wave and <= 128 VGPRs -> 16 waves per CU (4 per SIMD); the kernel is
// VALU-issue bound, so the file is compiled with -fno-slp-vectorize (v_pk_* f32 ops cost twice the
// pipe cycles of the scalar forms here and triple the register pressure).
//
// Reference semantics: feature.rs:99-148 (mfcc), :200-233 (mfe), processing.rs:65-181.
#include "ss_device.h"
#include "ss_fft_reg.h"
#include "ss_internal.h"

#include <cstdlib>

namespace ss {

namespace {

constexpr float kEpsM = 1.1920929e-7f;  // f32::EPSILON, functions.rs:70
constexpr int kZStride = 288;           // float2 per frame exchange region (2304 B = 9 bank rows)
constexpr int kPPitch = 132;            // floats per P row: bins 0..128 + 3 zero pad bins read by the last k-step
constexpr int kWaveFloats = 2 * kZStride * 2 + 8 * kPPitch + 8;  // 2-frame exchange | P tile [8][132] | ln(energy)[8]
constexpr int kWaveBytes = ((kWaveFloats * 4 + 63) / 64) * 64;
// table block (float units), identical layout in global memory and LDS (ss_internal.h)
constexpr int kTabTw2 = fast512m_layout::kTw2;
constexpr int kTabTwn = fast512m_layout::kTwn;
constexpr int kTabCt = fast512m_layout::kCt;
constexpr int kTabWt = fast512m_layout::kWt;

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row; every lane ends with the same bits
__device__ __forceinline__ float row16_sum_m(float v)
{
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}

// Wave-private LDS hand-off: LDS operations of one wave execute in order, so all that is needed is
// that the compiler keeps the program order of the accesses around this point.
__device__ __forceinline__ void wave_sync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float bperm(int addr, float v)
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

// One 16-filter tile of the block-sparse mel product over k-steps [s, sh).  LDS latency under load is several
// hundred cycles, so the operands of up to KB k-steps are fetched back to back (one wait) before the MFMAs
// issue back to back.  `wt` walks the weight table (tile-major), `pbp` is the lane's P row.
template <int KB>
__device__ __forceinline__ f32x4 mel_tile(const float *&wt, const float *pbp, int s, int sh)
{
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    while (s < sh) {
        const int n = min(sh - s, KB);
        float wa[KB], pb[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if (i < n) {
                wa[i] = wt[64 * i];
                pb[i] = pbp[4 * (s + i)];
            }
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if (i < n) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], pb[i], acc, 0, 0, 0);
        }
        wt += 64 * n;
        s += n;
    }
    return acc;
}

// zero handling (feature.rs:230) + ln (:105) feeding one k-step of the DCT product (:120-123)
__device__ __forceinline__ f32x4 dct_step(f32x4 o, float c, float x)
{
    x = x == 0.f ? kEpsM : x;
    return __builtin_amdgcn_mfma_f32_16x16x4f32(c, fast_ln(x), o, 0, 0, 0);
}

// Diagnostic stamp (SS_DEBUG_TIMES runs only): shader-clock time with the LDS queue drained, fenced against
// the scheduler on both sides.  Accumulates segment lengths into seg[i].
#define SS_STAMP(i)                                                                      \
    if (a.dbg) {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        seg[i] += t_ - t_prev;                                                           \
        t_prev = t_;                                                                     \
    }

template <int NE, bool EXACT>
__device__ __forceinline__ void load_quad(const Fast512MArgs &a, unsigned quad, unsigned total, int f, int j, float2 (&vin)[NE])
{
    unsigned gf = quad * 4 + f;
    gf = gf < total ? gf : total - 1;
    const unsigned clip = gf / a.n_frames;
    const unsigned t = gf - clip * a.n_frames;
    // stack_frames (processing.rs:65-129, contract framing): frame t starts at sample t*step
    const float2 *src = reinterpret_cast<const float2 *>(a.x + static_cast<unsigned long long>(clip) * a.ld + t * a.step);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int n = j + 16 * e;
        if (EXACT) vin[e] = src[n];
        else vin[e] = 2 * n < static_cast<int>(a.flen) ? src[n] : make_float2(0.f, 0.f);
    }
}

template <int NE, bool EXACT, bool POW2, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ss_mfcc_c256_mx(const Fast512MArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const unsigned long long t_start = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int f = lane >> 4;  // frame within the quad
    const int j = lane & 15;  // lane within the frame (DPP row)

    // ---- LDS carve: per-wave regions, then the shared read-only table block, then the chunk counter ----
    float *wbase = reinterpret_cast<float *>(smem + wave * kWaveBytes);
    float2 *zh = reinterpret_cast<float2 *>(wbase) + (f & 1) * kZStride;  // this frame's slot of the 2-frame exchange
    float *ptile = wbase + 2 * kZStride * 2;                              // [8][132]
    float *elog = ptile + 8 * kPPitch;                                    // ln(frame energy) [8]
    float *s_tab = reinterpret_cast<float *>(smem + WAVES * kWaveBytes);
    const float2 *s_tw2 = reinterpret_cast<const float2 *>(s_tab + kTabTw2);
    const float2 *s_twn = reinterpret_cast<const float2 *>(s_tab + kTabTwn);
    const float *s_ct = s_tab + kTabCt;
    const float *s_wt = s_tab + kTabWt;
    unsigned *s_next = reinterpret_cast<unsigned *>(s_tab + kTabWt + a.n_mm * 64);

    // chunk range of this workgroup (contiguous, balanced to within one chunk)
    const unsigned total = a.batch * a.n_frames;
    const unsigned quads = (total + 3) / 4;
    const unsigned chunks = (quads + 1) / 2;
    const unsigned c_lo = static_cast<unsigned>(static_cast<unsigned long long>(chunks) * blockIdx.x / gridDim.x);
    const unsigned c_hi = static_cast<unsigned>(static_cast<unsigned long long>(chunks) * (blockIdx.x + 1) / gridDim.x);

    // one float4 per thread brings the whole table block in (global layout == LDS layout)
    {
        const int n4 = (kTabWt + a.n_mm * 64) / 4;
        if (tid < n4) reinterpret_cast<float4 *>(s_tab)[tid] = reinterpret_cast<const float4 *>(a.tab)[tid];
        for (int i = tid + WAVES * 64; i < n4; i += WAVES * 64)
            reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(a.tab)[i];
        if (lane < 24) ptile[(lane / 3) * kPPitch + 129 + lane % 3] = 0.f;  // pad bins stay zero for good
        if (tid == 0) *s_next = c_lo + WAVES;
    }
    // first chunk of this wave; its first quad's loads are in flight across the barrier
    unsigned chunk = c_lo + wave;
    float2 vin[NE];
    if (chunk < c_hi) load_quad<NE, EXACT>(a, chunk * 2, total, f, j, vin);

    const int partner = (lane & 48) | ((16 - j) & 15);  // lane holding Z[256 - k]
    const int paddr = partner << 2;
    const int wbase1 = 34 * (j >> 1) + (j & 1);  // exchange write base (float2 units)
    const int Cc = static_cast<int>(a.n_ceps);
    // |X| = (1/2)|...|: the 1/2 of the untangle is folded into the scale (1/4 for the squared form)
    const float hscale = POW2 ? 0.25f * a.scale : 0.5f * a.scale;
    __syncthreads();
    const unsigned long long t_pro = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned n_done = 0;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    if (a.dbg) t_prev = __builtin_amdgcn_s_memtime();

    while (chunk < c_hi) {
        // claim the next chunk now so that its first quad can be prefetched during this one
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(s_next, 1u);
        next = __builtin_amdgcn_readfirstlane(next);
        const unsigned q0 = chunk * 2;
        const unsigned nq = min(2u, quads - q0);
        ++n_done;
        for (unsigned qi = 0; qi < nq; ++qi) {
            float2 v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = e < NE ? vin[e] : make_float2(0.f, 0.f);  // zero pad, processing.rs:147-156
            if (!(a.ablate & 1)) {
                if (qi + 1 < nq) load_quad<NE, EXACT>(a, q0 + 1, total, f, j, vin);
                else if (next < c_hi) load_quad<NE, EXACT>(a, next * 2, total, f, j, vin);
            }

            SS_STAMP(0)  // loop overhead + input wait + unpack
            // ---- 256-point complex FFT: radix-16, transpose through LDS (two frames at a time), twiddle, radix-16 ----
            fft16_reg(v);
            SS_STAMP(1)  // first radix-16
            float2 u[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = v[r];
            if (lane < 32 && !(a.ablate & 2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zh[wbase1 + 2 * r] = v[r];
                wave_sync();
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(&zh[34 * p + 2 * j]);
                    u[2 * p] = make_float2(t4.x, t4.y);
                    u[2 * p + 1] = make_float2(t4.z, t4.w);
                }
            }
            wave_sync();
            if (lane >= 32 && !(a.ablate & 2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zh[wbase1 + 2 * r] = v[r];
                wave_sync();
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(&zh[34 * p + 2 * j]);
                    u[2 * p] = make_float2(t4.x, t4.y);
                    u[2 * p + 1] = make_float2(t4.z, t4.w);
                }
            }
            wave_sync();
            SS_STAMP(2)  // LDS exchange
#pragma unroll
            for (int r = 1; r < 16; ++r) u[r] = cmul(u[r], (a.ablate & 32) ? make_float2(0.6f, 0.8f) : s_tw2[(r - 1) * 16 + j]);
            fft16_reg(u);  // u[r] = Z[j + 16 r]
            SS_STAMP(3)  // twiddle + second radix-16

            // ---- untangle Z -> X; |X| (processing.rs:168) * 1/N (:180); row sum (feature.rs:216) ----
            float esum = 0.f;
            float *prow = ptile + (qi * 4 + f) * kPPitch;
            // all 16 partner fetches (register 15 - r of lane 16 - j) and the 8 twiddles go out back to back: one LDS wait
            float2 zcs[8], ws[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                zcs[r] = u[15 - r];
                if (!(a.ablate & 4)) zcs[r] = make_float2(bperm(paddr, u[15 - r].x), bperm(paddr, u[15 - r].y));
                ws[r] = (a.ablate & 32) ? make_float2(0.6f, 0.8f) : s_twn[r * 16 + j];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 zk = u[r];
                // lane 0 pairs with itself: Z[256 - 16 r] = own register (16 - r) & 15
                const float2 zc = j == 0 ? u[(16 - r) & 15] : zcs[r];
                const float2 w = ws[r];
                const float2 s = make_float2(zk.x + zc.x, zk.y - zc.y);  // 2 E[k]
                const float2 d = make_float2(zk.x - zc.x, zk.y + zc.y);
                const float2 wd = cmul(w, d);
                const float xa_r = s.x + wd.y, xa_i = s.y - wd.x;  // 2 X[k]
                const float xb_r = s.x - wd.y, xb_i = s.y + wd.x;  // 2 conj X[256-k]
                const float na = xa_r * xa_r + xa_i * xa_i, nb = xb_r * xb_r + xb_i * xb_i;
                const float pa = hscale * (POW2 ? na : __builtin_amdgcn_sqrtf(na));
                const float pb = hscale * (POW2 ? nb : __builtin_amdgcn_sqrtf(nb));
                if (!(a.ablate & 64)) prow[j + 16 * r] = pa;  // only bins <= 128 can carry mel weight (the bank ends at (F+1)/2, feature.rs:69-70)
                esum += pa + pb;
            }
            if (j == 0) {
                // lane 0's pair k = 0 produced X[0] and X[256]; X[128] = conj Z[128] is the one extra bin
                const float2 z = u[8];
                const float n = 4.f * (z.x * z.x + z.y * z.y);
                const float p128 = hscale * (POW2 ? n : __builtin_amdgcn_sqrtf(n));
                prow[128] = p128;
                esum += p128;
            }
            float energy = row16_sum_m(esum);
            energy = energy == 0.f ? kEpsM : energy;  // zero_handling, feature.rs:219
            if (j == 0) elog[qi * 4 + f] = fast_ln(energy);
            SS_STAMP(4)  // untangle + magnitudes + energy
        }
        wave_sync();

        // ---- B1: block-sparse mel product on the matrix pipe (feature.rs:229) ----
        const float *pbp = ptile + (lane & 7) * kPPitch + (lane >> 4);
        const float *wt = s_wt + lane;
        const int mlo = (a.ablate & 8) ? 100 : 0;
        const f32x4 acc0 = mel_tile<18>(wt, pbp, a.ks_lo[0] + mlo, a.ks_hi[0]);
        const f32x4 acc1 = mel_tile<18>(wt, pbp, a.ks_lo[1] + mlo, a.ks_hi[1]);
        const f32x4 acc2 = mel_tile<18>(wt, pbp, a.ks_lo[2] + mlo, a.ks_hi[2]);
        // ---- B2: zero handling (feature.rs:230) + ln (:105);  B3: DCT-II (:120-123) ----
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) o = dct_step(o, s_ct[i * 64 + lane], acc0[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) o = dct_step(o, s_ct[(4 + i) * 64 + lane], acc1[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) o = dct_step(o, s_ct[(8 + i) * 64 + lane], acc2[i]);
        SS_STAMP(5)  // mel + ln + DCT on the matrix pipe
        // ---- B4: scaling + column-0 replacement (feature.rs:126-146), staged coalesced store ----
        float *stage = wbase;  // the exchange region is idle during phase B
        {
            const int fr = lane & 15, g = lane >> 4;
            if (fr < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * g + i;
                    float val = o[i] * a.dct_scale_k;
                    if (c == 0) {
                        if (a.dc_elimination) {
                            val = elog[fr];
                        } else {
                            const unsigned gfr = min(q0 * 4 + fr, total - 1);
                            const unsigned t = gfr % a.n_frames;
                            val = o[i] * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                        }
                    }
                    if (c < Cc) stage[fr * Cc + c] = val;
                }
            }
        }
        wave_sync();
        {
            const unsigned first = q0 * 4;
            const unsigned nfr = min(nq * 4, total - first);
            const int nout = static_cast<int>(nfr) * Cc;
            float *dst = a.out + static_cast<unsigned long long>(first) * Cc;
            if (!(a.ablate & 16))
                for (int i = lane; i < nout; i += 64) dst[i] = stage[i];
        }
        wave_sync();
        SS_STAMP(6)  // staging + store
        chunk = next;
    }
    if (a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + 4ull * (blockIdx.x * WAVES + wave);
        d[0] = t_start;
        d[1] = t_pro;
        d[2] = __builtin_amdgcn_s_memrealtime();
        d[3] = (static_cast<unsigned long long>(n_done) << 32) | __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // XCC_ID
        unsigned long long *sg = a.dbg + 4ull * gridDim.x * WAVES + 8ull * (blockIdx.x * WAVES + wave);
        for (int i = 0; i < 8; ++i) sg[i] = seg[i];
    }
}

template <int kWaves>
hipError_t launch_mx(const Fast512MArgs &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    const size_t lds = static_cast<size_t>(kWaves) * kWaveBytes + static_cast<size_t>(kTabWt + a.n_mm * 64) * sizeof(float) + 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
    if (total == 0) return hipSuccess;
    const unsigned long long chunks = ((total + 3) / 4 + 1) / 2;
    // one workgroup per CU; fewer when there is not at least one chunk per wave
    unsigned long long blocks = (chunks + kWaves - 1) / kWaves;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256);
    if (blocks > cap) blocks = cap;
    const unsigned grid = static_cast<unsigned>(blocks);
    auto go = [&](auto kern, const char *name) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(kWaves * 64), lds};
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), lds, stream, a);
        return hipGetLastError();
    };
    const bool pow2 = a.spectrum_exponent == 2;
    if (a.flen == 320) {
        return pow2 ? go(ss_mfcc_c256_mx<10, true, true, kWaves>, "ss_mfcc_c256_mx<10,true,pow2>")
                    : go(ss_mfcc_c256_mx<10, true, false, kWaves>, "ss_mfcc_c256_mx<10,true>");
    }
    if (a.flen == 512) {
        return pow2 ? go(ss_mfcc_c256_mx<16, true, true, kWaves>, "ss_mfcc_c256_mx<16,true,pow2>")
                    : go(ss_mfcc_c256_mx<16, true, false, kWaves>, "ss_mfcc_c256_mx<16,true>");
    }
    if (a.flen <= 320) {
        return pow2 ? go(ss_mfcc_c256_mx<10, false, true, kWaves>, "ss_mfcc_c256_mx<10,false,pow2>")
                    : go(ss_mfcc_c256_mx<10, false, false, kWaves>, "ss_mfcc_c256_mx<10,false>");
    }
    return pow2 ? go(ss_mfcc_c256_mx<16, false, true, kWaves>, "ss_mfcc_c256_mx<16,false,pow2>")
                : go(ss_mfcc_c256_mx<16, false, false, kWaves>, "ss_mfcc_c256_mx<16,false>");
}

}  // namespace

hipError_t launch_mfcc_c256_mx(const Fast512MArgs &a, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    static const char *w = std::getenv("SS_MX_WAVES");  // A/B knob for occupancy experiments
    if (w && std::atoi(w) == 16) return launch_mx<16>(a, stream, num_cus, info);
    if (w && std::atoi(w) == 8) return launch_mx<8>(a, stream, num_cus, info);
    return launch_mx<12>(a, stream, num_cus, info);
}

}  // namespace ss

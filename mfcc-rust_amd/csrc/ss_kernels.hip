// HIP kernels for the speechsauce hot path on gfx950 (MI355X, wave64).
//
// ss_front_generic<LOG2C>: framing -> (pre-emphasis, window) -> R2C FFT -> magnitude/power ->
// sparse mel -> log -> DCT-II, one launch, any power-of-two fft_points in [32, 4096].
//
//   * A real frame of N = 2C samples is packed as C complex points z[n] = x[2n] + i x[2n+1]
//     and transformed by a Stockham autosort FFT whose butterflies live in registers: every
//     thread owns 16 complex points, so C/16 threads cooperate on a frame and a 256-thread
//     workgroup carries 4096/C frames per pass.  Radix plan: 16, then 16, then C/256 (or 16
//     then C/16 for C < 256): at most three LDS exchanges per frame.
//   * The LDS exchange buffer uses the padded index i + (i >> 4): the stride-16 scatter of the
//     first pass then lands on distinct banks for the 16 lanes of a ds_write_b64 group.
//   * X[k] is untangled from Z[k], Z[C-k] pairwise, magnitudes go to an LDS row, the mel bank is
//     a banded reduction over that row (CSR-like start/len/weights; no MFMA - at most two
//     filters touch a bin), and the DCT-II is a [n_ceps x n_filters] table product.
//   * HBM traffic: the clip samples once (neighbouring frames re-read them through L1/L2) and
//     the features once.
//
//   * fft_points that are not a power of two (the reference takes any length) run the chirp-z (Bluestein) build BLU of the
//     same kernel: a[n] = x[n] c[n], c[n] = exp(-i pi n^2 / N); A = FFT_L(a) with the complex L-point transform above
//     (L >= N + N/2 a power of two); A .* FFT_L(conj c) (host table); the inverse transform as conj FFT_L(conj .);
//     X[k] = c[k] (a * conj c)[k] / L for the N/2 + 1 bins.  Two L-point transforms per frame instead of one N/2-point one:
//     a completeness path, not a fast one.
//
// Reference semantics (file:line relative to the reference checkout) are cited at each stage.
#include "ss_device.h"
#include "ss_fft_reg.h"

namespace ss {

namespace {

constexpr float kEps = 1.1920929e-7f;  // f32::EPSILON, functions.rs:70
constexpr int kBlock = 256;

__device__ __forceinline__ int phys(int i) { return i + (i >> 4); }

// ln(x) from v_log_f32 (log2) with the denormal pre-scale the library form uses; ~1 ulp of log2.
__device__ __forceinline__ float fast_ln(float x)
{
    const bool tiny = x < 1.17549435e-38f;
    const float l = __builtin_amdgcn_logf(tiny ? x * 4294967296.f : x);
    return (l - (tiny ? 32.f : 0.f)) * 0.69314718055994530942f;
}

// Hand-off between the threads of ONE frame.  A frame has C/16 threads; up to 64 of them sit in one wave, whose LDS
// operations execute in order, so only the compiler has to be kept from reordering (as in the dedicated kernels).  Frames
// that span waves (fft_points 4096) need the workgroup barrier.
template <int LOG2C>
__device__ __forceinline__ void frame_sync()
{
    if constexpr ((1 << LOG2C) / 16 <= 64) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// One Stockham pass of radix R with sub-transform length NS already done, on a frame of C points
// held 16 per thread.  `j` is the thread index within the frame (TPF = C/16 threads).
// Loads happen before the barrier-separated stores, so the pass works in place.
template <int LOG2C, int R, int NS, bool kLoad>
__device__ __forceinline__ void stockham_pass(float2 *zbuf, int j, const float2 *__restrict__ tw_c, float2 (&v)[16])
{
    constexpr int C = 1 << LOG2C;
    constexpr int TPF = C / 16 > 0 ? C / 16 : 1;
    constexpr int NB = 16 / R;  // butterflies per thread
    constexpr int STRIDE = C / R;
    if (kLoad) {
        frame_sync<LOG2C>();
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int b = j + TPF * q;
#pragma unroll
            for (int r = 0; r < R; ++r) v[q * R + r] = zbuf[phys(b + r * STRIDE)];
        }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int b = j + TPF * q;
        if (NS > 1) {
            const int k = b & (NS - 1);
            constexpr int TWS = C / (NS * R);  // exp(-2 pi i k r / (NS R)) = tw_c[k r TWS]
#pragma unroll
            for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], tw_c[k * r * TWS]);
        }
        fft_reg<R>(&v[q * R]);
    }
    frame_sync<LOG2C>();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int b = j + TPF * q;
        const int k = b & (NS - 1);
        const int j0 = (b - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) zbuf[phys(j0 + r * NS)] = v[q * R + r];
    }
}

// Full C-point complex FFT of the frame whose pass-1 inputs are already in v
// (v[e] = z[j + e * TPF]).  Result: natural order Z[k] at zbuf[phys(k)] (after a barrier).
template <int LOG2C>
__device__ __forceinline__ void frame_fft(float2 *zbuf, int j, const float2 *__restrict__ tw_c, float2 (&v)[16])
{
    constexpr int C = 1 << LOG2C;
    stockham_pass<LOG2C, 16, 1, false>(zbuf, j, tw_c, v);
    if constexpr (LOG2C >= 8) {
        stockham_pass<LOG2C, 16, 16, true>(zbuf, j, tw_c, v);
        if constexpr (LOG2C > 8) stockham_pass<LOG2C, C / 256, 256, true>(zbuf, j, tw_c, v);
    } else if constexpr (LOG2C > 4) {
        stockham_pass<LOG2C, C / 16, 16, true>(zbuf, j, tw_c, v);
    }
    frame_sync<LOG2C>();
}

template <int LOG2C>
struct Geo {
    static constexpr int C = 1 << LOG2C;
    static constexpr int N = 2 * C;
    static constexpr int F = C + 1;
    static constexpr int TPF = C / 16;           // threads per frame
    static constexpr int FPB = kBlock / TPF;     // frames per workgroup pass
    static constexpr int ZLEN = C + C / 16;      // padded complex buffer
    static constexpr int PLEN = (F + 3) & ~3;    // magnitude row
};

// Untangle Z -> X (real-input FFT of length N from the packed C-point FFT), scale, take
// magnitude / power, store the row to LDS and return this thread's partial row sum.
//   X[k]   = 1/2 [ (Z[k] + conj Z[C-k]) - i w (Z[k] - conj Z[C-k]) ],  w = exp(-2 pi i k / N)
//   X[C-k] = conj( 1/2 [ (Z[k] + conj Z[C-k]) + i w (Z[k] - conj Z[C-k]) ] )
template <int LOG2C>
__device__ __forceinline__ float untangle_row(const float2 *zbuf, float *prow, float2 *stft_row, int j,
                                              const FrontArgs &a, bool mel_mode, bool active)
{
    using G = Geo<LOG2C>;
    float esum = 0.0f;
    for (int k = j; k <= G::C / 2; k += G::TPF) {
        const float2 zk = zbuf[phys(k)];
        const float2 zc = zbuf[phys((G::C - k) & (G::C - 1))];
        const float2 w = a.tw_n[k];
        const float2 s = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));  // E[k]
        const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
        // -i w d  with d = (Z[k] - conj Z[C-k])/2
        const float2 wd = cmul(w, d);
        const float2 t = make_float2(wd.y, -wd.x);
        float2 xa = cadd(s, t);           // X[k]
        float2 xb = csub(s, t);           // conj X[C-k]
        xb.y = -xb.y;
        float pa, pb;
        if (mel_mode) {
            // functions.rs:166-169 (* wnorm) then feature.rs:164 (abs().powi(2))
            xa.x *= a.scale; xa.y *= a.scale;
            xb.x *= a.scale; xb.y *= a.scale;
            pa = xa.x * xa.x + xa.y * xa.y;
            pb = xb.x * xb.x + xb.y * xb.y;
            if (stft_row && active) {
                stft_row[k] = xa;
                if (k != G::C / 2) stft_row[G::C - k] = xb;
            }
        } else {
            // processing.rs:168 sqrt(re^2 + im^2), :180 * (1/N)
            const float ma = __builtin_amdgcn_sqrtf(xa.x * xa.x + xa.y * xa.y);
            const float mb = __builtin_amdgcn_sqrtf(xb.x * xb.x + xb.y * xb.y);
            pa = a.spectrum_exponent == 2 ? a.scale * (ma * ma) : a.scale * ma;
            pb = a.spectrum_exponent == 2 ? a.scale * (mb * mb) : a.scale * mb;
        }
        prow[k] = pa;
        esum += pa;
        if (k != G::C / 2) {
            prow[G::C - k] = pb;
            esum += pb;
        }
    }
    return esum;
}

// Chirp-z epilogue: zbuf holds FFT_L(conj(A .* B)) in natural order; X[k] = c[k] conj(zbuf[k]) / L for k < F.  Same scaling,
// magnitude / power and stft output as untangle_row.
template <int LOG2C>
__device__ __forceinline__ float blu_row(const float2 *zbuf, float *prow, float2 *stft_row, int j, const FrontArgs &a, bool mel_mode,
                                         bool active, int F)
{
    using G = Geo<LOG2C>;
    const float inv_l = 1.0f / static_cast<float>(G::C);
    float esum = 0.0f;
    for (int k = j; k < F; k += G::TPF) {
        const float2 w = zbuf[phys(k)];
        const float2 c = a.blu_c[k];
        float2 xa = cmul(c, make_float2(w.x * inv_l, -w.y * inv_l));
        float pa;
        if (mel_mode) {
            xa.x *= a.scale;
            xa.y *= a.scale;
            pa = xa.x * xa.x + xa.y * xa.y;
            if (stft_row && active) stft_row[k] = xa;
        } else {
            const float ma = __builtin_amdgcn_sqrtf(xa.x * xa.x + xa.y * xa.y);
            pa = a.spectrum_exponent == 2 ? a.scale * (ma * ma) : a.scale * ma;
        }
        prow[k] = pa;
        esum += pa;
    }
    return esum;
}

// After the forward transform of the chirped frame: conj(A[k] B[k]) back into the pass-1 registers (v[e] = z[j + e TPF]) and
// the second transform.  The first pass of frame_fft stores only after a frame-wide sync, so the reads here are safe.
template <int LOG2C>
__device__ __forceinline__ void blu_convolve(float2 *zbuf, int j, const FrontArgs &a, float2 (&v)[16])
{
    using G = Geo<LOG2C>;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = j + e * G::TPF;
        const float2 p = cmul(zbuf[phys(k)], a.blu_b[k]);
        v[e] = make_float2(p.x, -p.y);
    }
    frame_fft<LOG2C>(zbuf, j, a.tw_c, v);
}

// Banded mel reduction of one magnitude row (feature.rs:229 / :173 restricted to the non-zero taps).
__device__ __forceinline__ float mel_dot(const float *prow, const FrontArgs &a, int m)
{
    const int st = a.f_start[m], ln = a.f_len[m];
    const float *w = a.f_w + a.f_off[m];
    float s = 0.0f;
    for (int i = 0; i < ln; ++i) s = fmaf(w[i], prow[st + i], s);
    return s;
}

template <int LOG2C, bool BLU>
__global__ __launch_bounds__(kBlock) void ss_front_generic(const FrontArgs a)
{
    using G = Geo<LOG2C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int slot = tid / G::TPF;  // frame slot within the workgroup pass
    const int j = tid % G::TPF;     // thread within the frame

    const int M = static_cast<int>(a.n_filters);
    const int mpad = (M + 3) & ~3;
    // per-slot LDS carve: zbuf | prow | frow | red
    const size_t slot_bytes = sizeof(float2) * G::ZLEN + sizeof(float) * (G::PLEN + mpad + G::TPF + 4);
    unsigned char *sbase = smem_raw + slot_bytes * slot;
    float2 *zbuf = reinterpret_cast<float2 *>(sbase);
    float *prow = reinterpret_cast<float *>(sbase + sizeof(float2) * G::ZLEN);
    float *frow = prow + G::PLEN;
    float *red = frow + mpad;
    // MEL mode: transposed output tile [M][rows_tile + 1] after all slots
    float *tile = reinterpret_cast<float *>(smem_raw + slot_bytes * G::FPB);

    const bool mel_mode = a.out_kind == OUT_MEL || a.out_kind == OUT_STFT;
    const int F = BLU ? static_cast<int>(a.blu_n / 2 + 1) : G::F;  // bins per row

    if (!mel_mode) {
        // ---------------- MFCC / MFE / power-spectrum path: flat list of B*T frames ----------------
        const unsigned long long total = static_cast<unsigned long long>(a.batch) * a.n_frames;
        const unsigned long long groups = (total + G::FPB - 1) / G::FPB;
        for (unsigned long long g = blockIdx.x; g < groups; g += gridDim.x) {
            const unsigned long long gf = g * G::FPB + slot;
            const bool active = gf < total;
            const unsigned gf32 = static_cast<unsigned>(gf);  // launch_one rejects batches with >= 2^32 frames
            const unsigned clip = active ? gf32 / a.n_frames : 0u;
            const unsigned t = active ? gf32 - clip * a.n_frames : 0u;
            const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
            // stack_frames (processing.rs:65-129, contract framing) + zero pad to N (:147-156)
            const unsigned base = (a.frame_mode == FRAME_NORMAL || a.frame_mode == FRAME_PADDED) ? t * a.step : 0u;
            const unsigned lim = a.frame_mode == FRAME_ZERO ? 0u : (a.frame_mode == FRAME_FIRST ? (a.flen & ~1u) : a.flen);
            // sample i of the frame after framing, fused pre-emphasis and the optional window (zero beyond the frame)
            auto sample = [&](unsigned i) -> float {
                float val = 0.0f;
                if (active && i < lim) {
                    unsigned idx = base + i;
                    // FRAME_PADDED (stack_frames zero_padding = true, processing.rs:85-97): zeros past the signal
                    bool inside = a.frame_mode != FRAME_PADDED || idx < a.n_samples;
                    if (a.frame_mode == FRAME_CENTER) {
                        // librosa center=True: the frame is centred on t*step; outside the clip np.pad 'reflect'
                        // (mirror without repeating the edge sample) or zeros
                        long long pos = static_cast<long long>(t) * a.step + i - a.flen / 2;
                        const long long ns = a.n_samples;
                        if (pos < 0 || pos >= ns) {
                            if (a.pad_reflect) pos = pos < 0 ? -pos : 2 * (ns - 1) - pos;
                            else inside = false;
                        }
                        idx = static_cast<unsigned>(pos);
                    }
                    if (inside) {
                        val = xc[idx];
                        if (a.preemph != 0.0f) {  // processing.rs:31-53 fused
                            const unsigned sh = a.preemph_shift % a.n_samples;
                            const unsigned jdx = idx >= sh ? idx - sh : idx + a.n_samples - sh;
                            val -= a.preemph * xc[jdx];
                        }
                    }
                    if (a.window) val *= a.window[i];
                }
                return val;
            };
            float2 v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned n = static_cast<unsigned>(j + e * G::TPF);
                if (BLU) {  // chirp-z: one real sample per complex point, times the chirp
                    const float xv = sample(n);
                    const float2 c = n < a.blu_n ? a.blu_c[n] : make_float2(0.f, 0.f);
                    v[e] = make_float2(xv * c.x, xv * c.y);
                } else {
                    v[e] = make_float2(sample(2 * n), sample(2 * n + 1));
                }
            }
            frame_fft<LOG2C>(zbuf, j, a.tw_c, v);
            float part;
            if (BLU) {
                blu_convolve<LOG2C>(zbuf, j, a, v);
                part = blu_row<LOG2C>(zbuf, prow, nullptr, j, a, false, active, F);
            } else {
                part = untangle_row<LOG2C>(zbuf, prow, nullptr, j, a, false, active);
            }
            red[j] = part;
            frame_sync<LOG2C>();  // zbuf / prow / frow / red are private to the frame's slot

            if (a.out_kind == OUT_POWER) {
                if (active) {
                    float *dst = a.out0 + gf * F;
                    for (int k = j; k < F; k += G::TPF) dst[k] = prow[k];
                }
            } else {
                // feature.rs:216-219: frame energy + zero handling (deterministic serial sum)
                float energy = 0.0f;
                if (j == 0) {  // only the thread that owns coefficient 0 / the energy output needs it
                    for (int i = 0; i < G::TPF; ++i) energy += red[i];
                    energy = energy == 0.0f ? kEps : energy;
                }
                // feature.rs:229-230 banded; zero handling
                for (int m = j; m < M; m += G::TPF) {
                    float s = mel_dot(prow, a, m);
                    s = s == 0.0f ? kEps : s;
                    if (a.out_kind == OUT_MFE) {
                        if (active) a.out0[gf * M + m] = s;
                    } else {
                        frow[m] = fast_ln(s);  // feature.rs:105
                    }
                }
                if (a.out_kind == OUT_MFE) {
                    if (active && j == 0) a.out1[gf] = energy;
                } else {
                    frame_sync<LOG2C>();  // zbuf / prow / frow / red are private to the frame's slot
                    // feature.rs:120-146: DCT-II (first n_ceps outputs), scaling, column-0 replacement
                    const int Cc = static_cast<int>(a.n_ceps);
                    const int parts = G::TPF / Cc;  // threads per coefficient (uniform)
                    if (parts >= 2) {
                        // wide frames (many threads, long filter rows): split every coefficient's sum over `parts`
                        // threads, then combine the partials in fixed order (deterministic)
                        const int c = j % Cc, part = j / Cc;
                        const int ms = (M + parts - 1) / parts;
                        float s = 0.0f;
                        if (part < parts) {
                            const float *row = a.dct + c * M;
                            const int m1 = min(M, (part + 1) * ms);
                            for (int m = part * ms; m < m1; ++m) s = fmaf(frow[m], row[m], s);
                            red[j] = s;
                        }
                        frame_sync<LOG2C>();  // zbuf / prow / frow / red are private to the frame's slot
                        if (j < Cc) {
                            float tot = 0.0f;
                            for (int p = 0; p < parts; ++p) tot += red[p * Cc + j];
                            float o;
                            if (j == 0) o = a.dc_elimination ? fast_ln(energy) : tot * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                            else o = tot * a.dct_scale_k;
                            if (active) a.out0[gf * Cc + j] = o;
                        }
                    } else {
                        for (int c = j; c < Cc; c += G::TPF) {
                            const float *row = a.dct + c * M;
                            float s = 0.0f;
                            for (int m = 0; m < M; ++m) s = fmaf(frow[m], row[m], s);
                            float o;
                            if (c == 0) o = a.dc_elimination ? fast_ln(energy) : s * (t == 0 ? a.dct_scale_00 : a.dct_scale_0);
                            else o = s * a.dct_scale_k;
                            if (active) a.out0[gf * Cc + c] = o;
                        }
                    }
                }
            }
            frame_sync<LOG2C>();  // zbuf / prow / frow / red are private to the frame's slot
        }
    } else {
        // ---------------- STFT / mel-spectrogram path: one clip (channel) per workgroup visit -------
        const int R = static_cast<int>(a.rows);
        const int Rreal = static_cast<int>(a.real_rows);
        const int W = BLU ? static_cast<int>(a.blu_n) : G::N;
        constexpr int TILE = 32;  // rows buffered before a transposed, coalesced flush
        for (unsigned clip = blockIdx.x; clip < a.batch; clip += gridDim.x) {
            const float *xc = a.x + static_cast<unsigned long long>(clip) * a.ld;
            for (int r0 = 0; r0 < R; r0 += TILE) {
                const int rt = min(TILE, R - r0);
                for (int rp = 0; rp < rt; rp += G::FPB) {
                    const int rl = rp + slot;       // row within the tile
                    const int r = r0 + rl;          // output row
                    const bool active = rl < rt && r < Rreal;
                    // functions.rs:137-151: window over the last W samples ending at chunk r + n_pad
                    const long long start = static_cast<long long>(r + a.n_pad + 1) * a.hop - W;
                    auto wsample = [&](int i) -> float {
                        const long long idx = start + i;
                        return active && i < W && idx >= 0 && idx < static_cast<long long>(a.n_samples) ? xc[idx] * a.window[i] : 0.0f;
                    };
                    float2 v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = j + e * G::TPF;
                        if (BLU) {
                            const float xv = wsample(n);
                            const float2 c = n < W ? a.blu_c[n] : make_float2(0.f, 0.f);
                            v[e] = make_float2(xv * c.x, xv * c.y);
                        } else {
                            v[e] = make_float2(wsample(2 * n), wsample(2 * n + 1));
                        }
                    }
                    frame_fft<LOG2C>(zbuf, j, a.tw_c, v);
                    float2 *stft_row = nullptr;
                    if (a.out_kind == OUT_STFT)
                        stft_row = reinterpret_cast<float2 *>(a.out0) + (static_cast<unsigned long long>(clip) * R + r) * F;
                    if (BLU) {
                        blu_convolve<LOG2C>(zbuf, j, a, v);
                        blu_row<LOG2C>(zbuf, prow, stft_row, j, a, true, active, F);
                    } else {
                        untangle_row<LOG2C>(zbuf, prow, stft_row, j, a, true, active);
                    }
                    frame_sync<LOG2C>();  // prow is private to the frame; the shared tile has its own barriers below
                    if (a.out_kind == OUT_MEL && rl < rt) {
                        // feature.rs:173: out[n,m,t] = sum_f P[n,t,f] fb[m,f]; rows >= real_rows stay zero
                        for (int m = j; m < M; m += G::TPF) tile[m * (TILE + 1) + rl] = active ? mel_dot(prow, a, m) : 0.0f;
                    }
                    if (a.out_kind == OUT_STFT && rl < rt && r >= Rreal) {
                        for (int k = j; k < F; k += G::TPF) stft_row[k] = make_float2(0.0f, 0.0f);
                    }
                    __syncthreads();
                }
                if (a.out_kind == OUT_MEL) {
                    float *dst = a.out0 + static_cast<unsigned long long>(clip) * M * R;
                    for (int i = tid; i < M * rt; i += kBlock) {
                        const int m = i / rt, rl = i - m * rt;
                        dst[static_cast<unsigned long long>(m) * R + r0 + rl] = tile[m * (TILE + 1) + rl];
                    }
                    __syncthreads();
                }
            }
        }
    }
}

template <int LOG2C>
size_t front_lds_bytes(const FrontArgs &a)
{
    using G = Geo<LOG2C>;
    const size_t mpad = (a.n_filters + 3) & ~3u;
    const size_t slot_bytes = sizeof(float2) * G::ZLEN + sizeof(float) * (G::PLEN + mpad + G::TPF + 4);
    size_t total = slot_bytes * G::FPB;
    if (a.out_kind == OUT_MEL) total += sizeof(float) * a.n_filters * 33;
    return (total + 15) & ~static_cast<size_t>(15);
}

template <int LOG2C, bool BLU>
hipError_t launch_one(const FrontArgs &a, hipStream_t stream, int num_cus, LaunchInfo *info, const char *name)
{
    using G = Geo<LOG2C>;
    const size_t lds = front_lds_bytes<LOG2C>(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ss_front_generic<LOG2C, BLU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    unsigned long long work;
    if (a.out_kind == OUT_MEL || a.out_kind == OUT_STFT) work = a.batch;
    else work = (static_cast<unsigned long long>(a.batch) * a.n_frames + G::FPB - 1) / G::FPB;
    if (work == 0) return hipSuccess;
    if (static_cast<unsigned long long>(a.batch) * a.n_frames >= 0xffffffffull) return hipErrorInvalidValue;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256) * 8;
    const unsigned grid = static_cast<unsigned>(work < cap ? work : cap);
    if (info) *info = LaunchInfo{name, grid, static_cast<unsigned>(kBlock), lds};
    hipLaunchKernelGGL((ss_front_generic<LOG2C, BLU>), dim3(grid), dim3(kBlock), lds, stream, a);
    return hipGetLastError();
}

__global__ void ss_preemphasis_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n, size_t shift, float cof)
{
    // processing.rs:31-53: y[i] = x[i] - cof * x[(i - shift) mod n]
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t jx = i >= shift ? i - shift : i + n - shift;
        y[i] = x[i] - cof * x[jx];
    }
}

#if SS_LAB
// (lab library only: ss_debug_poison_lds)
// One workgroup takes a CU's whole LDS (160 KB), so a grid of several workgroups per CU sweeps every CU several times.
__global__ __launch_bounds__(256) void ss_poison_lds_kernel(unsigned words)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned *w = reinterpret_cast<unsigned *>(smem_raw);
    for (unsigned i = threadIdx.x; i < words; i += 256) w[i] = 0xFFFFFFFFu;
    __syncthreads();
    // keep the stores observable
    if (w[(threadIdx.x * 97u) % words] != 0xFFFFFFFFu) __builtin_trap();
}
#endif

}  // namespace

#if SS_LAB
hipError_t launch_poison_lds(hipStream_t stream, int num_cus)
{
    const size_t lds = 160 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ss_poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    const unsigned grid = static_cast<unsigned>(num_cus > 0 ? num_cus : 256) * 4;
    hipLaunchKernelGGL(ss_poison_lds_kernel, dim3(grid), dim3(256), lds, stream, static_cast<unsigned>(lds / 4));
    return hipGetLastError();
}
#endif

hipError_t launch_front_generic(const FrontArgs &a, uint32_t log2c, hipStream_t stream, int num_cus, LaunchInfo *info)
{
    if (a.blu_n) {
        switch (log2c) {
            case 4: return launch_one<4, true>(a, stream, num_cus, info, "ss_front_generic<4,chirpz>");
            case 5: return launch_one<5, true>(a, stream, num_cus, info, "ss_front_generic<5,chirpz>");
            case 6: return launch_one<6, true>(a, stream, num_cus, info, "ss_front_generic<6,chirpz>");
            case 7: return launch_one<7, true>(a, stream, num_cus, info, "ss_front_generic<7,chirpz>");
            case 8: return launch_one<8, true>(a, stream, num_cus, info, "ss_front_generic<8,chirpz>");
            case 9: return launch_one<9, true>(a, stream, num_cus, info, "ss_front_generic<9,chirpz>");
            case 10: return launch_one<10, true>(a, stream, num_cus, info, "ss_front_generic<10,chirpz>");
            case 11: return launch_one<11, true>(a, stream, num_cus, info, "ss_front_generic<11,chirpz>");
            case 12: return launch_one<12, true>(a, stream, num_cus, info, "ss_front_generic<12,chirpz>");
            default: return hipErrorInvalidValue;
        }
    }
    switch (log2c) {
        case 4: return launch_one<4, false>(a, stream, num_cus, info, "ss_front_generic<4>");
        case 5: return launch_one<5, false>(a, stream, num_cus, info, "ss_front_generic<5>");
        case 6: return launch_one<6, false>(a, stream, num_cus, info, "ss_front_generic<6>");
        case 7: return launch_one<7, false>(a, stream, num_cus, info, "ss_front_generic<7>");
        case 8: return launch_one<8, false>(a, stream, num_cus, info, "ss_front_generic<8>");
        case 9: return launch_one<9, false>(a, stream, num_cus, info, "ss_front_generic<9>");
        case 10: return launch_one<10, false>(a, stream, num_cus, info, "ss_front_generic<10>");
        case 11: return launch_one<11, false>(a, stream, num_cus, info, "ss_front_generic<11>");
        case 12: return launch_one<12, false>(a, stream, num_cus, info, "ss_front_generic<12>");
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_preemphasis(const float *x, float *y, size_t n, size_t shift, float cof, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const unsigned block = 256;
    size_t blocks = (n + block - 1) / block;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ss_preemphasis_kernel, dim3(static_cast<unsigned>(blocks)), dim3(block), 0, stream, x, y, n, shift, cof);
    return hipGetLastError();
}

}  // namespace ss

// Kernel argument block and launcher declarations shared by ss_kernels.hip and ss_api.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

// Product / lab builds.  The shipped library is the PRODUCT build: kernel selection is a pure function of the configuration
// and the call -- no environment knob is read anywhere, and the timing-attribution switches of the headline kernel
// (SS_ABLATE) are compiled out.  `make lab` (-DSS_LAB=1) builds libspeechsauce_amd_lab.so with the A/B knobs (SS_RES,
// SS_WAVES, SS_MEL_WAVES, SS_HOST_CHUNK_MB, SS_HOST_SMALL_KB, SS_DEBUG_TIMES, SS_DEBUG_ROWS) and the stage-removal builds
// tools/ablate.sh drives, and the process-wide test aids of include/speechsauce_amd_debug.h (LDS poisoning, kernel-selection
// overrides, fault injection, the stamp buffer): those exist in the lab library ONLY.
#ifndef SS_LAB
#define SS_LAB 0
#endif
#if !SS_LAB
#define SS_PRODUCT 1
#endif

namespace ss {

// Process-wide test aids set through include/speechsauce_amd_debug.h (ss_api.hip, lab builds); constants in the product build
#if SS_LAB
bool dbg_force_generic();        // ss_debug_force_generic: every configuration on the generic kernel
bool dbg_mel_tile_off();         // ss_debug_mel_tile(0): eight waves, direct stores instead of the whole-line tile
int dbg_mel_build();             // 0 automatic, 1 = the above, 2 eight-wave builds only, 3 the twelve-wave build where it exists
unsigned dbg_tile_spin_limit();  // polls before a tile hand-off counts as a protocol error (ss_debug_tile_fault: 0)
#else
constexpr bool dbg_force_generic() { return false; }
constexpr bool dbg_mel_tile_off() { return false; }
constexpr int dbg_mel_build() { return 0; }
constexpr unsigned dbg_tile_spin_limit() { return 1u << 24; }
#endif

enum OutKind : int32_t {
    OUT_MFCC = 0,   // [frames x num_cepstral]                    feature.rs:99-148
    OUT_MFE = 1,    // feat [frames x M] + energy [frames]        feature.rs:200-233
    OUT_POWER = 2,  // P [frames x F]                             processing.rs:179-181
    OUT_MEL = 3,    // [clips x M x R]                            feature.rs:151-174
    OUT_STFT = 4    // [clips x R x F x 2]                        functions.rs:86-123
};

enum FrameMode : int32_t { FRAME_NORMAL = 0, FRAME_ZERO = 1, FRAME_FIRST = 2, FRAME_CENTER = 3, FRAME_PADDED = 4 };

struct FrontArgs {
    // input: `batch` clips of `n_samples`, row stride `ld` elements
    const float *x;
    unsigned long long ld;
    uint32_t n_samples;
    uint32_t batch;
    // MFCC framing (processing.rs:65-129)
    uint32_t flen, step, n_frames;
    int32_t frame_mode;
    int32_t pad_reflect;  // FRAME_CENTER: 1 = np.pad 'reflect' outside the clip, 0 = zeros
    float preemph;
    uint32_t preemph_shift;
    // STFT framing (functions.rs:86-170)
    uint32_t hop, n_pad, rows, real_rows;
    const float *window;  // MFCC: [flen] or null; STFT: [n_fft]
    float scale;          // MFCC: 1/N (processing.rs:180); STFT: wnorm (config.rs:178)
    int32_t spectrum_exponent;
    // FFT tables
    const float2 *tw_c;  // exp(-2 pi i t / C), t < C
    const float2 *tw_n;  // exp(-2 pi i k / N), k <= C/2
    // chirp-z mode (fft_points not a power of two): the FFT is a complex one of C = blu_len points
    const float2 *blu_c;  // exp(-i pi n^2 / N), n < N
    const float2 *blu_b;  // FFT_C of the wrapped conjugate chirp
    uint32_t blu_n;       // N = fft_points (0: packed power-of-two mode)
    // sparse mel bank
    const int32_t *f_start, *f_len, *f_off;
    const float *f_w;
    uint32_t n_filters;
    // DCT
    const float *dct;  // [n_ceps x n_filters]
    uint32_t n_ceps;
    float dct_scale_k;   // multiplier of columns >= 1 (gain * norm)
    float dct_scale_0;   // multiplier of column 0, row t > 0
    float dct_scale_00;  // multiplier of element [0,0] of each clip
    int32_t dc_elimination;
    // outputs
    int32_t out_kind;
    float *out0;
    float *out1;
};

struct LaunchInfo {
    const char *kernel_name;
    unsigned grid, block;
    size_t lds_bytes;
};

// Generic front-end (any power-of-two fft_points in [32, 4096]; with a.blu_n != 0 the chirp-z build for other lengths, log2c then
// being the length of its complex FFT).
hipError_t launch_front_generic(const FrontArgs &a, uint32_t log2c, hipStream_t stream, int num_cus, LaunchInfo *info);
#if SS_LAB
// Test aid (lab library): every word of every CU's LDS := 0xFFFFFFFF (ss_debug_poison_lds).
hipError_t launch_poison_lds(hipStream_t stream, int num_cus);
#endif
// Element-wise pre-emphasis (processing.rs:31-53).
hipError_t launch_preemphasis(const float *x, float *y, size_t n, size_t shift, float cof, hipStream_t stream);

// Arguments of the fft_points = 512 MFCC kernel (ss_mfcc512.hip).
struct Fast512Args {
    const float *x;
    unsigned long long ld;
    uint32_t n_samples, batch, flen, step, n_frames;
    float scale;
    int32_t spectrum_exponent;
    // one table block, copied verbatim into LDS (layout: ss::fast512_layout in ss_internal.h):
    //   tw2   [15][16] float2  exp(-2 pi i j r / 256), r = 1..15
    //   twn   [8][16]  float2  exp(-2 pi i (j + 16 r) / 512)
    //   cos   [16][52]         row c: cos(pi c (2 m_q + 1) / 2M) for q = slot*16 + lane < 48 (0 for unused q / c >= n_ceps)
    //   start [3][16]  int32   first P bin of the filter owned by (slot, lane)
    //   filt  [3][16]  int32   its filter index (-1: none)
    //   melw  [16][mel_wpitch] lane row: taps of slot 0, 1, 2, each zero-padded to a multiple of 4
    const float *tab;
    int32_t mel_wpitch;  // floats per lane row = 4 * (mel_q4[0] + mel_q4[1] + mel_q4[2])
    int32_t mel_q4[3];   // taps / 4 per slot (lock-step loop lengths)
    uint32_t n_filters, n_ceps;
    float dct_scale_k, dct_scale_0, dct_scale_00;
    int32_t dc_elimination;
    float *out;          // MFCC [frames x n_ceps], or (out_mfe) mel energies [frames x n_filters]
    float *out_energy;   // out_mfe: frame energies [frames] (feature.rs:216-219)
    int32_t out_mfe;     // 1: stop after the mel stage and write mfe's (features, energy) (feature.rs:200-233);
                         // 2: stop after the spectrum and write power_spectrum's rows [frames x 257] (processing.rs:179-181)
    // optional front end (the switches of ss_params; default off, as in the reference's mfcc):
    int32_t win_floats;     // > 0: frame window [flen] behind the mel rows of the table block
    float preemph;          // != 0: y[n] = x[n] - preemph * x[(n - preemph_shift) mod n_samples] (processing.rs:31-53) on load
    uint32_t preemph_shift;
    // librosa-compatible variants (ss_params.framing = SS_FRAMING_CENTER, banks that cover the whole spectrum)
    int32_t center;         // frame t covers x[t*step - flen/2 : t*step + flen/2); flen % 4 == 0
    int32_t pad_reflect;    // center: np.pad 'reflect' outside the clip (else zeros)
    int32_t fullp;          // the table block was built for P rows of all 257 bins
    int32_t paired;         // 40 filters: filter m in slot 2 and 39 - m in slot 0 of one lane, (16 + i, 23 - i) in lanes 2i, 2i + 1 of slot 1;
                            // 2: and the filters of slots 1 / 2 lie inside the slots' first 6 / 2 taps
    unsigned long long *dbg;  // diagnostic runs only: per-wave realtime stamps, or null
    // filled by launch_mfcc_c256: floor(x / n_frames) = umulhi(x, nf_magic) >> nf_shift for x < 2^31 (nf_magic = 0: divide)
    uint32_t nf_magic, nf_shift;
    // filled by launch_mfcc_c256: workgroup b owns quads [b * q_base + min(b, q_rem), + q_base + (b < q_rem)) -- the balanced
    // contiguous split without a division in the kernel's prologue (everything in front of the first loads is start-up latency)
    uint32_t q_base, q_rem;
};

hipError_t launch_mfcc_c256(const Fast512Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);

// Several independent batches in ONE launch (ss_mfcc_batches_device / ss_mel_spectrogram_batches_device): the second argument of
// the kernel builds that take a batch table (MULTI).  Batch b has its own input block x[b] (clips of a.n_samples at row stride
// a.ld) and output block out[b]; its work units -- frame quads (512-point MFCC), frames (4096-point MFCC), row pairs (2048-point
// mel) -- are [uend[b-1], uend[b]) of the launch's unit range (entries past the last batch: 0xffffffff); total[b] = its clips *
// n_frames (the 512-point kernel's last quad of a batch may be partly filled).  Filled by the launch_*_multi functions.
constexpr int kMaxLaunchBatches = 8;
struct BatchTable {
    const float *x[kMaxLaunchBatches];
    float *out[kMaxLaunchBatches];
    uint32_t uend[kMaxLaunchBatches];
    uint32_t total[kMaxLaunchBatches];
};
using Fast512Multi = BatchTable;
// second kernel argument: the batch table of a MULTI build, nothing otherwise
template <bool MULTI>
struct MultiArg {
};
template <>
struct MultiArg<true> {
    BatchTable m;
};
// a: the argument block of one batch (x / out / batch are ignored); d_x / d_out / clips: n_batches <= kMaxLaunchBatches entries.
// hipErrorInvalidValue before the launch: the configuration has no multi-batch build (the caller launches batch by batch).
hipError_t launch_mfcc_c256_multi(const Fast512Args &a, int n_batches, const float *const *d_x, float *const *d_out, const size_t *clips,
                                  hipStream_t stream, int num_cus, LaunchInfo *info);
// whether the kernel has an mfe-output / windowed / pre-emphasised build for this shape (the default bank at flen 320)
bool mfcc_c256_has_mfe(const Fast512Args &a);

// Arguments of the fft_points = 2048 mel-spectrogram kernel (ss_mel2048.hip).
struct Mel2048Args {
    const float *x;
    unsigned long long ld;
    uint32_t n_samples, batch;
    uint32_t hop, n_pad, rows, real_rows;
    float scale;  // wnorm (config.rs:178)
    // one table block, copied verbatim into LDS (layout: ss::mel2048_layout in ss_internal.h)
    const float *tab;
    int32_t mel_wpitch;  // floats per lane weight row
    int32_t mel_q4[4];   // taps / 4 per slot
    uint32_t n_filters;
    float *out;       // [batch][n_filters][rows], or (out_stft) [batch][rows][1025][2]
    int32_t out_stft; // 1: write the scaled complex spectrum stft2 returns (functions.rs:86-123) instead of the mel rows
    int32_t fullp;    // the bank reaches past (F+1)/2: P rows hold every bin (not in the 4096-point kernel)
    // whole-line tile build (ss_mel_c1024<tile>): device view of the config's pinned error word, or null -- a wave sets it
    // when a tile hand-off never came (the host turns it into SS_ERR_DEVICE).  Touched on the cold path only; how long a wave
    // polls before it gives up is the word behind the table block.
    unsigned *ctl;
    // twelve-wave builds, diagnostic (ss_mel_spectrogram_timed_region): two words per wave -- shader cycles lived, 100 MHz ticks
    // lived -- or null
    unsigned long long *stamps;
};

hipError_t launch_mel_c1024(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);
// several blocks of channels in one launch of the twelve-wave mel build (a: one block's arguments; x / out / batch are ignored);
// hipErrorInvalidValue before the launch where the shape has no batch-table build
hipError_t launch_mel_c1024_multi(const Mel2048Args &a, int n_batches, const float *const *d_x, float *const *d_out, const size_t *channels,
                                  hipStream_t stream, int num_cus, LaunchInfo *info);
#if SS_LAB
// the retired whole-line-tile build (tools/experiments/ss_mel2048_tile.hip, linked into the lab library only);
// hipErrorInvalidValue where the shape has no tile build
hipError_t launch_mel_c1024_tile(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);
#endif
// fft_points = 1024 mel-spectrogram kernel (ss_mfcc1024.hip): same argument block, table layout ss::mfcc1024_layout
hipError_t launch_mel_c512(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);
// fft_points = 4096 mel-spectrogram kernel (ss_mfcc4096.hip): same argument block, table layout ss::mfcc4096_layout
// without cosine rows, Vorbis window [4096] behind the mel rows
hipError_t launch_mel_c2048(const Mel2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);

// Arguments of the fft_points = 512 mel-spectrogram kernel (ss_mel512.hip).
struct Mel512Args {
    const float *x;
    unsigned long long ld;
    uint32_t n_samples, batch;
    uint32_t hop, n_pad, rows, real_rows;
    float scale;  // wnorm (config.rs:178)
    const float *tab;    // table block (layout: ss::mel512_layout in ss_internal.h), copied verbatim into LDS
    int32_t mel_wpitch;  // floats per lane weight row
    int32_t mel_q4[5];   // taps / 4 per slot
    int32_t fullp;       // the bank reaches past bin 128: P rows hold all 257 bins
    uint32_t n_filters;
    float *out;          // [batch][n_filters][rows], or (out_stft) [batch][rows][257][2]
    int32_t out_stft;    // 1: write the scaled complex spectrum stft2 returns (functions.rs:86-123) instead of the mel rows
};

hipError_t launch_mel_c256(const Mel512Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);

// Arguments of the fft_points = 2048 MFCC / mfe kernel (ss_mfcc2048.hip).
struct Mfcc2048Args {
    const float *x;
    unsigned long long ld;
    uint32_t n_samples, batch, flen, step, n_frames;
    float scale;  // 1/N (processing.rs:180)
    int32_t spectrum_exponent;
    const float *tab;    // table block (layout: ss::mfcc2048_layout in ss_internal.h), copied verbatim into LDS
    int32_t mel_wpitch;  // floats per lane weight row
    int32_t mel_q4[4];   // taps / 4 per slot
    uint32_t n_filters, n_ceps;
    float dct_scale_k, dct_scale_0, dct_scale_00;
    int32_t dc_elimination;
    int32_t windowed;    // the table block carries a frame window (mfcc_window switch)
    int32_t out_mfe;     // 1: write mfe's (features, energy) instead of the cepstra
    int32_t center;      // librosa center=True framing
    int32_t pad_reflect; // np.pad 'reflect' (else zeros) outside the clip for centred frames
    int32_t fullp;       // the bank reaches past the reference's (F+1)/2: P rows hold every bin up to fft_points/2
    float preemph;       // fused pre-emphasis coefficient (0 = off) and shift (processing.rs:31-53)
    uint32_t preemph_shift;
    float *out;
    float *out_energy;
};

hipError_t launch_mfcc_c1024(const Mfcc2048Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);

// fft_points = 1024 MFCC / mfe kernel (ss_mfcc1024.hip): same argument block, table layout ss::mfcc1024_layout
using Mfcc1024Args = Mfcc2048Args;
hipError_t launch_mfcc_c512(const Mfcc1024Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);

// Arguments of the fft_points = 256 MFCC / mfe kernel (ss_mfcc256.hip): two frames per 256-point complex transform.
struct Mfcc256Args {
    const float *x;
    unsigned long long ld;
    uint32_t n_samples, batch, flen, step, n_frames;
    float scale;  // 1/N (processing.rs:180)
    int32_t spectrum_exponent;
    const float *tab;    // table block (layout: ss::mfcc256_layout in ss_internal.h), copied verbatim into LDS
    int32_t mel_wpitch;  // floats per lane weight row
    int32_t mel_q4[5];   // taps / 4 per slot (three slots in the 256-point kernel, five in the wide-bank 512-point one)
    uint32_t n_filters, n_ceps;
    float dct_scale_k, dct_scale_0, dct_scale_00;
    int32_t dc_elimination;
    int32_t windowed;    // the table block carries a frame window (mfcc_window switch)
    int32_t out_mfe;     // 1: write mfe's (features, energy) instead of the cepstra
    int32_t center;      // wide-bank 512-point kernel only: librosa center=True framing (flen % 4 == 0)
    int32_t pad_reflect; // np.pad 'reflect' (else zeros) outside the clip for centred frames
    float preemph;       // fused pre-emphasis coefficient (0 = off) and shift (processing.rs:31-53)
    uint32_t preemph_shift;
    uint32_t nf_magic, nf_shift;  // set by the launcher: frame -> (clip, t) by multiply-high
    float *out;
    float *out_energy;
};

hipError_t launch_mfcc_c256x2(const Mfcc256Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);
// fft_points = 512 MFCC / mfe with up to 80 filters (ss_mfcc512w.hip): same argument block, table layout ss::mfcc512w_layout
hipError_t launch_mfcc_c256w(const Mfcc256Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);

// Arguments of the fft_points = 4096 MFCC kernel (ss_mfcc4096.hip).
struct Mfcc4096Args {
    const float *x;
    unsigned long long ld;
    uint32_t n_samples, batch, flen, step, n_frames;
    float scale;  // 1/N (processing.rs:180)
    int32_t spectrum_exponent;
    const float *tab;    // table block (layout: ss::mfcc4096_layout in ss_internal.h), copied verbatim into LDS
    int32_t mel_wpitch;  // floats per lane weight row
    int32_t mel_q4[4];   // taps / 4 per slot
    int32_t cos_floats;  // floats of the cosine block in front of the mel rows
    int32_t dct_fold2;   // 1: per-lane cosine rows of the twice-folded DCT (n_filters % 4 == 0, n_ceps <= 43)
    uint32_t n_filters, n_ceps;
    float dct_scale_k, dct_scale_0, dct_scale_00;
    int32_t dc_elimination;
    float preemph;       // fused pre-emphasis coefficient (0 = off) and shift (processing.rs:31-53)
    uint32_t preemph_shift;
    float *out;          // MFCC [frames x n_ceps], or (out_mfe) mel energies [frames x n_filters]
    float *out_energy;   // out_mfe: frame energies [frames]
    int32_t out_mfe;     // 1: stop after the mel stage and write mfe's (features, energy) (feature.rs:200-233)
    const float *window; // optional frame window [flen] in device memory (mfcc_window switch), or null
    float *dbg;  // diagnostic (SS_DEBUG_ROWS): frame 0's P row [1028] + ln(mel) row [256], or null
    // twelve-wave default-shape build, diagnostic (ss_mfcc_timed_region): two words per wave -- shader cycles lived, 100 MHz ticks
    // lived -- or null
    unsigned long long *stamps;
};

hipError_t launch_mfcc_c2048(const Mfcc4096Args &a, hipStream_t stream, int num_cus, LaunchInfo *info);
// several batches in one launch of the twelve-wave default-shape build (see launch_mfcc_c256_multi)
hipError_t launch_mfcc_c2048_multi(const Mfcc4096Args &a, int n_batches, const float *const *d_x, float *const *d_out, const size_t *clips,
                                   hipStream_t stream, int num_cus, LaunchInfo *info);

}  // namespace ss

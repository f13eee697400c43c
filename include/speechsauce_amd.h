/*
 * speechsauce_amd.h -- C ABI of the MI355X-native MFCC / mel-spectrogram hot path.
 *
 * The reference crate (secretsauceai/mfcc-rust, `speechsauce`) has no FFI of its own: the path
 * sits behind plain Rust functions and a PyO3 module.  This header is the `extern "C"` layer a
 * maintainer binds instead of those functions; every entry point cites the reference item it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows the Rust / Python
 * side of the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; f32 everywhere (the crate is f32-only, README.md:17);
 *     row-major, contiguous; batches carry an explicit leading dimension `ld` (in elements).
 *   - the caller allocates outputs (sizes from the query functions); the library owns only the
 *     opaque config handle and its device-resident tables.
 *   - every function returns an ss_status; nothing unwinds across the boundary.  The reference
 *     panics where these return an error (usize underflow processing.rs:101,105; config.rs:162;
 *     functions.rs:136; asserts feature.rs:47-51).
 *   - `*_device` variants take device pointers and a hipStream_t (passed as void*) and are
 *     asynchronous; the others take host pointers and are synchronous (H2D + kernels + D2H).
 *   - a config handle is immutable after creation (no STFT carry-over state, unlike
 *     config.rs:126,130) and may be used from several threads / streams concurrently.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with SS_ERR_HIP.
 */
#ifndef SPEECHSAUCE_AMD_H
#define SPEECHSAUCE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 7 /* 7: ss_mfcc_batches_device, ss_mel_spectrogram_batches_device, ss_mfcc_timed_region; 6: ss_shader_clock_probe; 5: config-free stack_frames entry points, ss_mfcc_shader_clock; the ss_debug_* test aids left the product library */

typedef enum ss_status {
    SS_OK = 0,
    SS_ERR_SHORT_SIGNAL = 1, /* fewer samples than one frame / zero frames (reference: usize underflow panic) */
    SS_ERR_BAD_CONFIG = 2,   /* parameter combination the reference asserts on or underflows with */
    SS_ERR_ARG = 3,          /* null pointer, bad leading dimension, ... */
    SS_ERR_HIP = 4,          /* HIP runtime error or no device (see ss_last_error_string) */
    SS_ERR_UNSUPPORTED = 5,  /* valid in the reference but not built here (fft_points > 8192, or > 2730 and not a power of two) */
    SS_ERR_DEVICE = 6        /* a kernel reported a device-side protocol error through the config's error word: the results of
                                that launch are incomplete (ss_config_device_status) */
} ss_status;

enum { SS_FRAMING_CONTRACT = 0, SS_FRAMING_LITERAL = 1, SS_FRAMING_CENTER = 2, SS_FRAMING_PADDED = 3 };
enum { SS_DCT_REFERENCE = 0, SS_DCT_ORTHO = 1 };
enum { SS_WINDOW_RECT = 0, SS_WINDOW_HANN = 1 /* periodic, functions.rs:349-357 */, SS_WINDOW_VORBIS = 2 };
/* librosa-compatible variants (the reference's stated goal, README.md:3,44) */
enum { SS_MEL_REFERENCE = 0, SS_MEL_SLANEY = 1, SS_MEL_HTK = 2 };
enum { SS_MEL_NORM_NONE = 0, SS_MEL_NORM_SLANEY = 1 };
enum { SS_PAD_REFLECT = 0, SS_PAD_CONSTANT = 1 };

/* DCT-II gain of the un-vendored ndrustfft `nddct2` (feature.rs:123): scipy's un-normalised
 * convention y[k] = 2 * sum x[n] cos(pi k (2n+1) / 2N).  One named constant; "parity unpinned". */
#define SS_DCT2_GAIN 2.0f

/*
 * ss_params: the nine arguments of SpeechConfig::new (config.rs:140-150) one-for-one, followed by
 * the switches of SURVEY.md section 0.  ss_params_default() fills SpeechConfigBuilder::new's
 * defaults (config.rs:35-47) and the reference-mode switches.
 */
typedef struct ss_params {
    uint32_t struct_size;       /* = sizeof(ss_params); checked by ss_config_create */
    uint32_t sample_rate;       /* config.rs:141 */
    uint32_t fft_points;        /* :142  (power of two 32..8192, or any length 16..2730: chirp-z transform) */
    float    frame_length;      /* :143  seconds */
    float    frame_stride;      /* :144  seconds */
    uint32_t num_cepstral;      /* :145 */
    uint32_t num_filters;       /* :146 */
    float    low_frequency;     /* :147 */
    float    high_frequency;    /* :148 */
    int32_t  dc_elimination;    /* :149 */
    /* ---- switches (reference mode = what ss_params_default sets) ---- */
    int32_t  framing;           /* SS_FRAMING_CONTRACT: frames[t,:] = x[t*step : t*step+flen], the documented
                                   contract (processing.rs:55-64).  SS_FRAMING_LITERAL: the exact_chunks copy as
                                   written (processing.rs:110-120), which leaves every frame zero for > 2 frames.
                                   SS_FRAMING_CENTER: librosa center=True -- frame t covers
                                   x[t*step - flen/2 : t*step + flen/2), 1 + n/step frames, edges per pad_mode.
                                   SS_FRAMING_PADDED: stack_frames(zero_padding = true), processing.rs:85-97 --
                                   ceil((L - flen) / step) frames, the last ones reading appended zeros (what the
                                   reference's test_stack_frames calls, lib.rs:50-68) */
    int32_t  spectrum_exponent; /* 1: |X|/N as written (processing.rs:168,180); 2: |X|^2/N (speechpy) */
    int32_t  dct_norm;          /* SS_DCT_REFERENCE: scaling as written (feature.rs:126-131); SS_DCT_ORTHO */
    float    dct2_gain;         /* SS_DCT2_GAIN */
    int32_t  mfcc_window;       /* window on the MFCC frames; reference applies none (feature.rs:203-210) */
    float    preemph_coef;      /* fused pre-emphasis y[n] = x[n] - c*x[(n-shift) mod L] (processing.rs:31-53);
                                   0 = off (reference mfcc() applies none) */
    int32_t  preemph_shift;     /* >= 1 */
    /* ---- librosa-compatible variants (all 0 in reference mode) ---- */
    int32_t  mel_scale;         /* SS_MEL_REFERENCE: feature.rs:36-90 as written (HTK formula, integer bin mapping
                                   floor((F+1) hz / sr)).  SS_MEL_SLANEY / SS_MEL_HTK: librosa.filters.mel -- triangles
                                   in Hz evaluated at the FFT bin frequencies, mel points on the Slaney (htk=False) or
                                   HTK scale */
    int32_t  mel_norm;          /* SS_MEL_NORM_SLANEY: each filter scaled by 2 / (f[m+2] - f[m]) (librosa norm="slaney");
                                   needs a non-reference mel_scale */
    int32_t  pad_mode;          /* SS_FRAMING_CENTER only: how samples outside the clip are read (np.pad reflect / zeros) */
} ss_params;

typedef struct ss_config ss_config; /* opaque; replaces speechsauce::config::SpeechConfig (config.rs:99-131) */

/* ---- configuration ------------------------------------------------------------------------ */

/* SpeechConfigBuilder::new(sample_rate) defaults, config.rs:35-47 (Default = 16 kHz, :133-137). */
int ss_params_default(ss_params *p, uint32_t sample_rate);

/* SpeechConfig::new, config.rs:140-185: derives sizes, builds the Vorbis window, the sparse mel
 * bank (feature.rs:36-90), FFT twiddles and the DCT table, and uploads them to the current HIP
 * device.  Fails with SS_ERR_HIP when no device is usable. */
int ss_config_create(const ss_params *p, ss_config **out);
void ss_config_destroy(ss_config *cfg);
int ss_config_params(const ss_config *cfg, ss_params *out);
/* Device-side status of the asynchronous (*_device) launches made on this config: SS_OK, or SS_ERR_DEVICE if a kernel has
 * reported a protocol error since the last call (the word is cleared).  Call it after synchronising the stream; the
 * host-pointer entry points check it themselves before they return, and every launch on a config with a pending error
 * fails with SS_ERR_DEVICE instead of queueing more work behind a broken one. */
int ss_config_device_status(const ss_config *cfg);

/* Validation and table construction only (no device): what ss_config_create checks. */
int ss_params_validate(const ss_params *p);

/* ---- derived sizes (host only, no device needed) --------------------------------------------- */

/* frame_sample_length / frame_step_size, processing.rs:77-78 */
int ss_frame_sizes(const ss_params *p, size_t *frame_len, size_t *frame_step);
/* numframes with zero_padding=false, processing.rs:101 (what mfe/mfcc use, feature.rs:203-210) */
int ss_num_frames(const ss_params *p, size_t n_samples, size_t *n_frames);
/* STFT geometry: hop = frame_size (config.rs:154), n_pad (functions.rs:96), wnorm (config.rs:178) */
int ss_stft_sizes(const ss_params *p, size_t *hop, size_t *n_pad, float *wnorm);
/* rows returned by stft1/stft2 = ceil(n/hop) (functions.rs:95-98,121); the last n_pad are zero */
int ss_stft_rows(const ss_params *p, size_t n_samples, size_t *rows, size_t *real_rows);

/* ---- tables (host only) ------------------------------------------------------------------- */

/* dense bank [num_filters x (fft_points/2+1)], feature.rs:36-90; idx (may be NULL) gets the
 * num_filters+2 bin indices of feature.rs:69-70 */
int ss_filterbank(const ss_params *p, float *fb, int32_t *idx);
/* SpeechConfig.window (Vorbis), config.rs:151-160; n = fft_points */
int ss_vorbis_window(size_t n, float *w);

/* ---- hot path, host pointers (synchronous) -------------------------------------------------- */

/* speechsauce::feature::mfcc(ArrayView1<f32>, &SpeechConfig) -> Array2<f32>  (feature.rs:99-148)
 * out: [n_frames x num_cepstral] */
int ss_mfcc(const ss_config *cfg, const float *x, size_t n_samples, float *out);
/* speechsauce::feature::mfe -> (Array2<f32>, Array1<f32>)  (feature.rs:200-233)
 * feat: [n_frames x num_filters], energy: [n_frames] */
int ss_mfe(const ss_config *cfg, const float *x, size_t n_samples, float *feat, float *energy);
/* mel_spectrogram1 (channels = 1) / mel_spectrogram2 (feature.rs:151-174); x: [channels x n_samples],
 * out: [channels x num_filters x rows] */
int ss_mel_spectrogram(const ss_config *cfg, const float *x, size_t channels, size_t n_samples, float *out);
/* speechsauce::processing::preemphasis (processing.rs:31-53) */
int ss_preemphasis(const float *x, size_t n_samples, long shift, float cof, float *y);
/* speechsauce::functions::stft1 (channels = 1, functions.rs:199-233) / stft2 (functions.rs:86-123) -> Array2 / Array3<Complex32>:
 * x [channels x n_samples]; out: interleaved re, im  [channels x rows x (fft_points/2+1) x 2], rows = ss_stft_rows
 * (the reference slices its n_pad leading rows off, functions.rs:121: the trailing n_pad rows are zero). */
int ss_stft(const ss_config *cfg, const float *x, size_t channels, size_t n_samples, float *out);
/* speechsauce::processing::stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding)
 * (processing.rs:65-129) with the config's framing switch (contract / literal / padded = zero_padding) and frame window
 * (the `filter` argument; mfcc_window switch): frames [n_frames x frame_len], sizes from ss_num_frames / ss_frame_sizes. */
int ss_stack_frames(const ss_config *cfg, const float *x, size_t n_samples, float *frames);
/* The same function with the reference's own argument list and nothing else -- no SpeechConfig, no FFT length (the reference's
 * stack_frames has no FFT dependency: 44.1 kHz x 25 ms frames of round(1102.5) = 1103 samples are fine): contract framing
 * frames[t][i] = x[t * step + i], frame_len = round(sample_rate * frame_length), step likewise (processing.rs:77-78),
 * floor((n - frame_len) / step) frames, or ceil with the tail reading appended zeros when zero_padding != 0 (:85-106).
 * `window`: frame_len floats that multiply every frame -- row 0 of the Array2 the reference's `filter(frame_len)` returns
 * (processing.rs:122-126) -- or NULL.  ss_stack_frames_shape gives the output shape without touching the device. */
int ss_stack_frames_shape(size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride, int zero_padding,
                          size_t *num_frames, size_t *frame_len);
int ss_stack_frames_signal(const float *x, size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride,
                           const float *window, int zero_padding, float *frames);
/* speechsauce::processing::power_spectrum(frames: Array2<f32>, fft_points) -> Array2<f32>  (processing.rs:179-181; fft_spectrum
 * :143-171 zero-pads rows shorter than fft_points): frames [rows x cols], cols <= the config's fft_points;
 * P [rows x (fft_points/2+1)] = |rfft(row)| / fft_points.  Only fft_points of the config is used. */
int ss_power_spectrum_frames(const ss_config *cfg, const float *frames, size_t rows, size_t cols, float *P);
/* the same stage on the frames mfe cuts from a signal (stack_frames + power_spectrum, feature.rs:203-214):
 * P [n_frames x (fft_points/2+1)] */
int ss_power_spectrum(const ss_config *cfg, const float *x, size_t n_samples, float *P);

/* batch of equal-length clips: x [batch x n_samples] with row stride ld >= n_samples.
 * out: [batch x n_frames x num_cepstral] */
int ss_mfcc_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *out);
int ss_mfe_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld,
                 float *feat, float *energy);
/* P [batch x n_frames x (fft_points/2+1)] */
int ss_power_spectrum_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *P);

/* ---- hot path, device pointers (asynchronous on `stream`) ----------------------------------- */

int ss_mfcc_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                         float *d_out, void *stream);
/* Several independent batches per call (a loop over speechsauce::feature::mfcc, feature.rs:99-148, per clip of every batch):
 * d_x / batch / d_out are HOST arrays of n_batches entries -- device pointer of batch b's clips [batch[b] x n_samples] (row
 * stride ld), its clip count, device pointer of its output block [batch[b] x n_frames x num_cepstral].  The arrays are read
 * before the call returns (nothing keeps pointing at them); batches with batch[b] == 0 are skipped.  Where the configuration's
 * kernel takes a batch table (fft_points = 512 with the default frame shape and bank: the headline kernel) up to 8 batches share
 * ONE launch, whose persistent workgroups run over all the batches' frames -- a launch's start-up and its one-unit tail are paid
 * once per call, not once per batch; other configurations are served batch by batch on `stream`.  Results are bit-identical to
 * n_batches separate ss_mfcc_batch_device calls either way. */
int ss_mfcc_batches_device(const ss_config *cfg, size_t n_batches, const float *const *d_x, const size_t *batch, size_t n_samples,
                           size_t ld, float *const *d_out, void *stream);
/* the mel_spectrogram2 form (feature.rs:163-174 per block): block b is [channels[b] x n_samples], its output
 * [channels[b] x num_filters x rows]; served block by block on `stream` */
int ss_mel_spectrogram_batches_device(const ss_config *cfg, size_t n_batches, const float *const *d_x, const size_t *channels,
                                      size_t n_samples, size_t ld, float *const *d_out, void *stream);
int ss_mfe_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                        float *d_feat, float *d_energy, void *stream);
int ss_mel_spectrogram_device(const ss_config *cfg, const float *d_x, size_t channels, size_t n_samples,
                              size_t ld, float *d_out, void *stream);
int ss_preemphasis_device(const float *d_x, size_t n_samples, long shift, float cof, float *d_y, void *stream);
/* power_spectrum over the frames of each clip: [batch x n_frames x (fft_points/2+1)] (processing.rs:179-181) */
int ss_power_spectrum_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples,
                                   size_t ld, float *d_P, void *stream);
/* power_spectrum of a frames matrix [rows x cols] with row stride ld: [rows x (fft_points/2+1)] */
int ss_power_spectrum_frames_device(const ss_config *cfg, const float *d_frames, size_t rows, size_t cols, size_t ld,
                                    float *d_P, void *stream);
/* stft2 (functions.rs:86-123): interleaved re,im  [channels x rows x (fft_points/2+1) x 2] */
int ss_stft_device(const ss_config *cfg, const float *d_x, size_t channels, size_t n_samples, size_t ld,
                   float *d_out, void *stream);
/* stack_frames over a batch of clips: frames [batch x n_frames x frame_len] */
int ss_stack_frames_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                           float *d_frames, void *stream);

/* ss_stack_frames_signal on device pointers (d_window: frame_len floats in device memory, or NULL) */
int ss_stack_frames_signal_device(const float *d_x, size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride,
                                  const float *d_window, int zero_padding, float *d_frames, void *stream);

/* ---- post-processing on the feature matrix (row-major [rows x cols] f32) ---------- */

/* speechsauce::processing::cmvn(ArrayView2<f32>, variance_normalization) -> Array2<f32>  (processing.rs:265-300):
 * subtract the column means; optionally divide by (population std + 2^-30). */
int ss_cmvn(const float *vec, size_t rows, size_t cols, int variance_normalization, float *out);
/* the same for `batch` matrices stored back to back ([batch x rows x cols], e.g. the block ss_mfcc_batch_device wrote) */
int ss_cmvn_batch_device(const float *d_vec, size_t batch, size_t rows, size_t cols, int variance_normalization,
                         float *d_out, void *stream);
/* speechsauce::processing::cmvnw(Array2<f32>, win_size = 301, variance_normalization)  (processing.rs:315-371):
 * sliding-window normalisation over win_size rows of the symmetric-padded matrix (np.pad 'symmetric', util.rs:108-115).
 * SS_ERR_BAD_CONFIG for an even win_size (the reference asserts). */
int ss_cmvnw(const float *vec, size_t rows, size_t cols, size_t win_size, int variance_normalization, float *out);
int ss_cmvnw_batch_device(const float *d_vec, size_t batch, size_t rows, size_t cols, size_t win_size,
                          int variance_normalization, float *d_out, void *stream);
/* speechsauce::processing::derivative_extraction(&Array2<f32>, delta_windows)  (processing.rs:222-254): edge-padded
 * differences along the FEATURE axis, sum_R (R f[c+R] - f[c-R]) / sum_R 2R^2 (the reference's literal arithmetic).
 * Rows are independent, so a [batch x rows x cols] block is passed as batch*rows rows. */
int ss_derivative_extraction(const float *feat, size_t rows, size_t cols, size_t delta_windows, float *out);
int ss_derivative_extraction_device(const float *d_feat, size_t rows, size_t cols, size_t delta_windows, float *d_out,
                                    void *stream);
/* speechsauce::feature::extract_derivative_feature(Array2<f32>) -> Array3<f32>  (feature.rs:253-269):
 * cube [rows x cols x 3] = (feature, derivative_extraction(feature, 2), derivative_extraction(that, 2)) */
int ss_extract_derivative_feature(const float *feat, size_t rows, size_t cols, float *cube);
int ss_extract_derivative_feature_device(const float *d_feat, size_t rows, size_t cols, float *d_cube, void *stream);

/* lmfe (feature.rs:242-245, README.md:14 "log mel filterbank energies"): ln of mfe's zero-handled energies, [frames x
 * num_filters].  d_energy may be NULL (the frame energies mfe also produces are then kept in a stream-ordered temporary). */
int ss_lmfe(const ss_config *cfg, const float *x, size_t n_samples, float *feat);
int ss_lmfe_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *feat);
int ss_lmfe_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                         float *d_feat, float *d_energy, void *stream);
/* in-place natural logarithm of n device floats (the element-wise pass of lmfe; util.rs:372-381) */
int ss_ln_device(float *d_x, size_t n, void *stream);
/* librosa.power_to_db (the reference's stated remaining work, README.md:44-46): 10 log10(max(amin, S)) - 10 log10(max(amin,
 * |ref|)), then floored at max(result) - top_db; top_db < 0 switches the floor off.  amin > 0. */
int ss_power_to_db(const float *s, size_t n, float ref, float amin, float top_db, float *out);
int ss_power_to_db_device(const float *d_s, size_t n, float ref, float amin, float top_db, float *d_out, void *stream);

/* ---- multi-GPU callers below Python (one process or thread per GPU) -------------------------------------
 * Clips are independent, so a batch shards by contiguous blocks with no exchange inside the path: rank r of `world`
 * computes clips [lo, hi) of ss_shard_bounds on its own device with the *_device entry points.  The north-star's "RCCL
 * gather over xGMI of the final [n_frames x n_mfcc] blocks" is ss_gather_features: every rank passes its block of
 * elems_per_rank floats (pad uneven shards to the largest); rank `root` receives [world x elems_per_rank] in rank order
 * (one ncclRecv per peer and one device copy of its own block inside an ncclGroup: every peer has a direct xGMI link to
 * the root, so the blocks arrive concurrently), the other ranks only send (d_out may be NULL there).
 * ss_all_gather_features is the all-to-all form for consumers that need the whole corpus on every GPU (one
 * ncclAllGather; every rank then writes (world - 1) blocks into its HBM).
 * `nccl_comm` is the caller's ncclComm_t and MUST come from the RCCL library this one resolves: the copy already mapped
 * into the process (e.g. torch's bundled librccl.so) is preferred, then librccl.so.1 / librccl.so by name;
 * ss_rccl_library(path) names the library explicitly (before the first collective) -- passing a communicator created by a
 * different RCCL copy is undefined behaviour.  Nothing is linked at build time.  The automatic search runs ONCE per process: if
 * the first collective finds no RCCL (it ran before torch / librccl was mapped), every later collective fails with SS_ERR_HIP
 * as well until ss_rccl_library(path) is called -- load RCCL first, or name it. */
int ss_shard_bounds(size_t n_items, int world, int rank, size_t *lo, size_t *hi);
int ss_rccl_library(const char *path);
int ss_gather_features(void *nccl_comm, const float *d_block, size_t elems_per_rank, float *d_out, int root, int rank,
                       int world, void *stream);
int ss_all_gather_features(void *nccl_comm, const float *d_block, size_t elems_per_rank, float *d_out, void *stream);

/* ---- device / diagnostics ------------------------------------------------------------------- */

int ss_device_count(int *count);
int ss_set_device(int device);
/* name of the kernel the last *_device call on this thread launched (for rocprof cross-checks) */
const char *ss_last_kernel_name(void);
/* Times `iters` back-to-back launches of the MFCC batch kernel with HIP events recorded on `stream`
 * (the stream the kernel runs on) and returns the average launch duration in milliseconds. */
int ss_time_mfcc_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                              float *d_out, void *stream, int iters, float *avg_ms);
int ss_time_mel_spectrogram_device(const ss_config *cfg, const float *d_x, size_t channels, size_t n_samples,
                                   size_t ld, float *d_out, void *stream, int iters, float *avg_ms);

/* Shader clock (GHz) the part held during `launches` launches of the MFCC batch kernel on `stream`: every wave of the kernel
 * writes its lifetime once, in shader cycles and on the constant 100 MHz clock, into a buffer that THIS CALL owns; the result
 * is the mean ratio over the waves of the last launch.  A per-call diagnostic (bench.py's roofline.clock_ghz_measured): no
 * process-wide state, other threads' launches are unaffected.  The stamps exist in the fft_points = 512 kernel:
 * SS_ERR_UNSUPPORTED for configurations served by another kernel. */
int ss_mfcc_shader_clock(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld, float *d_out,
                         void *stream, int launches, float *ghz);

/* A timed region whose duration and shader clock come from the SAME launches (bench.py's secondary.cfg2 / cfg5): `launches`
 * launches of the MFCC batch kernel on `stream`, launch i reading d_x[i % n_x] and writing d_out[i % n_out] (host arrays of device pointers: a
 * ring of inputs larger than the 256 MiB Infinity Cache keeps the samples coming from HBM), HIP events recorded on `stream` around
 * all of them, and the per-wave stamps of the LAST min(stamped, launches, 4096) launches kept, each launch in a slot of its own of
 * a buffer this call owns.  *avg_ms = region / launches; *ghz = (sum of the waves' shader cycles) / (sum of their lifetimes on
 * the 100 MHz clock) over every stamped launch; stamped = 0 times only.  *wall_ms (may be NULL) = host time from the first launch
 * call to the completion of the last launch (the region starts on a synchronised stream; reading the stamps back is not part of
 * it).  SS_ERR_UNSUPPORTED (after timing) where the launches ran on a kernel without stamps.  Blocks until the region has run. */
int ss_mfcc_timed_region(const ss_config *cfg, const float *const *d_x, size_t n_x, size_t batch, size_t n_samples, size_t ld,
                         float *const *d_out, size_t n_out, void *stream, int launches, int stamped, float *avg_ms, float *ghz,
                         float *wall_ms);

/* The same for the mel-spectrogram path.  The wave stamps exist in the 512-point MFCC kernel and in the twelve-wave builds of the
 * 4096-point MFCC kernel (default shape) and of the 2048-point mel kernel (the builds of BASELINE configurations 2 / 4, 5 and 3):
 * with stamped > 0 on any other kernel both functions time the region and then return SS_ERR_UNSUPPORTED. */
int ss_mel_spectrogram_timed_region(const ss_config *cfg, const float *const *d_x, size_t n_x, size_t channels, size_t n_samples,
                                    size_t ld, float *const *d_out, size_t n_out, void *stream, int launches, int stamped,
                                    float *avg_ms, float *ghz, float *wall_ms);

/* Shader clock (GHz) of the device while WHATEVER ELSE runs on it: one wave on `stream` sleeps through a lead-in (a tenth of
 * `micros`, at most 200 us), reads the shader-cycle counter and the constant 100 MHz counter, sleeps (no memory traffic, no LDS,
 * a handful of registers) for about `micros` microseconds and reads both again; *ghz = cycles / time.  Call it on a side stream
 * FIRST (from a thread of its own: the call blocks) and launch the kernels of interest on their stream right behind it: the
 * probe wave is resident before they start, fits beside the persistent workgroups (they leave wave slots free) and sees the
 * clock the part holds under that load (the power cap).  Enqueued BEHIND a backlog of launches it may only start
 * when the backlog has drained and read the idle clock; the call may also return only once the other stream is idle.  Works
 * for every kernel of the library (bench.py's clock_ghz_measured where a kernel has no stamps of its own; checked against the
 * 512-point kernel's own stamps in profiles/r05/clock_probe_check.txt).  10 <= micros <= 1 000 000. */
int ss_shader_clock_probe(void *stream, uint32_t micros, float *ghz);
/* The same probe without the wait: queues the one-wave kernel on `stream` and returns; when the stream has run it,
 * d_words[0] = shader cycles and d_words[1] = ticks of the 100 MHz counter over the counted interval (GHz = d_words[0] /
 * (10 * d_words[1])).  `d_words`: two 64-bit words of device memory.  This is the form to queue AHEAD of the launches to watch. */
int ss_shader_clock_probe_async(void *stream, uint32_t micros, unsigned long long *d_words);

/* Process-wide test aids (LDS poisoning, kernel-selection overrides, fault injection, a stamp buffer) are NOT part of this
 * library: include/speechsauce_amd_debug.h, exported by the lab build libspeechsauce_amd_lab.so only. */

const char *ss_status_string(int status);
const char *ss_last_error_string(void); /* thread-local detail of the last failure */
int ss_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHSAUCE_AMD_H */

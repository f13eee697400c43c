/*
 * ss_oracle.h -- CPU oracle for the speechsauce MFCC / mel-spectrogram hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (mfcc-rust_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the timed CPU
 * stand-in -- never as the thing shipped.
 *
 * It restates, in plain C, the algorithm of the reference crate (Rust, cannot be
 * compiled in this image).  Every function cites the reference file:line it
 * follows (paths relative to the reference checkout).
 *
 * PARITY STATUS: "parity unpinned" for the third-party arithmetic conventions.
 * The reference has no golden vectors or known-answer tests for this path (its
 * tests assert shapes and "no NaN" only: speechsauce/src/lib.rs:50-68,93-134),
 * and the FFT / DCT-II arithmetic lives in crates that are not vendored and not
 * version-pinned (ndrustfft ^0.4.0, realfft ^3.2.0; Cargo.lock is git-ignored):
 *   - forward R2C FFT: unnormalised, exp(-2*pi*i*k*n/N)   (published rustfft convention)
 *   - DCT-II: y[k] = g * sum_n x[n] cos(pi*k*(2n+1)/(2N)) with g = ORC_DCT2_GAIN = 2
 *     (ndrustfft's documented scipy-compatible, un-normalised convention).  g is a
 *     run-time parameter (orc_params.dct2_gain) so a different pin is one number.
 * What IS pinned: every shape the reference tests assert (tests/test_oracle_pins.py),
 * analytic known answers (impulse, zero signal, literal-framing constant output),
 * and an independent numpy restatement (oracle/oracle_np.py) that generated the
 * committed fixtures under tests/golden/.
 *
 * Deviations from the literal reference code, each behind a switch:
 *   D1 framing   : contract framing frames[t,:] = x[t*step : t*step+flen]
 *                  (doc comment speechsauce/src/processing.rs:55-64); the literal
 *                  exact_chunks behaviour (processing.rs:110-120) is ORC_FRAMING_LITERAL.
 *   D2 mel 1-D   : mel_spectrogram1 == mel_spectrogram2 with one channel (feature.rs:151-174).
 *   D3 STFT state: every clip/channel starts from zero analysis memory (functions.rs:125-170
 *                  keeps state across calls in a RefCell).
 */
#ifndef SS_ORACLE_H
#define SS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DCT2_GAIN 2.0f

enum { ORC_OK = 0, ORC_ERR_SHORT_SIGNAL = 1, ORC_ERR_BAD_CONFIG = 2, ORC_ERR_ARG = 3 };
enum { ORC_FRAMING_CONTRACT = 0, ORC_FRAMING_LITERAL = 1, ORC_FRAMING_CENTER = 2, ORC_FRAMING_PADDED = 3 };
enum { ORC_MEL_REFERENCE = 0, ORC_MEL_SLANEY = 1, ORC_MEL_HTK = 2 };   /* librosa-compatible variants, SURVEY 8f-4 */
enum { ORC_MEL_NORM_NONE = 0, ORC_MEL_NORM_SLANEY = 1 };
enum { ORC_PAD_REFLECT = 0, ORC_PAD_CONSTANT = 1 };
enum { ORC_DCT_REFERENCE = 0, ORC_DCT_ORTHO = 1 };
enum { ORC_WINDOW_RECT = 0, ORC_WINDOW_HANN = 1, ORC_WINDOW_VORBIS = 2 };

/* Same field order as ss_params in include/speechsauce_amd.h (own definition). */
typedef struct orc_params {
    uint32_t struct_size;
    uint32_t sample_rate;
    uint32_t fft_points;
    float    frame_length;   /* seconds */
    float    frame_stride;   /* seconds */
    uint32_t num_cepstral;
    uint32_t num_filters;
    float    low_frequency;
    float    high_frequency;
    int32_t  dc_elimination;
    /* switches; zero-initialised + dct2_gain=2 == reference mode */
    int32_t  framing;
    int32_t  spectrum_exponent; /* 1: |X|/N (reference, Q2)  2: |X|^2/N (speechpy) */
    int32_t  dct_norm;
    float    dct2_gain;
    int32_t  mfcc_window;
    float    preemph_coef;      /* 0 = off (reference mfcc() applies none) */
    int32_t  preemph_shift;
    /* librosa-compatible variants (0 = reference mode): restated from librosa's documented algorithms (filters.mel,
     * stft center/pad_mode); librosa is not installed here, so these are "parity unpinned" against it as well */
    int32_t  mel_scale;
    int32_t  mel_norm;
    int32_t  pad_mode;
} orc_params;

void orc_params_default(orc_params *p, uint32_t sample_rate);

/* derived sizes */
int orc_frame_sizes(const orc_params *p, size_t *flen, size_t *step);
int orc_num_frames(const orc_params *p, size_t n_samples, size_t *n_frames);
int orc_num_frames_padded(const orc_params *p, size_t n_samples, size_t *n_frames);
int orc_stft_sizes(const orc_params *p, size_t *hop, size_t *n_pad, float *wnorm);
int orc_stft_rows(const orc_params *p, size_t n_samples, size_t *rows, size_t *real_rows);

/* tables */
void orc_vorbis_window(size_t n, float *w);
void orc_hann_window(size_t win_length, float *w);
int  orc_filterbank(const orc_params *p, float *fb /* M x F */, int32_t *idx /* M+2 */);

/* stages, f64 accumulation on the reference's f32 constants */
int orc_power_spectrum(const orc_params *p, const float *x, size_t n, double *P /* T x F */);
int orc_stack_frames(const orc_params *p, const float *x, size_t n, double *frames /* T x flen */);
int orc_power_spectrum_frames(const float *frames, size_t rows, size_t cols, size_t fft_points, double *P /* rows x F */);
int orc_mfe(const orc_params *p, const float *x, size_t n, double *feat /* T x M */, double *energy /* T */);
int orc_mfcc(const orc_params *p, const float *x, size_t n, double *out /* T x C */);
int orc_stft(const orc_params *p, const float *x, size_t channels, size_t n,
             double *out_re_im /* ch x R x F x 2 */);
int orc_mel_spectrogram(const orc_params *p, const float *x, size_t channels, size_t n,
                        double *out /* ch x M x R */);
int orc_preemphasis(const float *x, size_t n, long shift, float cof, double *y);
/* post-processing (processing.rs:222-371, feature.rs:253-269); matrices row-major [rows x cols] */
int orc_cmvn(const float *vec, size_t rows, size_t cols, int variance_normalization, double *out);
int orc_cmvnw(const float *vec, size_t rows, size_t cols, size_t win_size, int variance_normalization, double *out);
int orc_derivative_extraction(const double *feat, size_t rows, size_t cols, size_t delta_windows, double *out);
int orc_extract_derivative_feature(const float *feat, size_t rows, size_t cols, double *cube);

/* Reference-shaped single-thread f32 port: pass-for-pass the structure of the Rust code
 * (materialised frames, pad copy, per-row FFT, magnitude pass, dense mel GEMM, full DCT).
 * This is the timed CPU stand-in ("cpu_baseline.kind = port"). */
int port_mfcc_f32(const orc_params *p, const float *x, size_t n, float *out /* T x C */);
int port_mfe_f32(const orc_params *p, const float *x, size_t n, float *feat, float *energy);
int port_mel_spectrogram_f32(const orc_params *p, const float *x, size_t channels, size_t n,
                             float *out /* ch x M x R */);

#ifdef __cplusplus
}
#endif
#endif

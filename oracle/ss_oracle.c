/*
 * ss_oracle.c -- CPU oracle (see ss_oracle.h: TEST INFRASTRUCTURE ONLY, "parity unpinned"
 * for the un-vendored FFT/DCT conventions, pinned on the reference's shape tests and
 * analytic known answers).
 *
 * Part 1 (orc_*): f64 accumulation over the reference's f32 constants -- the parity checker.
 * Part 2 (port_*): single-thread f32 port shaped pass-for-pass like the Rust code -- the timed
 *                  CPU baseline.
 *
 * File:line citations refer to the reference checkout (speechsauce/src/...).
 */
#include "ss_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* f32::EPSILON, functions.rs:70 */
#define ORC_EPS_F32 1.1920929e-7f

/* ------------------------------------------------------------------------------------------ */
/* parameters and derived sizes                                                               */
/* ------------------------------------------------------------------------------------------ */

/* config.rs:35-47 (SpeechConfigBuilder::new defaults) */
void orc_params_default(orc_params *p, uint32_t sample_rate)
{
    memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(*p);
    p->sample_rate = sample_rate;
    p->fft_points = 512;
    p->frame_length = 0.02f;
    p->frame_stride = 0.01f;
    p->num_cepstral = 13;
    p->num_filters = 40;
    p->low_frequency = 0.0f;
    p->high_frequency = (float)sample_rate / 2.0f;
    p->dc_elimination = 1;
    p->framing = ORC_FRAMING_CONTRACT;
    p->spectrum_exponent = 1;
    p->dct_norm = ORC_DCT_REFERENCE;
    p->dct2_gain = ORC_DCT2_GAIN;
    p->mfcc_window = ORC_WINDOW_RECT;
    p->preemph_coef = 0.0f;
    p->preemph_shift = 1;
    p->mel_scale = ORC_MEL_REFERENCE;
    p->mel_norm = ORC_MEL_NORM_NONE;
    p->pad_mode = ORC_PAD_REFLECT;
}

/* processing.rs:77-78: (sample_rate as f32 * seconds).round() as usize  (round half away from 0) */
int orc_frame_sizes(const orc_params *p, size_t *flen, size_t *step)
{
    float fl = roundf((float)p->sample_rate * p->frame_length);
    float st = roundf((float)p->sample_rate * p->frame_stride);
    if (!(fl >= 1.0f) || !(st >= 1.0f)) return ORC_ERR_BAD_CONFIG;
    *flen = (size_t)fl;
    *step = (size_t)st;
    return ORC_OK;
}

/* processing.rs:101-106 (zero_padding = false, what mfe passes: feature.rs:203-210).
 * usize underflow for n < flen, and (numframes - 1) underflow for numframes == 0, are panics
 * in the reference -> error codes here. */
int orc_num_frames(const orc_params *p, size_t n, size_t *n_frames)
{
    size_t flen, step;
    int rc = orc_frame_sizes(p, &flen, &step);
    if (rc) return rc;
    if (p->framing == ORC_FRAMING_CENTER) {
        /* librosa center=True: y padded by flen/2 on both sides -> 1 + n / hop frames */
        if (n == 0 || (p->pad_mode == ORC_PAD_REFLECT && n <= flen / 2)) return ORC_ERR_SHORT_SIGNAL;
        *n_frames = 1 + n / step;
        return ORC_OK;
    }
    if (n < flen) return ORC_ERR_SHORT_SIGNAL;
    if (p->framing == ORC_FRAMING_PADDED) {
        /* stack_frames(zero_padding = true), processing.rs:85-97: ceil instead of floor; the frames that reach past the
         * signal read the appended zeros (contract framing otherwise).  What the reference's test_stack_frames calls
         * (lib.rs:50-68: 3124 frames for 1e6 samples, 20 ms / 20 ms @16 kHz). */
        size_t tp = (size_t)ceilf((float)(n - flen) / (float)step);
        if (tp == 0) return ORC_ERR_SHORT_SIGNAL;
        *n_frames = tp;
        return ORC_OK;
    }
    float q = floorf((float)(n - flen) / (float)step);
    size_t t = (size_t)q;
    if (t == 0) return ORC_ERR_SHORT_SIGNAL;
    *n_frames = t;
    return ORC_OK;
}

/* processing.rs:91-92 (zero_padding = true; only the reference's own test uses it, lib.rs:50-68) */
int orc_num_frames_padded(const orc_params *p, size_t n, size_t *n_frames)
{
    size_t flen, step;
    int rc = orc_frame_sizes(p, &flen, &step);
    if (rc) return rc;
    if (n < flen) return ORC_ERR_SHORT_SIGNAL;
    *n_frames = (size_t)ceilf((float)(n - flen) / (float)step);
    return ORC_OK;
}

/* config.rs:154 frame_size = (frame_length * sample_rate as f32) as usize   (truncation)
 * config.rs:178 wnorm = 1 / (fft_points^2 as f32 / (2*frame_size) as f32)
 * functions.rs:96 n_pad = window_size / frame_size - 1
 * config.rs:162 + functions.rs:136: needs fft_points - frame_size >= frame_size (usize underflow
 * otherwise) -> ORC_ERR_BAD_CONFIG. */
int orc_stft_sizes(const orc_params *p, size_t *hop, size_t *n_pad, float *wnorm)
{
    size_t W = p->fft_points;
    size_t H = (size_t)(p->frame_length * (float)p->sample_rate);
    if (H == 0 || W < 2 * H) return ORC_ERR_BAD_CONFIG;
    *hop = H;
    *n_pad = W / H - 1;
    *wnorm = 1.0f / ((float)(W * W) / (float)(2 * H));
    return ORC_OK;
}

/* functions.rs:95-98,121: tfd = ceil(T/H) + n_pad rows allocated, first n_pad dropped ->
 * ceil(T/H) rows returned; only ceil(T/H) chunks are analysed (zip stops at the shorter side),
 * so the last n_pad returned rows are never written. */
int orc_stft_rows(const orc_params *p, size_t n, size_t *rows, size_t *real_rows)
{
    size_t H, n_pad;
    float wn;
    int rc = orc_stft_sizes(p, &H, &n_pad, &wn);
    if (rc) return rc;
    size_t chunks = (size_t)ceilf((float)n / (float)H);
    *rows = chunks;
    *real_rows = chunks > n_pad ? chunks - n_pad : 0;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* tables                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* config.rs:151-160: w[i] = sin(pi/2 * sin^2(pi*(i+0.5)/N)), f64 then cast */
void orc_vorbis_window(size_t n, float *w)
{
    size_t half = n / 2;
    for (size_t i = 0; i < n; ++i) {
        double s = sin(0.5 * M_PI * ((double)i + 0.5) / (double)half);
        w[i] = (float)sin(0.5 * M_PI * s * s);
    }
}

/* functions.rs:349-357 (commented-out periodic Hann; offered as an option, never the default) */
void orc_hann_window(size_t win_length, float *w)
{
    for (size_t i = 0; i < win_length; ++i)
        w[i] = (float)(0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)win_length)));
}

/* functions.rs:19-21 */
static float hz_to_mel(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
/* functions.rs:36-41 */
static float mel_to_hz(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

/* feature.rs:36-90 with functions.rs:43-60 (triangle).  All f32, glibc logf/expf, one fixed op
 * order (the top index sits on an integer boundary, so the op order matters). */
int orc_filterbank(const orc_params *p, float *fb, int32_t *idx_out)
{
    const size_t M = p->num_filters;
    const size_t F = p->fft_points / 2 + 1;
    const float sr = (float)p->sample_rate;
    if (M == 0) return ORC_ERR_BAD_CONFIG;
    if (p->high_frequency > sr / 2.0f) return ORC_ERR_BAD_CONFIG; /* feature.rs:47-50 assert */
    if (p->low_frequency < 0.0f) return ORC_ERR_BAD_CONFIG;       /* feature.rs:51 assert */
    if (p->mel_norm == ORC_MEL_NORM_SLANEY && p->mel_scale == ORC_MEL_REFERENCE) return ORC_ERR_BAD_CONFIG;
    if (p->mel_scale != ORC_MEL_REFERENCE) {
        /* librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk, norm): triangles in Hz evaluated at the rfft bin
         * frequencies; mel points on the Slaney scale (linear below 1 kHz, log above: 27 steps per factor 6.4) or HTK */
        if (!(p->high_frequency > p->low_frequency)) return ORC_ERR_BAD_CONFIG;
        const int htk = p->mel_scale == ORC_MEL_HTK;
        const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
        double *mel_f = (double *)malloc((M + 2) * sizeof(double));
        if (!mel_f) return ORC_ERR_ARG;
        double lo = p->low_frequency, hi = p->high_frequency, mlo, mhi;
        if (htk) { mlo = 2595.0 * log10(1.0 + lo / 700.0); mhi = 2595.0 * log10(1.0 + hi / 700.0); }
        else {
            mlo = lo >= min_log_hz ? min_log_mel + log(lo / min_log_hz) / logstep : lo / f_sp;
            mhi = hi >= min_log_hz ? min_log_mel + log(hi / min_log_hz) / logstep : hi / f_sp;
        }
        for (size_t i = 0; i < M + 2; ++i) {
            double m = mlo + (mhi - mlo) * (double)i / (double)(M + 1);
            if (htk) mel_f[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0);
            else mel_f[i] = m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
            if (idx_out) idx_out[i] = 0;
        }
        for (size_t m = 0; m < M; ++m) {
            double enorm = p->mel_norm == ORC_MEL_NORM_SLANEY ? 2.0 / (mel_f[m + 2] - mel_f[m]) : 1.0;
            for (size_t k = 0; k < F; ++k) {
                double f = (double)k * (double)p->sample_rate / (double)p->fft_points;
                double lower = (f - mel_f[m]) / (mel_f[m + 1] - mel_f[m]);
                double upper = (mel_f[m + 2] - f) / (mel_f[m + 2] - mel_f[m + 1]);
                double w = lower < upper ? lower : upper;
                fb[m * F + k] = (float)((w > 0.0 ? w : 0.0) * enorm);
            }
        }
        free(mel_f);
        return ORC_OK;
    }

    /* ndarray linspace: start + step*i with step = (end-start)/(n-1), all f32 (feature.rs:57-61) */
    const float m_lo = hz_to_mel(p->low_frequency);
    const float m_hi = hz_to_mel(p->high_frequency);
    const size_t npts = M + 2;
    const float mstep = (m_hi - m_lo) / (float)(npts - 1);
    size_t *idx = (size_t *)malloc(npts * sizeof(size_t));
    if (!idx) return ORC_ERR_ARG;
    for (size_t i = 0; i < npts; ++i) {
        float mel = m_lo + mstep * (float)i;
        float hz = mel_to_hz(mel);
        float v = (float)(F + 1) * hz / sr; /* feature.rs:70 */
        idx[i] = v > 0.0f ? (size_t)v : 0;  /* `as usize`: truncate, saturate at 0 */
        if (idx_out) idx_out[i] = (int32_t)idx[i];
    }
    memset(fb, 0, M * F * sizeof(float));
    for (size_t i = 0; i < M; ++i) {
        size_t l = idx[i], m = idx[i + 1], r = idx[i + 2];
        if (r < l || r + 1 > F) { free(idx); return ORC_ERR_BAD_CONFIG; } /* slice panic */
        float lf = (float)l, mf = (float)m, rf = (float)r;
        for (size_t x = l; x <= r; ++x) {
            /* z = linspace(l, r, r-l+1) -> exactly the integers l..r in f32 (feature.rs:81) */
            float xf = (float)x, v = 0.0f;
            if (xf >= lf && xf < rf) {                 /* (left..right).contains(x) */
                if (xf <= mf) v = (xf - lf) / (mf - lf);
                if (mf <= xf) v = (rf - xf) / (rf - mf);
            }
            fb[i * F + x] = v;
        }
    }
    free(idx);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* f64 FFT (oracle only)                                                                      */
/* ------------------------------------------------------------------------------------------ */

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

/* in-place iterative radix-2, forward, unnormalised, exp(-i...) */
static void fft_pow2_f64(double *re, double *im, size_t n)
{
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                size_t a = i + k, b = i + k + len / 2;
                double xr = re[b] * wr - im[b] * wi;
                double xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr;        im[a] += xi;
            }
        }
    }
}

/* real input of length n (any n), bins 0..n/2; naive DFT when n is not a power of two */
static void rfft_f64(const double *x, size_t n, double *out_re, double *out_im, double *wre, double *wim)
{
    size_t F = n / 2 + 1;
    if (is_pow2(n)) {
        for (size_t i = 0; i < n; ++i) { wre[i] = x[i]; wim[i] = 0.0; }
        fft_pow2_f64(wre, wim, n);
        for (size_t k = 0; k < F; ++k) { out_re[k] = wre[k]; out_im[k] = wim[k]; }
    } else {
        for (size_t k = 0; k < F; ++k) {
            double sr = 0.0, si = 0.0;
            for (size_t i = 0; i < n; ++i) {
                double a = -2.0 * M_PI * (double)((k * i) % n) / (double)n;
                sr += x[i] * cos(a);
                si += x[i] * sin(a);
            }
            out_re[k] = sr; out_im[k] = si;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Part 1: f64 oracle                                                                         */
/* ------------------------------------------------------------------------------------------ */
/* Post-processing on the feature matrix.  No golden vectors exist in the reference for these
 * either ("parity unpinned"); the pads follow the numpy semantics the reference quotes in its comments
 * (util.rs:108-124: np.pad 'symmetric' / 'edge'), the arithmetic follows the Rust expressions literally. */

/* np.pad(..., 'symmetric') index map along an axis of length n: reflection including the edge sample, period 2n */
static size_t sym_index(long p, size_t n)
{
    long period = 2 * (long)n;
    long m = p % period;
    if (m < 0) m += period;
    return (size_t)(m < (long)n ? m : period - 1 - m);
}

/* processing.rs:265-300: column mean over the rows; optionally divide by (population std + 2^-30) */
int orc_cmvn(const float *vec, size_t rows, size_t cols, int variance_normalization, double *out)
{
    if (rows == 0 || cols == 0) return ORC_ERR_ARG;
    const double eps = 9.313225746154785e-10; /* 2f32.powf(-30.), processing.rs:266 */
    for (size_t c = 0; c < cols; ++c) {
        double mean = 0.0;
        for (size_t r = 0; r < rows; ++r) mean += (double)vec[r * cols + c];
        mean /= (double)rows;
        double var = 0.0;
        for (size_t r = 0; r < rows; ++r) {
            double d = (double)vec[r * cols + c] - mean;
            out[r * cols + c] = d;
            var += d * d;
        }
        if (variance_normalization) {
            /* std_axis(Axis(0), 0.) of the mean-subtracted matrix (its own mean is 0 up to rounding; ndarray subtracts it) */
            double m2 = 0.0;
            for (size_t r = 0; r < rows; ++r) m2 += out[r * cols + c];
            m2 /= (double)rows;
            var = 0.0;
            for (size_t r = 0; r < rows; ++r) {
                double d = out[r * cols + c] - m2;
                var += d * d;
            }
            double sd = sqrt(var / (double)rows);
            for (size_t r = 0; r < rows; ++r) out[r * cols + c] /= (sd + eps);
        }
    }
    return ORC_OK;
}

/* processing.rs:315-371: sliding-window mean (and std) over win_size rows of the symmetric-padded matrix */
int orc_cmvnw(const float *vec, size_t rows, size_t cols, size_t win_size, int variance_normalization, double *out)
{
    if (rows == 0 || cols == 0) return ORC_ERR_ARG;
    if (win_size % 2 != 1) return ORC_ERR_BAD_CONFIG; /* assert!(win_size % 2 == 1), processing.rs:327 */
    const double eps = 9.313225746154785e-10;
    const long pad = (long)((win_size - 1) / 2);
    double *ms = (double *)malloc(rows * cols * sizeof(double));
    if (!ms) return ORC_ERR_ARG;
    for (size_t i = 0; i < rows; ++i)
        for (size_t c = 0; c < cols; ++c) {
            double s = 0.0;
            for (size_t w = 0; w < win_size; ++w) s += (double)vec[sym_index((long)i + (long)w - pad, rows) * cols + c];
            ms[i * cols + c] = (double)vec[i * cols + c] - s / (double)win_size;
        }
    if (!variance_normalization) {
        memcpy(out, ms, rows * cols * sizeof(double));
    } else {
        for (size_t i = 0; i < rows; ++i)
            for (size_t c = 0; c < cols; ++c) {
                double s = 0.0;
                for (size_t w = 0; w < win_size; ++w) s += ms[sym_index((long)i + (long)w - pad, rows) * cols + c];
                const double m = s / (double)win_size;
                double v = 0.0;
                for (size_t w = 0; w < win_size; ++w) {
                    double d = ms[sym_index((long)i + (long)w - pad, rows) * cols + c] - m;
                    v += d * d;
                }
                out[i * cols + c] = ms[i * cols + c] / (sqrt(v / (double)win_size) + eps);
            }
    }
    free(ms);
    return ORC_OK;
}

/* processing.rs:222-254, literally: edge pad along the FEATURE axis, dif_R = R * f[c + R] - f[c - R],
 * out = sum_R dif_R / sum_R 2 R^2 */
int orc_derivative_extraction(const double *feat, size_t rows, size_t cols, size_t delta_windows, double *out)
{
    if (rows == 0 || cols == 0 || delta_windows == 0) return ORC_ERR_ARG; /* scale = 0 -> division by zero */
    double scale = 0.0;
    for (size_t R = 1; R <= delta_windows; ++R) scale += 2.0 * (double)R * (double)R;
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) {
            double acc = 0.0;
            for (size_t R = 1; R <= delta_windows; ++R) {
                size_t hi = c + R < cols ? c + R : cols - 1;
                size_t lo = c >= R ? c - R : 0;
                acc += feat[r * cols + hi] * (double)R - feat[r * cols + lo];
            }
            out[r * cols + c] = acc / scale;
        }
    return ORC_OK;
}

/* feature.rs:253-269: cube [rows x cols x 3] = (feature, derivative(feature, 2), derivative(that, 2)) */
int orc_extract_derivative_feature(const float *feat, size_t rows, size_t cols, double *cube)
{
    if (rows == 0 || cols == 0) return ORC_ERR_ARG;
    double *f0 = (double *)calloc(3 * rows * cols, sizeof(double));
    if (!f0) return ORC_ERR_ARG;
    double *d1 = f0 + rows * cols, *d2 = d1 + rows * cols;
    for (size_t i = 0; i < rows * cols; ++i) f0[i] = (double)feat[i];
    orc_derivative_extraction(f0, rows, cols, 2, d1);
    orc_derivative_extraction(d1, rows, cols, 2, d2);
    for (size_t i = 0; i < rows * cols; ++i) {
        cube[3 * i] = f0[i];
        cube[3 * i + 1] = d1[i];
        cube[3 * i + 2] = d2[i];
    }
    free(f0);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */

/* processing.rs:31-53: y[n] = x[n] - cof * x[(n - shift) mod L]  (np.roll semantics) */
int orc_preemphasis(const float *x, size_t n, long shift, float cof, double *y)
{
    if (n == 0) return ORC_ERR_ARG;
    if (shift <= 0 || (size_t)shift > n) return ORC_ERR_ARG; /* slice panics in the reference */
    for (size_t i = 0; i < n; ++i) {
        size_t j = (i + n - (size_t)shift) % n;
        y[i] = (double)x[i] - (double)cof * (double)x[j];
    }
    return ORC_OK;
}

/* sample of the (optionally pre-emphasised) signal, f64 */
static double sample_at(const orc_params *p, const float *x, size_t n, size_t i)
{
    if (p->preemph_coef != 0.0f) {
        size_t sh = (size_t)(p->preemph_shift > 0 ? p->preemph_shift : 1) % n;
        size_t j = (i + n - sh) % n;
        return (double)x[i] - (double)p->preemph_coef * (double)x[j];
    }
    return (double)x[i];
}

/* stack_frames (processing.rs:65-129) + fft_spectrum (:143-169) + power_spectrum (:179-181) */
int orc_power_spectrum(const orc_params *p, const float *x, size_t n, double *P)
{
    size_t flen, step, T;
    int rc = orc_frame_sizes(p, &flen, &step);
    if (rc) return rc;
    rc = orc_num_frames(p, n, &T);
    if (rc) return rc;
    const size_t N = p->fft_points, F = N / 2 + 1;
    if (flen > N) return ORC_ERR_BAD_CONFIG; /* ndfft_r2c size assert */
    if (p->spectrum_exponent != 1 && p->spectrum_exponent != 2) return ORC_ERR_BAD_CONFIG;

    float *win = NULL;
    if (p->mfcc_window != ORC_WINDOW_RECT) {
        win = (float *)malloc(flen * sizeof(float));
        if (p->mfcc_window == ORC_WINDOW_HANN) orc_hann_window(flen, win);
        else orc_vorbis_window(flen, win);
    }
    double *buf = (double *)calloc(N, sizeof(double));
    double *re = (double *)malloc(F * sizeof(double)), *im = (double *)malloc(F * sizeof(double));
    double *wre = (double *)malloc(N * sizeof(double)), *wim = (double *)malloc(N * sizeof(double));
    const double inv_n = (double)(1.0f / (float)N); /* processing.rs:180: (1. / fft_points as f32) */

    for (size_t t = 0; t < T; ++t) {
        memset(buf, 0, N * sizeof(double));
        if (p->framing == ORC_FRAMING_LITERAL) {
            /* processing.rs:110-120 as written: exact_chunks((numframes,2)) over a (2,len) array
             * yields 2/numframes chunks along axis 0 -> nothing is copied when numframes > 2;
             * numframes <= 2 copies x[0..flen] into every row (flen even). */
            if (T <= 2)
                for (size_t i = 0; i < (flen & ~(size_t)1); ++i) buf[i] = sample_at(p, x, n, i);
        } else if (p->framing == ORC_FRAMING_CENTER) {
            /* frame centred on t*step; np.pad(y, flen/2, mode) semantics outside the clip */
            for (size_t i = 0; i < flen; ++i) {
                long long pos = (long long)(t * step + i) - (long long)(flen / 2);
                if (pos < 0 || pos >= (long long)n) {
                    if (p->pad_mode != ORC_PAD_REFLECT) continue; /* zeros */
                    pos = pos < 0 ? -pos : 2 * ((long long)n - 1) - pos;
                }
                buf[i] = sample_at(p, x, n, (size_t)pos);
            }
        } else {
            /* ORC_FRAMING_PADDED: samples past the signal are the appended zeros (processing.rs:93-96) */
            for (size_t i = 0; i < flen; ++i) buf[i] = t * step + i < n ? sample_at(p, x, n, t * step + i) : 0.0;
        }
        if (win) for (size_t i = 0; i < flen; ++i) buf[i] *= (double)win[i];
        rfft_f64(buf, N, re, im, wre, wim);
        for (size_t k = 0; k < F; ++k) {
            double mag = sqrt(re[k] * re[k] + im[k] * im[k]);
            P[t * F + k] = inv_n * (p->spectrum_exponent == 2 ? mag * mag : mag);
        }
    }
    free(win); free(buf); free(re); free(im); free(wre); free(wim);
    return ORC_OK;
}

/* stack_frames (processing.rs:65-129) on its own: frames [T x flen], same framing switch as orc_power_spectrum.  The
 * `filter` argument of the reference is the mfcc_window switch (frames * window, processing.rs:122-126).  Pre-emphasis is
 * not part of stack_frames and is ignored here. */
int orc_stack_frames(const orc_params *p, const float *x, size_t n, double *frames)
{
    size_t flen, step, T;
    int rc = orc_frame_sizes(p, &flen, &step);
    if (rc) return rc;
    rc = orc_num_frames(p, n, &T);
    if (rc) return rc;
    float *win = NULL;
    if (p->mfcc_window != ORC_WINDOW_RECT) {
        win = (float *)malloc(flen * sizeof(float));
        if (p->mfcc_window == ORC_WINDOW_HANN) orc_hann_window(flen, win);
        else orc_vorbis_window(flen, win);
    }
    for (size_t t = 0; t < T; ++t) {
        double *row = frames + t * flen;
        for (size_t i = 0; i < flen; ++i) row[i] = 0.0;
        if (p->framing == ORC_FRAMING_LITERAL) {
            /* processing.rs:110-120 as written (see orc_power_spectrum) */
            if (T <= 2)
                for (size_t i = 0; i < (flen & ~(size_t)1); ++i) row[i] = (double)x[i];
        } else if (p->framing == ORC_FRAMING_CENTER) {
            for (size_t i = 0; i < flen; ++i) {
                long long pos = (long long)(t * step + i) - (long long)(flen / 2);
                if (pos < 0 || pos >= (long long)n) {
                    if (p->pad_mode != ORC_PAD_REFLECT) continue;
                    pos = pos < 0 ? -pos : 2 * ((long long)n - 1) - pos;
                }
                row[i] = (double)x[pos];
            }
        } else {
            /* contract framing frames[t, i] = x[t*step + i]; ORC_FRAMING_PADDED reads the appended zeros (processing.rs:93-96) */
            for (size_t i = 0; i < flen; ++i) row[i] = t * step + i < n ? (double)x[t * step + i] : 0.0;
        }
        if (win) for (size_t i = 0; i < flen; ++i) row[i] *= (double)win[i];
    }
    free(win);
    return ORC_OK;
}

/* power_spectrum(frames: Array2<f32>, fft_points) (processing.rs:179-181) -> fft_spectrum (:143-171): rows shorter than
 * fft_points are zero-padded on the right (:147-156), R2C FFT of every row, sqrt(re^2 + im^2), times (1 / fft_points as f32). */
int orc_power_spectrum_frames(const float *frames, size_t rows, size_t cols, size_t fft_points, double *P)
{
    if (!frames || !P || cols == 0 || fft_points < 2) return ORC_ERR_ARG;
    if (cols > fft_points) return ORC_ERR_BAD_CONFIG; /* ndfft_r2c size assert */
    const size_t N = fft_points, F = N / 2 + 1;
    double *buf = (double *)calloc(N, sizeof(double));
    double *re = (double *)malloc(F * sizeof(double)), *im = (double *)malloc(F * sizeof(double));
    double *wre = (double *)malloc(N * sizeof(double)), *wim = (double *)malloc(N * sizeof(double));
    const double inv_n = (double)(1.0f / (float)N);
    for (size_t t = 0; t < rows; ++t) {
        memset(buf, 0, N * sizeof(double));
        for (size_t i = 0; i < cols; ++i) buf[i] = (double)frames[t * cols + i];
        rfft_f64(buf, N, re, im, wre, wim);
        for (size_t k = 0; k < F; ++k) P[t * F + k] = inv_n * sqrt(re[k] * re[k] + im[k] * im[k]);
    }
    free(buf); free(re); free(im); free(wre); free(wim);
    return ORC_OK;
}

/* functions.rs:66-71: exact == 0.0 -> f32::EPSILON */
static double zero_handling(double v) { return v == 0.0 ? (double)ORC_EPS_F32 : v; }

/* feature.rs:200-233 */
int orc_mfe(const orc_params *p, const float *x, size_t n, double *feat, double *energy)
{
    size_t T;
    int rc = orc_num_frames(p, n, &T);
    if (rc) return rc;
    const size_t M = p->num_filters, F = p->fft_points / 2 + 1;
    float *fb = (float *)malloc(M * F * sizeof(float));
    rc = orc_filterbank(p, fb, NULL);
    if (rc) { free(fb); return rc; }
    double *P = (double *)malloc(T * F * sizeof(double));
    rc = orc_power_spectrum(p, x, n, P);
    if (rc) { free(fb); free(P); return rc; }
    for (size_t t = 0; t < T; ++t) {
        double e = 0.0;
        for (size_t k = 0; k < F; ++k) e += P[t * F + k];     /* feature.rs:216 */
        energy[t] = zero_handling(e);                           /* :219 */
        for (size_t m = 0; m < M; ++m) {                        /* :229 P . fb^T */
            double s = 0.0;
            for (size_t k = 0; k < F; ++k) s += P[t * F + k] * (double)fb[m * F + k];
            feat[t * M + m] = zero_handling(s);                 /* :230 */
        }
    }
    free(fb); free(P);
    return ORC_OK;
}

/* feature.rs:99-148 */
int orc_mfcc(const orc_params *p, const float *x, size_t n, double *out)
{
    size_t T;
    int rc = orc_num_frames(p, n, &T);
    if (rc) return rc;
    const size_t M = p->num_filters, C = p->num_cepstral;
    if (C == 0 || C > M) return ORC_ERR_BAD_CONFIG; /* slice_move [.., ..C] panics for C > M */
    double *feat = (double *)malloc(T * M * sizeof(double));
    double *energy = (double *)malloc(T * sizeof(double));
    rc = orc_mfe(p, x, n, feat, energy);
    if (rc) { free(feat); free(energy); return rc; }

    const double g = (double)p->dct2_gain;
    /* feature.rs:126-131: n = T*M as f32; [[0,0]] *= 1/sqrt(4n); columns 1.. *= 1/sqrt(2n);
     * column 0 of rows >= 1 is left unscaled. */
    const float nn = (float)(T * M);
    const double s00 = (double)(1.0f / sqrtf(4.0f * nn));
    const double s1 = (double)(1.0f / sqrtf(2.0f * nn));
    const double o0 = 1.0 / sqrt(4.0 * (double)M), o1 = 1.0 / sqrt(2.0 * (double)M);

    for (size_t t = 0; t < T; ++t) {
        for (size_t k = 0; k < C; ++k) {
            double s = 0.0;
            for (size_t m = 0; m < M; ++m)                      /* ln: feature.rs:105, util.rs:372-381 */
                s += log(feat[t * M + m]) * cos(M_PI * (double)k * (2.0 * (double)m + 1.0) / (2.0 * (double)M));
            s *= g;                                             /* nddct2, feature.rs:123 */
            if (p->dct_norm == ORC_DCT_ORTHO) s *= (k == 0 ? o0 : o1);
            else if (k == 0) { if (t == 0) s *= s00; }
            else s *= s1;
            out[t * C + k] = s;
        }
        if (p->dc_elimination) out[t * C] = log(energy[t]);     /* feature.rs:137-146 */
    }
    free(feat); free(energy);
    return ORC_OK;
}

/* stft2 (functions.rs:86-123) + frame_analysis (:125-170), zero initial state per channel (D3).
 * Returned row r is chunk j = r + n_pad: the window covers stream samples
 * [(j+1)*H - W, (j+1)*H), zero outside [0, n). */
int orc_stft(const orc_params *p, const float *x, size_t channels, size_t n, double *out)
{
    size_t H, n_pad, R, Rreal;
    float wnorm;
    int rc = orc_stft_sizes(p, &H, &n_pad, &wnorm);
    if (rc) return rc;
    rc = orc_stft_rows(p, n, &R, &Rreal);
    if (rc) return rc;
    const size_t W = p->fft_points, F = W / 2 + 1;
    float *win = (float *)malloc(W * sizeof(float));
    orc_vorbis_window(W, win);
    double *buf = (double *)malloc(W * sizeof(double));
    double *re = (double *)malloc(F * sizeof(double)), *im = (double *)malloc(F * sizeof(double));
    double *wre = (double *)malloc(W * sizeof(double)), *wim = (double *)malloc(W * sizeof(double));
    memset(out, 0, channels * R * F * 2 * sizeof(double));
    for (size_t c = 0; c < channels; ++c) {
        const float *xc = x + c * n;
        for (size_t r = 0; r < Rreal; ++r) {
            size_t j = r + n_pad;
            long long start = (long long)((j + 1) * H) - (long long)W;
            for (size_t i = 0; i < W; ++i) {
                long long s = start + (long long)i;
                double v = (s >= 0 && (size_t)s < n) ? (double)xc[s] : 0.0;
                buf[i] = v * (double)win[i];                    /* functions.rs:137-151 */
            }
            rfft_f64(buf, W, re, im, wre, wim);                 /* :161-164 */
            double *o = out + ((c * R + r) * F) * 2;
            for (size_t k = 0; k < F; ++k) {                    /* :166-169 */
                o[2 * k] = re[k] * (double)wnorm;
                o[2 * k + 1] = im[k] * (double)wnorm;
            }
        }
    }
    free(win); free(buf); free(re); free(im); free(wre); free(wim);
    return ORC_OK;
}

/* mel_spectrogram2 (feature.rs:163-174): |X|^2 then out[n,m,t] = sum_f P[n,t,f] fb[m,f] */
int orc_mel_spectrogram(const orc_params *p, const float *x, size_t channels, size_t n, double *out)
{
    size_t R, Rreal;
    int rc = orc_stft_rows(p, n, &R, &Rreal);
    if (rc) return rc;
    const size_t M = p->num_filters, F = p->fft_points / 2 + 1;
    float *fb = (float *)malloc(M * F * sizeof(float));
    rc = orc_filterbank(p, fb, NULL);
    if (rc) { free(fb); return rc; }
    double *S = (double *)malloc(channels * R * F * 2 * sizeof(double));
    rc = orc_stft(p, x, channels, n, S);
    if (rc) { free(fb); free(S); return rc; }
    for (size_t c = 0; c < channels; ++c)
        for (size_t m = 0; m < M; ++m)
            for (size_t r = 0; r < R; ++r) {
                const double *s = S + ((c * R + r) * F) * 2;
                double acc = 0.0;
                for (size_t k = 0; k < F; ++k)
                    acc += (s[2 * k] * s[2 * k] + s[2 * k + 1] * s[2 * k + 1]) * (double)fb[m * F + k];
                out[(c * M + m) * R + r] = acc;
            }
    free(fb); free(S);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 2: reference-shaped single-thread f32 port (timed CPU baseline)                       */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float re, im; } c32;

/* f32 complex FFT plan: iterative radix-2 with a precomputed twiddle table (built per call,
 * like the reference's per-call R2cFftHandler::new, processing.rs:146). */
typedef struct { size_t n; c32 *tw; uint32_t *rev; } fft_plan32;

static int plan32_init(fft_plan32 *pl, size_t n)
{
    pl->n = n;
    pl->tw = (c32 *)malloc((n / 2 + 1) * sizeof(c32));
    pl->rev = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!pl->tw || !pl->rev) return 1;
    for (size_t k = 0; k < n / 2; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        pl->tw[k].re = (float)cos(a);
        pl->tw[k].im = (float)sin(a);
    }
    size_t bits = 0;
    while (((size_t)1 << bits) < n) ++bits;
    for (size_t i = 0; i < n; ++i) {
        size_t r = 0;
        for (size_t b = 0; b < bits; ++b) if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
        pl->rev[i] = (uint32_t)r;
    }
    return 0;
}
static void plan32_free(fft_plan32 *pl) { free(pl->tw); free(pl->rev); }

static void fft32(const fft_plan32 *pl, c32 *a)
{
    const size_t n = pl->n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = pl->rev[i];
        if (i < j) { c32 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, tstep = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; ++k) {
                c32 w = pl->tw[k * tstep];
                c32 *u = &a[i + k], *v = &a[i + k + half];
                float xr = v->re * w.re - v->im * w.im;
                float xi = v->re * w.im + v->im * w.re;
                v->re = u->re - xr; v->im = u->im - xi;
                u->re += xr;        u->im += xi;
            }
    }
}

/* R2C of one real row of length n (power of two) via a half-size complex FFT */
static void rfft32_row(const fft_plan32 *half_plan, const fft_plan32 *full_plan, const float *row,
                       c32 *out /* n/2+1 */, c32 *work /* n/2 */)
{
    const size_t n = full_plan->n, h = n / 2;
    for (size_t i = 0; i < h; ++i) { work[i].re = row[2 * i]; work[i].im = row[2 * i + 1]; }
    fft32(half_plan, work);
    for (size_t k = 0; k <= h; ++k) {
        c32 zk = work[k % h], zc = work[(h - k) % h];
        float er = 0.5f * (zk.re + zc.re), ei = 0.5f * (zk.im - zc.im);
        float dr = 0.5f * (zk.re - zc.re), di = 0.5f * (zk.im + zc.im);
        /* odd part O = (zk - conj zc)/(2i) = (di, -dr) */
        float orr = di, oi = -dr;
        c32 w = (k < h) ? full_plan->tw[k] : (c32){-1.0f, 0.0f};
        out[k].re = er + (orr * w.re - oi * w.im);
        out[k].im = ei + (orr * w.im + oi * w.re);
    }
}

/* mfe, shaped like feature.rs:200-233 + processing.rs:65-181: every intermediate is a full array */
int port_mfe_f32(const orc_params *p, const float *x, size_t n, float *feat, float *energy)
{
    size_t flen, step, T;
    int rc = orc_frame_sizes(p, &flen, &step);
    if (rc) return rc;
    rc = orc_num_frames(p, n, &T);
    if (rc) return rc;
    const size_t N = p->fft_points, F = N / 2 + 1, M = p->num_filters;
    if (flen > N || !is_pow2(N) || N < 4) return ORC_ERR_BAD_CONFIG;
    if (p->spectrum_exponent != 1 && p->spectrum_exponent != 2) return ORC_ERR_BAD_CONFIG;

    /* pass 1: stack_frames -> frames[T, flen] (contract framing; literal handled as in the oracle) */
    float *frames = (float *)calloc(T * flen, sizeof(float));
    float *pre = NULL;
    const float *src = x;
    if (p->preemph_coef != 0.0f) {
        pre = (float *)malloc(n * sizeof(float));
        size_t sh = (size_t)(p->preemph_shift > 0 ? p->preemph_shift : 1) % n;
        for (size_t i = 0; i < n; ++i) pre[i] = x[i] - p->preemph_coef * x[(i + n - sh) % n];
        src = pre;
    }
    for (size_t t = 0; t < T; ++t) {
        if (p->framing == ORC_FRAMING_LITERAL) {
            if (T <= 2) memcpy(frames + t * flen, src, (flen & ~(size_t)1) * sizeof(float));
        } else {
            const size_t have = t * step >= n ? 0 : (n - t * step < flen ? n - t * step : flen); /* PADDED: zeros past the signal */
            memcpy(frames + t * flen, src + t * step, have * sizeof(float));
        }
    }
    if (p->mfcc_window != ORC_WINDOW_RECT) {
        float *win = (float *)malloc(flen * sizeof(float));
        if (p->mfcc_window == ORC_WINDOW_HANN) orc_hann_window(flen, win);
        else orc_vorbis_window(flen, win);
        for (size_t t = 0; t < T; ++t)
            for (size_t i = 0; i < flen; ++i) frames[t * flen + i] *= win[i];
        free(win);
    }
    /* pass 2: pad to N (processing.rs:147-156) */
    float *padded = (float *)calloc(T * N, sizeof(float));
    for (size_t t = 0; t < T; ++t) memcpy(padded + t * N, frames + t * flen, flen * sizeof(float));
    /* pass 3: per-row R2C into a complex array (processing.rs:157-164); plan built per call */
    fft_plan32 full, half;
    if (plan32_init(&full, N) || plan32_init(&half, N / 2)) return ORC_ERR_ARG;
    c32 *spec = (c32 *)malloc(T * F * sizeof(c32));
    c32 *work = (c32 *)malloc((N / 2) * sizeof(c32));
    for (size_t t = 0; t < T; ++t) rfft32_row(&half, &full, padded + t * N, spec + t * F, work);
    /* pass 4: magnitude (processing.rs:168) */
    float *mag = (float *)malloc(T * F * sizeof(float));
    for (size_t i = 0; i < T * F; ++i) mag[i] = sqrtf(spec[i].re * spec[i].re + spec[i].im * spec[i].im);
    /* pass 5: scale (processing.rs:180) */
    float *P = (float *)malloc(T * F * sizeof(float));
    const float inv_n = 1.0f / (float)N;
    for (size_t i = 0; i < T * F; ++i) P[i] = inv_n * (p->spectrum_exponent == 2 ? mag[i] * mag[i] : mag[i]);
    /* pass 6: energies + zero handling (feature.rs:216-219) */
    for (size_t t = 0; t < T; ++t) {
        float e = 0.0f;
        for (size_t k = 0; k < F; ++k) e += P[t * F + k];
        energy[t] = e == 0.0f ? ORC_EPS_F32 : e;
    }
    /* pass 7: dense filterbank (clone per call) and dense product (feature.rs:222-230) */
    float *fb = (float *)malloc(M * F * sizeof(float));
    rc = orc_filterbank(p, fb, NULL);
    if (!rc) {
        for (size_t t = 0; t < T; ++t)
            for (size_t m = 0; m < M; ++m) {
                float s = 0.0f;
                const float *pr = P + t * F, *fr = fb + m * F;
                for (size_t k = 0; k < F; ++k) s += pr[k] * fr[k];
                feat[t * M + m] = s == 0.0f ? ORC_EPS_F32 : s;
            }
    }
    plan32_free(&full); plan32_free(&half);
    free(frames); free(pre); free(padded); free(spec); free(work); free(mag); free(P); free(fb);
    return rc;
}

/* feature.rs:99-148, f32 */
int port_mfcc_f32(const orc_params *p, const float *x, size_t n, float *out)
{
    size_t T;
    int rc = orc_num_frames(p, n, &T);
    if (rc) return rc;
    const size_t M = p->num_filters, C = p->num_cepstral;
    if (C == 0 || C > M) return ORC_ERR_BAD_CONFIG;
    float *feat = (float *)malloc(T * M * sizeof(float));
    float *energy = (float *)malloc(T * sizeof(float));
    rc = port_mfe_f32(p, x, n, feat, energy);
    if (rc) { free(feat); free(energy); return rc; }
    for (size_t i = 0; i < T * M; ++i) feat[i] = logf(feat[i]);          /* feature.rs:105 */
    /* full M-point DCT-II per row (new handler per call, feature.rs:120-123), then slice */
    float *ct = (float *)malloc(M * M * sizeof(float));
    for (size_t k = 0; k < M; ++k)
        for (size_t m = 0; m < M; ++m)
            ct[k * M + m] = (float)cos(M_PI * (double)k * (2.0 * (double)m + 1.0) / (2.0 * (double)M));
    float *tr = (float *)malloc(T * M * sizeof(float));
    for (size_t t = 0; t < T; ++t)
        for (size_t k = 0; k < M; ++k) {
            float s = 0.0f;
            for (size_t m = 0; m < M; ++m) s += feat[t * M + m] * ct[k * M + m];
            tr[t * M + k] = p->dct2_gain * s;
        }
    const float nn = (float)(T * M);
    if (p->dct_norm == ORC_DCT_ORTHO) {
        const float o0 = 1.0f / sqrtf(4.0f * (float)M), o1 = 1.0f / sqrtf(2.0f * (float)M);
        for (size_t t = 0; t < T; ++t) {
            tr[t * M] *= o0;
            for (size_t k = 1; k < M; ++k) tr[t * M + k] *= o1;
        }
    } else {
        tr[0] *= 1.0f / sqrtf(4.0f * nn);                                 /* feature.rs:128 */
        const float s1 = 1.0f / sqrtf(2.0f * nn);
        for (size_t t = 0; t < T; ++t)
            for (size_t k = 1; k < M; ++k) tr[t * M + k] *= s1;           /* :129-131 */
    }
    for (size_t t = 0; t < T; ++t) {
        for (size_t k = 0; k < C; ++k) out[t * C + k] = tr[t * M + k];    /* :133 */
        if (p->dc_elimination) out[t * C] = logf(energy[t]);              /* :137-146 */
    }
    free(feat); free(energy); free(ct); free(tr);
    return ORC_OK;
}

/* mel_spectrogram2 shaped like feature.rs:163-174 + functions.rs:86-170: sequential channels,
 * per-frame buffer, complex spectrum array, |.|^2 pass, dense contraction. */
int port_mel_spectrogram_f32(const orc_params *p, const float *x, size_t channels, size_t n, float *out)
{
    size_t H, n_pad, R, Rreal;
    float wnorm;
    int rc = orc_stft_sizes(p, &H, &n_pad, &wnorm);
    if (rc) return rc;
    rc = orc_stft_rows(p, n, &R, &Rreal);
    if (rc) return rc;
    const size_t W = p->fft_points, F = W / 2 + 1, M = p->num_filters;
    if (!is_pow2(W) || W < 4) return ORC_ERR_BAD_CONFIG;
    float *win = (float *)malloc(W * sizeof(float));
    orc_vorbis_window(W, win);
    fft_plan32 full, half;
    if (plan32_init(&full, W) || plan32_init(&half, W / 2)) return ORC_ERR_ARG;
    c32 *spec = (c32 *)calloc(channels * R * F, sizeof(c32));
    c32 *work = (c32 *)malloc((W / 2) * sizeof(c32));
    float *buf = (float *)malloc(W * sizeof(float));
    float *mem = (float *)malloc((W - H) * sizeof(float));
    float *chunk = (float *)malloc(H * sizeof(float));
    c32 *scratch_row = (c32 *)malloc(F * sizeof(c32));
    const size_t chunks = R; /* ceil(n/H) */
    for (size_t c = 0; c < channels; ++c) {
        memset(mem, 0, (W - H) * sizeof(float)); /* D3: zero state per channel */
        for (size_t j = 0; j < chunks; ++j) {
            size_t have = (j + 1) * H <= n ? H : (n > j * H ? n - j * H : 0);
            memcpy(chunk, x + c * n + j * H, have * sizeof(float));
            memset(chunk + have, 0, (H - have) * sizeof(float));
            for (size_t i = 0; i < W - H; ++i) buf[i] = mem[i] * win[i];
            for (size_t i = 0; i < H; ++i) buf[W - H + i] = chunk[i] * win[W - H + i];
            memmove(mem, mem + H, (W - 2 * H) * sizeof(float));
            memcpy(mem + (W - 2 * H), chunk, H * sizeof(float));
            /* the first n_pad rows are analysed too and then dropped by slice_axis_inplace
             * (functions.rs:121); keep that work so the timing has the reference's shape */
            c32 *o = j < n_pad ? scratch_row : spec + (c * R + (j - n_pad)) * F;
            rfft32_row(&half, &full, buf, o, work);
            for (size_t k = 0; k < F; ++k) { o[k].re *= wnorm; o[k].im *= wnorm; }
        }
    }
    /* |X|^2 via abs().powi(2) (feature.rs:164) */
    float *P = (float *)malloc(channels * R * F * sizeof(float));
    for (size_t i = 0; i < channels * R * F; ++i) {
        float a = sqrtf(spec[i].re * spec[i].re + spec[i].im * spec[i].im);
        P[i] = a * a;
    }
    float *fb = (float *)malloc(M * F * sizeof(float));
    rc = orc_filterbank(p, fb, NULL);
    if (!rc)
        for (size_t c = 0; c < channels; ++c)
            for (size_t m = 0; m < M; ++m)
                for (size_t r = 0; r < R; ++r) {
                    const float *pr = P + (c * R + r) * F, *fr = fb + m * F;
                    float s = 0.0f;
                    for (size_t k = 0; k < F; ++k) s += pr[k] * fr[k];
                    out[(c * M + m) * R + r] = s;
                }
    plan32_free(&full); plan32_free(&half);
    free(win); free(spec); free(work); free(buf); free(mem); free(chunk); free(scratch_row); free(P); free(fb);
    return rc;
}

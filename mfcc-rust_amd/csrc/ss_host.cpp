// Host side of the config handle: parameter validation, derived sizes and every table the HIP
// kernels consume.  Pure C++ (no HIP): this is what SpeechConfig::new (config.rs:140-185) and
// feature::filterbanks (feature.rs:36-90) do on the host in the reference.
//
// Build with -ffp-contract=off: the mel bank indices sit on integer boundaries after an f32
// ln -> exp round trip, so the f32 operation order below is part of the contract.
#include "ss_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>

namespace ss {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int status, const std::string &msg)
{
    g_last_error = msg;
    return status;
}
const std::string &last_error() { return g_last_error; }

static bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }

int derive(const ss_params &p, Derived &d)
{
    // processing.rs:77-78: (sample_rate as f32 * seconds).round() as usize
    const float fl = roundf(static_cast<float>(p.sample_rate) * p.frame_length);
    const float st = roundf(static_cast<float>(p.sample_rate) * p.frame_stride);
    if (!(fl >= 1.0f) || !(st >= 1.0f) || fl > 16777216.0f || st > 16777216.0f)
        return fail(SS_ERR_BAD_CONFIG, "frame_length/frame_stride give a zero or absurd sample count");
    d.flen = static_cast<uint32_t>(fl);
    d.step = static_cast<uint32_t>(st);
    d.n_fft = p.fft_points;
    d.n_bins = p.fft_points / 2 + 1;
    d.log2c = 0;
    d.bluestein = !is_pow2(p.fft_points);
    if (d.bluestein) {
        // chirp-z: circular convolution long enough for a[n], n < N, against the chirp at offsets -(N-1) .. N/2
        d.blu_len = 16;
        d.log2c = 4;
        while (d.blu_len < p.fft_points + p.fft_points / 2 + 1) {
            d.blu_len *= 2;
            ++d.log2c;
        }
    } else {
        while ((2u << d.log2c) < p.fft_points) ++d.log2c;
    }
    // config.rs:154: frame_size = (frame_length * sample_rate as f32) as usize  (truncation)
    const float hs = p.frame_length * static_cast<float>(p.sample_rate);
    const uint32_t hop = hs >= 1.0f ? static_cast<uint32_t>(hs) : 0u;
    // config.rs:162 (N - frame_size) and functions.rs:136 ((N - frame_size) - frame_size) are usize:
    // the STFT path only exists for N >= 2*frame_size.
    d.stft_ok = hop >= 1 && static_cast<uint64_t>(p.fft_points) >= 2ull * hop;
    d.hop = hop;
    if (d.stft_ok) {
        d.n_pad = p.fft_points / hop - 1;  // functions.rs:96
        // config.rs:178: 1.0 / (fft_points.pow(2) as f32 / (2 * frame_size) as f32)
        d.wnorm = 1.0f / (static_cast<float>(static_cast<uint64_t>(p.fft_points) * p.fft_points) /
                          static_cast<float>(2u * hop));
    }
    return SS_OK;
}

int validate(const ss_params &p)
{
    if (p.struct_size != sizeof(ss_params)) return fail(SS_ERR_ARG, "ss_params.struct_size mismatch (ABI)");
    if (p.sample_rate == 0) return fail(SS_ERR_BAD_CONFIG, "sample_rate must be > 0");
    // powers of two in [32, 8192]; any other length from 16 up whose chirp-z transform fits a 4096-point complex FFT
    // (fft_points + fft_points/2 < 4096, i.e. up to 2730: 400, 441, 800, 882, 1000, 1103, 1764, 2000, 2205 ...)
    if (is_pow2(p.fft_points) ? (p.fft_points < 32 || p.fft_points > 8192) : (p.fft_points < 16 || p.fft_points + p.fft_points / 2 + 1 > 4096))
        return fail(SS_ERR_UNSUPPORTED, "fft_points must be a power of two in [32, 8192] or any length in [16, 2730]");
    if (p.num_filters == 0 || p.num_filters > 1024) return fail(SS_ERR_BAD_CONFIG, "num_filters out of range");
    // feature.rs:133 slice_move(s![.., ..num_cepstral]) panics for num_cepstral > num_filters
    if (p.num_cepstral == 0 || p.num_cepstral > p.num_filters)
        return fail(SS_ERR_BAD_CONFIG, "num_cepstral must be in [1, num_filters]");
    const float sr = static_cast<float>(p.sample_rate);
    if (!(p.high_frequency <= sr / 2.0f))  // feature.rs:47-50
        return fail(SS_ERR_BAD_CONFIG, "High frequency cannot be greater than half of the sampling frequency!");
    if (!(p.low_frequency >= 0.0f))  // feature.rs:51
        return fail(SS_ERR_BAD_CONFIG, "low frequency cannot be less than zero!");
    if (p.framing < SS_FRAMING_CONTRACT || p.framing > SS_FRAMING_PADDED) return fail(SS_ERR_BAD_CONFIG, "framing switch");
    if (p.mel_scale < SS_MEL_REFERENCE || p.mel_scale > SS_MEL_HTK) return fail(SS_ERR_BAD_CONFIG, "mel_scale switch");
    if (p.mel_norm != SS_MEL_NORM_NONE && p.mel_norm != SS_MEL_NORM_SLANEY) return fail(SS_ERR_BAD_CONFIG, "mel_norm switch");
    if (p.mel_norm == SS_MEL_NORM_SLANEY && p.mel_scale == SS_MEL_REFERENCE)
        return fail(SS_ERR_BAD_CONFIG, "mel_norm = slaney needs mel_scale = slaney or htk");
    if (p.pad_mode != SS_PAD_REFLECT && p.pad_mode != SS_PAD_CONSTANT) return fail(SS_ERR_BAD_CONFIG, "pad_mode switch");
    if (p.spectrum_exponent != 1 && p.spectrum_exponent != 2)
        return fail(SS_ERR_BAD_CONFIG, "spectrum_exponent must be 1 or 2");
    if (p.dct_norm != SS_DCT_REFERENCE && p.dct_norm != SS_DCT_ORTHO) return fail(SS_ERR_BAD_CONFIG, "dct_norm switch");
    if (p.mfcc_window < SS_WINDOW_RECT || p.mfcc_window > SS_WINDOW_VORBIS)
        return fail(SS_ERR_BAD_CONFIG, "mfcc_window switch");
    if (p.preemph_coef != 0.0f && p.preemph_shift < 1) return fail(SS_ERR_BAD_CONFIG, "preemph_shift must be >= 1");
    Derived d;
    int rc = derive(p, d);
    if (rc) return rc;
    // ndfft_r2c asserts the lane length equals the handler size: frames longer than fft_points panic
    // (processing.rs:146-164); shorter ones are zero-padded.
    if (d.flen > p.fft_points) return fail(SS_ERR_BAD_CONFIG, "frame longer than fft_points");
    std::vector<float> fb;
    std::vector<int32_t> idx;
    return build_filterbank(p, fb, idx);
}

int num_frames(const ss_params &p, size_t n, size_t &t)
{
    Derived d;
    int rc = derive(p, d);
    if (rc) return rc;
    if (p.framing == SS_FRAMING_CENTER) {
        // librosa center=True: the clip is padded by flen/2 on both sides -> 1 + n / step frames; np.pad 'reflect' needs
        // more than flen/2 samples to mirror
        if (n == 0 || (p.pad_mode == SS_PAD_REFLECT && n <= d.flen / 2))
            return fail(SS_ERR_SHORT_SIGNAL, "signal too short for centred frames");
        t = 1 + n / d.step;
        return SS_OK;
    }
    // processing.rs:101: ((len - flen) as f32 / step as f32).floor() as usize; len < flen underflows.
    if (n < d.flen) return fail(SS_ERR_SHORT_SIGNAL, "signal shorter than one frame");
    if (p.framing == SS_FRAMING_PADDED) {
        // stack_frames(zero_padding = true), processing.rs:91-92: ceil in f32; frames past the signal read appended zeros
        t = static_cast<size_t>(ceilf(static_cast<float>(n - d.flen) / static_cast<float>(d.step)));
        if (t == 0) return fail(SS_ERR_SHORT_SIGNAL, "signal yields zero frames");
        return SS_OK;
    }
    const float q = floorf(static_cast<float>(n - d.flen) / static_cast<float>(d.step));
    t = static_cast<size_t>(q);
    // processing.rs:105: (numframes - 1) underflows for numframes == 0.
    if (t == 0) return fail(SS_ERR_SHORT_SIGNAL, "signal yields zero frames");
    return SS_OK;
}

int stft_rows(const ss_params &p, size_t n, size_t &rows, size_t &real_rows)
{
    Derived d;
    int rc = derive(p, d);
    if (rc) return rc;
    if (!d.stft_ok) return fail(SS_ERR_BAD_CONFIG, "STFT path needs fft_points >= 2 * frame_size (functions.rs:136)");
    // functions.rs:97: (ttd as f32 / frame_size as f32).ceil()
    rows = static_cast<size_t>(ceilf(static_cast<float>(n) / static_cast<float>(d.hop)));
    real_rows = rows > d.n_pad ? rows - d.n_pad : 0;
    return SS_OK;
}

void vorbis_window(size_t n, float *w)
{
    // config.rs:151-160, f64 then cast
    const double pi = 3.14159265358979323846;
    const double half = static_cast<double>(n / 2);
    for (size_t i = 0; i < n; ++i) {
        const double s = std::sin(0.5 * pi * (static_cast<double>(i) + 0.5) / half);
        w[i] = static_cast<float>(std::sin(0.5 * pi * s * s));
    }
}

void hann_window(size_t n, float *w)
{
    // functions.rs:349-357 (periodic Hann, commented out in the reference; optional here)
    const double pi = 3.14159265358979323846;
    for (size_t i = 0; i < n; ++i)
        w[i] = static_cast<float>(0.5 * (1.0 - std::cos(2.0 * pi * static_cast<double>(i) / static_cast<double>(n))));
}

// librosa.filters.mel (htk = mel_scale == SS_MEL_HTK, norm = "slaney" | None): f64 arithmetic, f32 result
static void build_filterbank_librosa(const ss_params &p, std::vector<float> &fb)
{
    const size_t M = p.num_filters, F = p.fft_points / 2 + 1;
    const double sr = p.sample_rate, fmin = p.low_frequency, fmax = p.high_frequency;
    const bool htk = p.mel_scale == SS_MEL_HTK;
    constexpr double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) {
        if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
        return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
    };
    auto mel_to_hz = [&](double m) {
        if (htk) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
        return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
    };
    std::vector<double> mel_f(M + 2);
    const double mlo = hz_to_mel(fmin), mhi = hz_to_mel(fmax);
    for (size_t i = 0; i < M + 2; ++i) mel_f[i] = mel_to_hz(mlo + (mhi - mlo) * static_cast<double>(i) / static_cast<double>(M + 1));
    fb.assign(M * F, 0.0f);
    for (size_t m = 0; m < M; ++m) {
        const double fd0 = mel_f[m + 1] - mel_f[m], fd1 = mel_f[m + 2] - mel_f[m + 1];
        const double enorm = p.mel_norm == SS_MEL_NORM_SLANEY ? 2.0 / (mel_f[m + 2] - mel_f[m]) : 1.0;
        for (size_t k = 0; k < F; ++k) {
            const double f = static_cast<double>(k) * sr / static_cast<double>(p.fft_points);  // np.fft.rfftfreq
            const double lower = (f - mel_f[m]) / fd0, upper = (mel_f[m + 2] - f) / fd1;
            const double w = std::max(0.0, std::min(lower, upper));
            fb[m * F + k] = static_cast<float>(w * enorm);
        }
    }
}

int build_filterbank(const ss_params &p, std::vector<float> &fb, std::vector<int32_t> &idx)
{
    const size_t M = p.num_filters, F = p.fft_points / 2 + 1;
    const float sr = static_cast<float>(p.sample_rate);
    if (p.mel_scale != SS_MEL_REFERENCE) {
        if (!(p.high_frequency > p.low_frequency)) return fail(SS_ERR_BAD_CONFIG, "high_frequency must exceed low_frequency");
        idx.assign(M + 2, 0);  // the integer bin edges exist in reference mode only
        build_filterbank_librosa(p, fb);
        return SS_OK;
    }
    auto mel = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };     // functions.rs:19-21
    auto hz = [](float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); };     // functions.rs:36-41
    const float lo = mel(p.low_frequency), hi = mel(p.high_frequency);
    const float dm = (hi - lo) / static_cast<float>(M + 1);                    // ndarray linspace, feature.rs:57-61
    idx.assign(M + 2, 0);
    for (size_t i = 0; i < M + 2; ++i) {
        const float f = hz(lo + dm * static_cast<float>(i));
        const float v = static_cast<float>(F + 1) * f / sr;                    // feature.rs:69-70
        idx[i] = v > 0.0f ? static_cast<int32_t>(v) : 0;
    }
    fb.assign(M * F, 0.0f);
    for (size_t i = 0; i < M; ++i) {
        const int32_t l = idx[i], m = idx[i + 1], r = idx[i + 2];
        if (r < l || static_cast<size_t>(r) + 1 > F)                           // slice_mut panic, feature.rs:84
            return fail(SS_ERR_BAD_CONFIG, "mel filter edges fall outside the spectrum");
        const float lf = static_cast<float>(l), mf = static_cast<float>(m), rf = static_cast<float>(r);
        for (int32_t x = l; x <= r; ++x) {                                     // triangle, functions.rs:43-60
            const float xf = static_cast<float>(x);
            float v = 0.0f;
            if (xf >= lf && xf < rf) {
                if (xf <= mf) v = (xf - lf) / (mf - lf);
                if (mf <= xf) v = (rf - xf) / (rf - mf);
            }
            fb[i * F + static_cast<size_t>(x)] = v;
        }
    }
    return SS_OK;
}

void sparsify(const std::vector<float> &fb, size_t M, size_t F, SparseBank &out)
{
    out = SparseBank{};
    out.start.resize(M);
    out.len.resize(M);
    out.off.resize(M);
    for (size_t m = 0; m < M; ++m) {
        size_t first = F, last = 0;
        for (size_t k = 0; k < F; ++k)
            if (fb[m * F + k] != 0.0f) {
                if (first == F) first = k;
                last = k + 1;
            }
        out.off[m] = static_cast<int32_t>(out.w.size());
        if (first == F) {  // empty filter (cfg5 has 19): contributes exactly 0 -> zero_handling -> EPS
            out.start[m] = 0;
            out.len[m] = 0;
            continue;
        }
        out.start[m] = static_cast<int32_t>(first);
        out.len[m] = static_cast<int32_t>(last - first);
        for (size_t k = first; k < last; ++k) out.w.push_back(fb[m * F + k]);
        if (out.len[m] > out.max_len) out.max_len = out.len[m];
        if (static_cast<int32_t>(last) > out.last_bin) out.last_bin = static_cast<int32_t>(last);
    }
    if (out.w.empty()) out.w.push_back(0.0f);
}

int build_tables(const ss_params &p, HostTables &t)
{
    int rc = validate(p);
    if (rc) return rc;
    t.params = p;
    rc = derive(p, t.d);
    if (rc) return rc;
    rc = build_filterbank(p, t.fb_dense, t.fb_idx);
    if (rc) return rc;
    sparsify(t.fb_dense, p.num_filters, t.d.n_bins, t.bank);

    t.window_mfcc.clear();
    if (p.mfcc_window == SS_WINDOW_HANN) {
        t.window_mfcc.resize(t.d.flen);
        hann_window(t.d.flen, t.window_mfcc.data());
    } else if (p.mfcc_window == SS_WINDOW_VORBIS) {
        t.window_mfcc.resize(t.d.flen);
        vorbis_window(t.d.flen, t.window_mfcc.data());
    }
    t.window_stft.resize(p.fft_points);
    vorbis_window(p.fft_points, t.window_stft.data());

    const double pi = 3.14159265358979323846;
    const size_t C = t.d.bluestein ? t.d.blu_len : p.fft_points / 2, N = p.fft_points;
    t.blu_c.clear();
    t.blu_b.clear();
    if (t.d.bluestein) {
        // X[k] = c[k] sum_n (x[n] c[n]) conj(c[k - n]),  c[n] = exp(-i pi n^2 / N); n^2 is reduced mod 2N in integers
        const size_t L = t.d.blu_len;
        auto chirp = [&](long long m, double *re, double *im) {
            const unsigned long long q = static_cast<unsigned long long>(m * m) % (2ull * N);
            const double ang = -pi * static_cast<double>(q) / static_cast<double>(N);
            *re = std::cos(ang);
            *im = std::sin(ang);
        };
        t.blu_c.resize(2 * N);
        for (size_t n = 0; n < N; ++n) {
            double re, im;
            chirp(static_cast<long long>(n), &re, &im);
            t.blu_c[2 * n] = static_cast<float>(re);
            t.blu_c[2 * n + 1] = static_cast<float>(im);
        }
        // b[m] = conj c[m] at offsets -(N-1) .. N/2 (negative ones wrapped to L + m), transformed once in f64
        std::vector<double> br(L, 0.0), bi(L, 0.0);
        for (long long m = -static_cast<long long>(N) + 1; m <= static_cast<long long>(N / 2); ++m) {
            double re, im;
            chirp(m, &re, &im);
            const size_t at = m >= 0 ? static_cast<size_t>(m) : L - static_cast<size_t>(-m);
            br[at] = re;
            bi[at] = -im;
        }
        // iterative radix-2 FFT (L is a power of two)
        for (size_t i = 1, jr = 0; i < L; ++i) {
            size_t bit = L >> 1;
            for (; jr & bit; bit >>= 1) jr ^= bit;
            jr ^= bit;
            if (i < jr) {
                std::swap(br[i], br[jr]);
                std::swap(bi[i], bi[jr]);
            }
        }
        for (size_t len = 2; len <= L; len <<= 1) {
            const double ang = -2.0 * pi / static_cast<double>(len);
            for (size_t i = 0; i < L; i += len)
                for (size_t k = 0; k < len / 2; ++k) {
                    const double wr = std::cos(ang * static_cast<double>(k)), wi = std::sin(ang * static_cast<double>(k));
                    const size_t u = i + k, v = i + k + len / 2;
                    const double xr = br[v] * wr - bi[v] * wi, xi = br[v] * wi + bi[v] * wr;
                    br[v] = br[u] - xr;
                    bi[v] = bi[u] - xi;
                    br[u] += xr;
                    bi[u] += xi;
                }
        }
        t.blu_b.resize(2 * L);
        for (size_t k = 0; k < L; ++k) {
            t.blu_b[2 * k] = static_cast<float>(br[k]);
            t.blu_b[2 * k + 1] = static_cast<float>(bi[k]);
        }
    }
    t.tw_c.resize(2 * C);
    for (size_t i = 0; i < C; ++i) {
        const double a = -2.0 * pi * static_cast<double>(i) / static_cast<double>(C);
        t.tw_c[2 * i] = static_cast<float>(std::cos(a));
        t.tw_c[2 * i + 1] = static_cast<float>(std::sin(a));
    }
    t.tw_n.resize(2 * (C / 2 + 1));
    for (size_t k = 0; k <= C / 2; ++k) {
        const double a = -2.0 * pi * static_cast<double>(k) / static_cast<double>(N);
        t.tw_n[2 * k] = static_cast<float>(std::cos(a));
        t.tw_n[2 * k + 1] = static_cast<float>(std::sin(a));
    }
    const size_t M = p.num_filters, Cc = p.num_cepstral;
    t.dct.resize(Cc * M);
    for (size_t k = 0; k < Cc; ++k)
        for (size_t m = 0; m < M; ++m)
            t.dct[k * M + m] = static_cast<float>(
                std::cos(pi * static_cast<double>(k) * (2.0 * static_cast<double>(m) + 1.0) / (2.0 * static_cast<double>(M))));
    return SS_OK;
}


// The stft builds of the mel-spectrogram kernels do not touch the bank: when a configuration's bank does not fit a kernel's
// mel stage (filter count, tap count, last bin), its table block is built once more without filters and marked stft_only.
static HostTables without_bank(const HostTables &t)
{
    HostTables e = t;
    e.params.num_filters = 0;
    e.bank = decltype(e.bank){};
    return e;
}


// Bank-conflict-free placement of one tap slot's filters for kernels that read taps as aligned float4s of a P row
// (ds_read_b128).  A wave64 ds_read_b128 is served in four groups of 16 lanes (MI355X guide, LDS section); 64 banks hold 16
// float4 slots, so a group is conflict-free when its lanes' first float4 slots differ mod 16.  `lanes` (32 or 64) lanes share
// one P row; filter q may start at any float4 slot in [lo4[q], hi4[q]] (earlier starts mean zero weights in front).  A
// bipartite matching (Kuhn) assigns filters to (group, residue) cells; lane_of[q] / start4_of[q] return the placement, and
// start4_idle[lane] a harmless first slot for lanes without a filter.  Filters that cannot be matched take a free cell at
// their latest start (a conflict costs time, never correctness).
static void place_taps_b128(int lanes, const std::vector<int32_t> &lo4, const std::vector<int32_t> &hi4, std::vector<int32_t> &lane_of,
                            std::vector<int32_t> &start4_of, std::vector<int32_t> &start4_idle, int32_t max4)
{
    static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int n = static_cast<int>(lo4.size()), groups = lanes / 16, cells = groups * 16;
    std::vector<int> owner(cells, -1);  // cell = group * 16 + residue -> filter
    std::vector<int> cell_of(n, -1);
    lane_of.assign(n, 0);
    start4_of.assign(n, 0);
    std::function<bool(int, std::vector<char> &)> place = [&](int q, std::vector<char> &seen) -> bool {
        for (int32_t b = hi4[q]; b >= lo4[q] && b > hi4[q] - 16; --b) {
            for (int g = 0; g < groups; ++g) {
                const int c = g * 16 + (b & 15);
                if (seen[c]) continue;
                seen[c] = 1;
                if (owner[c] < 0 || place(owner[c], seen)) {
                    owner[c] = q;
                    cell_of[q] = c;
                    start4_of[q] = b;
                    return true;
                }
            }
        }
        return false;
    };
    for (int q = 0; q < n && q < cells; ++q) {
        std::vector<char> seen(cells, 0);
        place(q, seen);
    }
    for (int q = 0; q < n && q < cells; ++q) {
        if (cell_of[q] >= 0) continue;
        for (int c = 0; c < cells; ++c)
            if (owner[c] < 0) {
                owner[c] = q;
                cell_of[q] = c;
                start4_of[q] = hi4[q];
                break;
            }
    }
    auto lane_of_cell = [&](int c) { return kGroup[(c / 16) & 1][c & 15] + 32 * (c / 32); };
    start4_idle.assign(lanes, 0);
    for (int c = 0; c < cells; ++c) {
        if (owner[c] >= 0) lane_of[owner[c]] = lane_of_cell(c);
        else start4_idle[lane_of_cell(c)] = std::min<int32_t>(c & 15, max4);  // its own residue (no conflict with the group's other lanes), inside the row
    }
}

// Re-deals the filters over the slots before they are placed (place_taps_b128 works slot by slot).  `order` deals the filters
// by span, longest first, `lanes` per slot, and the slots' spans (q4) follow from that; but short filters cluster at the low
// bins, and a slot can only take `lanes / 16` filters per float4 residue without a bank conflict (cfg3: seven of the 32
// shortest filters could not be matched).  A short filter also fits a longer slot, so one matching over the cells of ALL slots
// -- filter m may take cell (slot s, group g, residue r) if its span fits slot s and a start with that residue is admissible
// there -- finds a deal in which every slot can be placed conflict-free, where one exists.  Only for banks that fill every slot
// (the per-slot loops take `lanes` consecutive entries of `order`); `order` is left alone when no complete matching exists.
static void balance_slots(const HostTables &t, int lanes, int nslots, const int32_t *q4, int32_t kRow, std::vector<int32_t> &order)
{
    const int M = static_cast<int>(order.size());
    if (M != lanes * nslots) return;
    const int groups = lanes / 16, cells = nslots * lanes;
    std::vector<int> owner(cells, -1);  // cell = slot * lanes + group * 16 + residue -> filter
    std::vector<int> cell_of(M, -1), home(M, 0);
    for (int q = 0; q < M; ++q) home[order[q]] = q / lanes;
    auto window = [&](int m, int s, int32_t &lo, int32_t &hi) {
        const int32_t span = 4 * q4[s], len = t.bank.len[m], st = len ? t.bank.start[m] : 0;
        if (span == 0) return false;
        hi = len ? std::min(st, kRow - span) / 4 : (kRow - span) / 4;
        lo = std::max<int32_t>(0, st + len - span + 3) / 4;
        if (lo > hi && s == home[m]) lo = hi;  // (in the slot the sorted deal gave it the per-slot placement accepts it this way)
        return lo <= hi;
    };
    std::function<bool(int, std::vector<char> &)> place = [&](int m, std::vector<char> &seen) -> bool {
        for (int s = nslots - 1; s >= 0; --s) {  // the shortest slot that takes it first
            int32_t lo, hi;
            if (!window(m, s, lo, hi)) continue;
            for (int32_t b = hi; b >= lo && b > hi - 16; --b)
                for (int g = 0; g < groups; ++g) {
                    const int c = s * lanes + g * 16 + (b & 15);
                    if (seen[c]) continue;
                    seen[c] = 1;
                    if (owner[c] < 0 || place(owner[c], seen)) {
                        owner[c] = m;
                        cell_of[m] = c;
                        return true;
                    }
                }
        }
        return false;
    };
    for (int q = 0; q < M; ++q) {
        std::vector<char> seen(cells, 0);
        if (!place(order[q], seen)) return;  // no complete matching: the sorted deal stays
    }
    std::vector<int32_t> dealt;
    dealt.reserve(M);
    for (int s = 0; s < nslots; ++s)
        for (int q = 0; q < M; ++q)
            if (cell_of[order[q]] / lanes == s) dealt.push_back(order[q]);
    order = dealt;
}

void build_fast512(const HostTables &t, Fast512Tables &f)
{
    namespace L = fast512_layout;
    f = Fast512Tables{};
    const size_t M = t.params.num_filters, Cc = t.params.num_cepstral;
    if (t.d.n_fft != 512 || M > 48 || Cc > 16) return;
    // reference banks end at (F+1)/2 (feature.rs:69-70): the kernel then keeps P bins 0..128 only; a bank that reaches
    // higher (mel_scale = slaney / htk) gets the builds with the whole row of 257 bins
    if (t.bank.last_bin > 257) return;
    f.fullp = t.bank.last_bin > 129;
    const int32_t kRow = f.fullp ? 260 : 132;  // P bins a tap may touch, including three zero pad bins
    // order filters by tap count (longest first) and deal them 16 per slot
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return t.bank.len[a] > t.bank.len[b]; });
    // (slot, lane) cell -> filter, -1: unused
    std::vector<int32_t> cell(48, -1);
    for (size_t q = 0; q < M; ++q) cell[q] = order[q];
    auto spans = [&](const std::vector<int32_t> &c, int32_t (&q4)[3]) {
        int32_t maxlen[3] = {0, 0, 0};
        for (size_t q = 0; q < 48; ++q)
            if (c[q] >= 0) maxlen[q / 16] = std::max(maxlen[q / 16], t.bank.len[c[q]]);
        for (int s = 0; s < 3; ++s) q4[s] = (maxlen[s] + 3) / 4;
    };
    spans(cell, f.q4);
    if (M == 40 && !f.fullp) {
        // PAIRED layout for the symmetric DCT of the default filter count (ss_mfcc512.hip): the DCT needs s[m] = L[m] + L[39-m] and
        // d[m] = L[m] - L[39-m]; with filter 39 - m in slot 0 and filter m in slot 2 (m < 8) or slot 1 (8 <= m < 16) of the SAME
        // lane m, and the pairs (16 + i, 23 - i) in neighbouring lanes 2i, 2i + 1 of slot 1, every s and d is formed in registers -- no ln(mel) row in
        // LDS, two dependent LDS round trips fewer per quad.  Which lane of a slot holds which filter is free as far as the taps'
        // bank placement goes (that depends on the SET of first bins in the slot).  Taken when it needs no wider slot than the
        // sorted layout.
        // (cells 0 .. 39 only: the other builds of the kernel take their DCT over the first 40 cells of the (slot, lane) row)
        std::vector<int32_t> pc(48, -1);
        for (int j = 0; j < 16; ++j) pc[j] = 39 - j;       // slot 0: the wide filters 39 .. 24
        for (int j = 0; j < 8; ++j) pc[32 + j] = j;        // slot 2, lanes 0 .. 7: filters 0 .. 7
        for (int j = 8; j < 16; ++j) pc[16 + j] = j;       // slot 1, lanes 8 .. 15: filters 8 .. 15
        for (int i = 0; i < 4; ++i) {                      // slot 1, lanes 0 .. 7: the middle pairs
            pc[16 + 2 * i] = 16 + i;
            pc[16 + 2 * i + 1] = 23 - i;
        }
        int32_t pq4[3];
        spans(pc, pq4);
        if (pq4[0] <= f.q4[0] && pq4[1] <= f.q4[1] && pq4[2] <= f.q4[2]) {
            cell = pc;
            f.paired = true;
            for (int s = 0; s < 3; ++s) f.q4[s] = pq4[s];
        }
    }
    // TIGHT (the default bank: 16 / 5 / 1 taps at most in the three slots): the filters of slots 1 and 2 are placed inside the
    // slots' first 6 and 2 taps, and the paired build of the kernel reads and multiplies no more than those (the slots keep
    // their float4 pitch, and every other build of the kernel reads the same table with its 8 and 4 taps: zeros behind).
    int32_t tspan[3] = {4 * f.q4[0], 4 * f.q4[1], 4 * f.q4[2]};
    if (f.paired && f.q4[0] == 4 && f.q4[1] == 2 && f.q4[2] == 1) {
        int32_t maxlen[3] = {0, 0, 0};
        for (size_t q = 0; q < 48; ++q)
            if (cell[q] >= 0) maxlen[q / 16] = std::max(maxlen[q / 16], t.bank.len[cell[q]]);
        if (maxlen[1] <= 6 && maxlen[2] <= 2) {
            f.tight = true;
            tspan[1] = 6;
            tspan[2] = 2;
        }
    }
    f.wpitch = 4 * (f.q4[0] + f.q4[1] + f.q4[2]);
    if (f.wpitch == 0) f.wpitch = 4;
    if (f.wpitch > 160) return;
    f.tab.assign(static_cast<size_t>(L::kMelW) + 16 * f.wpitch, 0.0f);
    for (int r = 1; r < 16; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i j r / 256) = tw_c[j r]; two twiddles per 16-byte slot
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half] = t.tw_c[2 * (j * r)];
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half + 1] = t.tw_c[2 * (j * r) + 1];
        }
    for (int r = 0; r < 8; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i (j + 16 r) / 512) = tw_n[j + 16 r]
            f.tab[L::kTwn + (r * 16 + j) * 2] = t.tw_n[2 * (j + 16 * r)];
            f.tab[L::kTwn + (r * 16 + j) * 2 + 1] = t.tw_n[2 * (j + 16 * r) + 1];
        }
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 3; ++s) {
        const int32_t span = 4 * f.q4[s];
        // The 16 lanes of a slot read tap i of their filters in one ds_read_b32 group together with the 16 lanes of the
        // neighbouring frame, whose P row sits 16 banks further (144 floats): the group is conflict-free when the lanes'
        // first bins differ mod 16.  A filter shorter than the slot's span may start up to span - len bins early (zero
        // weights in front), so the first bins are chosen by a bipartite matching lanes -> residues (Kuhn's algorithm).
        int32_t lo[16], hi[16], chosen[16];  // admissible first bins [lo, hi] per lane
        for (int j = 0; j < 16; ++j) {
            const size_t q = static_cast<size_t>(s) * 16 + j;
            lo[j] = hi[j] = 0;
            if (cell[q] < 0) continue;
            const int32_t m = cell[q], st = t.bank.start[m], len = t.bank.len[m];
            hi[j] = std::min(st, kRow - span);          // the lock-step loop reads `span` taps: st + span stays inside the row
            lo[j] = std::max<int32_t>(0, st + len - tspan[s]);  // the filter's last tap stays inside the span (TIGHT: the taps the kernel reads)
            if (lo[j] > hi[j]) lo[j] = hi[j];
        }
        {
            int32_t owner[16];  // residue -> lane
            for (int r = 0; r < 16; ++r) owner[r] = -1;
            for (int j = 0; j < 16; ++j) chosen[j] = hi[j];
            if (!f.fullp) {
                std::function<bool(int, std::vector<char> &)> place = [&](int j, std::vector<char> &seen) -> bool {
                    for (int32_t b = hi[j]; b >= lo[j]; --b) {
                        const int r = b & 15;
                        if (seen[r]) continue;
                        seen[r] = 1;
                        if (owner[r] < 0 || place(owner[r], seen)) {
                            owner[r] = j;
                            chosen[j] = b;
                            return true;
                        }
                    }
                    return false;
                };
                for (int j = 0; j < 16; ++j) {
                    if (cell[static_cast<size_t>(s) * 16 + j] < 0) continue;
                    std::vector<char> seen(16, 0);
                    place(j, seen);  // unmatched lanes keep their latest admissible first bin
                }
                // an unused (slot, lane) reads the same words as a used one (same address: a broadcast, never a conflict)
                int used = -1;
                for (int j = 0; j < 16; ++j)
                    if (cell[static_cast<size_t>(s) * 16 + j] >= 0) used = j;
                for (int j = 0; j < 16; ++j)
                    if (cell[static_cast<size_t>(s) * 16 + j] < 0 && used >= 0) chosen[j] = chosen[used];
            }
        }
        for (int j = 0; j < 16; ++j) {
            const size_t q = static_cast<size_t>(s) * 16 + j;
            start[q] = chosen[j];
            filt[q] = -1;
            if (cell[q] < 0) continue;  // unused (slot, lane): zero weights -> 0 -> EPS -> ln, times a zero cosine column
            const int32_t m = cell[q];
            filt[q] = m;
            const int32_t len = t.bank.len[m];
            const int32_t st = chosen[j];
            const int32_t shift = t.bank.start[m] - st;  // zero weights in front of the filter's first tap
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
            for (size_t c = 0; c < Cc; ++c) f.tab[L::kCos + c * 52 + q] = t.dct[c * M + m];
        }
        off += span;
    }
    if (M == 40)  // half rows for the symmetric DCT of the default filter count
        for (size_t c = 0; c < Cc; ++c)
            for (size_t m = 0; m < 20; ++m) f.tab[L::kCosH + c * 20 + m] = t.dct[c * M + m];
    if (!t.window_mfcc.empty()) {  // optional frame window (mfcc_window switch), read as sample pairs by the kernel
        // all 256 pairs a 16-input build may touch (zero beyond flen): a shorter table would let the padded inputs multiply
        // whatever follows it in LDS -- 0 x NaN-patterned table words of an earlier kernel is NaN
        f.win_floats = 512;
        const size_t base = f.tab.size();
        f.tab.resize(base + static_cast<size_t>(f.win_floats), 0.0f);
        for (size_t i = 0; i < t.window_mfcc.size(); ++i) f.tab[base + i] = t.window_mfcc[i];
    }
    f.ok = true;
}

static void build_mel512_bank(const HostTables &t, Mel512Tables &f)
{
    namespace L = mel512_layout;
    f = Mel512Tables{};
    const size_t M = t.params.num_filters;
    if (t.d.n_fft != 512 || !t.d.stft_ok || M > 80 || t.window_stft.size() != 512) return;
    if (t.bank.last_bin > 257) return;
    f.fullp = t.bank.last_bin > 129;  // reference banks end at (F+1)/2
    const int32_t kRow = f.fullp ? 260 : 132;  // P bins a tap may touch, including three zero pad bins
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return t.bank.len[a] > t.bank.len[b]; });
    int32_t maxlen[5] = {0, 0, 0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 16] = std::max(maxlen[q / 16], t.bank.len[order[q]]);
    f.wpitch = 0;
    for (int s = 0; s < 5; ++s) {
        f.q4[s] = (maxlen[s] + 3) / 4;
        if (static_cast<size_t>(s) * 16 < M && f.q4[s] == 0) f.q4[s] = 1;  // the kernel stops at the first empty slot
        f.wpitch += 4 * f.q4[s];
    }
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: the 16 lanes' ds_read_b128 rows spread over the banks
    if (f.wpitch > 320) return;
    f.tab.assign(static_cast<size_t>(L::kMelW) + 16 * static_cast<size_t>(f.wpitch), 0.0f);
    for (int r = 1; r < 16; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i j r / 256) = tw_c[j r]; two twiddles per 16-byte slot
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half] = t.tw_c[2 * (j * r)];
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half + 1] = t.tw_c[2 * (j * r) + 1];
        }
    for (int r = 0; r < 8; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i (j + 16 r) / 512) = tw_n[j + 16 r]
            f.tab[L::kTwn + (r * 16 + j) * 2] = t.tw_n[2 * (j + 16 * r)];
            f.tab[L::kTwn + (r * 16 + j) * 2 + 1] = t.tw_n[2 * (j + 16 * r) + 1];
        }
    for (int i = 0; i < 512; ++i) f.tab[L::kWin + i] = t.window_stft[i];
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 5; ++s) {
        const int32_t span = 4 * f.q4[s];
        for (int j = 0; j < 16; ++j) {
            const size_t q = static_cast<size_t>(s) * 16 + j;
            start[q] = 0;
            filt[q] = -1;
            if (q >= M) continue;
            const int32_t m = order[q];
            filt[q] = m;
            int32_t st = t.bank.start[m];
            const int32_t len = t.bank.len[m];
            int32_t shift = 0;  // the lock-step loop reads `span` taps: keep st + span inside the row
            if (st + span > kRow) shift = st + span - kRow;
            st -= shift;
            start[q] = st;
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
        }
        off += span;
    }
    f.ok = true;
}

void build_mfcc512w(const HostTables &t, Mfcc512wTables &f)
{
    namespace L = mfcc512w_layout;
    f = Mfcc512wTables{};
    const size_t M = t.params.num_filters, Cc = t.params.num_cepstral;
    if (t.d.n_fft != 512 || M > 80 || Cc > 32) return;
    if (t.bank.last_bin > 257) return;
    constexpr int32_t kRow = 260;  // P bins a tap may touch: 0..256 plus three zero pad bins
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return t.bank.len[a] > t.bank.len[b]; });
    int32_t maxlen[5] = {0, 0, 0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 16] = std::max(maxlen[q / 16], t.bank.len[order[q]]);
    f.wpitch = 0;
    for (int s = 0; s < 5; ++s) {
        f.q4[s] = (maxlen[s] + 3) / 4;
        f.wpitch += 4 * f.q4[s];
    }
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: the 16 lanes' ds_read_b128 rows spread over the banks
    if (f.wpitch > 320) return;
    f.tab.assign(static_cast<size_t>(L::kMelW) + 16 * static_cast<size_t>(f.wpitch), 0.0f);
    for (int r = 1; r < 16; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i j r / 256) = tw_c[j r]; two twiddles per 16-byte slot
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half] = t.tw_c[2 * (j * r)];
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half + 1] = t.tw_c[2 * (j * r) + 1];
        }
    for (int r = 0; r < 8; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i (j + 16 r) / 512) = tw_n[j + 16 r]
            f.tab[L::kTwn + (r * 16 + j) * 2] = t.tw_n[2 * (j + 16 * r)];
            f.tab[L::kTwn + (r * 16 + j) * 2 + 1] = t.tw_n[2 * (j + 16 * r) + 1];
        }
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 5; ++s) {
        const int32_t span = 4 * f.q4[s];
        for (int j = 0; j < 16; ++j) {
            const size_t q = static_cast<size_t>(s) * 16 + j;
            start[q] = 0;
            filt[q] = -1;
            if (q >= M) continue;  // unused (slot, lane): zero weights -> 0 -> EPS -> ln, times a zero cosine column
            const int32_t m = order[q];
            filt[q] = m;
            int32_t st = t.bank.start[m];
            const int32_t len = t.bank.len[m];
            int32_t shift = 0;  // the lock-step loop reads `span` taps: keep st + span inside the row
            if (st + span > kRow) shift = st + span - kRow;
            st -= shift;
            start[q] = st;
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
            for (size_t c = 0; c < Cc; ++c) f.tab[L::kCos + c * L::kCosPitch + q] = t.dct[c * M + m];
        }
        off += span;
    }
    if (!t.window_mfcc.empty()) {  // optional frame window (mfcc_window switch), read as sample pairs
        f.windowed = true;
        const size_t base = f.tab.size();
        f.tab.resize(base + 512, 0.0f);
        for (size_t i = 0; i < t.window_mfcc.size() && i < 512; ++i) f.tab[base + i] = t.window_mfcc[i];
    }
    f.ok = true;
}

void build_mfcc256(const HostTables &t, Mfcc256Tables &f)
{
    namespace L = mfcc256_layout;
    f = Mfcc256Tables{};
    const size_t M = t.params.num_filters, Cc = t.params.num_cepstral;
    if (t.d.n_fft != 256 || M > 48 || Cc > 32) return;
    if (t.bank.last_bin > 129) return;
    constexpr int32_t kRow = 132;  // P bins a tap may touch: 0..128 plus three zero pad bins
    // order filters by tap count (longest first) and deal them 16 per slot
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return t.bank.len[a] > t.bank.len[b]; });
    int32_t maxlen[3] = {0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 16] = std::max(maxlen[q / 16], t.bank.len[order[q]]);
    for (int s = 0; s < 3; ++s) f.q4[s] = (maxlen[s] + 3) / 4;
    f.wpitch = 4 * (f.q4[0] + f.q4[1] + f.q4[2]);
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: the 16 lanes' ds_read_b128 rows spread over the banks
    if (f.wpitch > 160) return;
    f.tab.assign(static_cast<size_t>(L::kMelW) + 16 * static_cast<size_t>(f.wpitch), 0.0f);
    const double pi = 3.14159265358979323846;
    for (int r = 1; r < 16; ++r)
        for (int j = 0; j < 16; ++j) {  // exp(-2 pi i j r / 256); two twiddles per 16-byte slot
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            const double ang = -2.0 * pi * static_cast<double>(j * r) / 256.0;
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half] = static_cast<float>(std::cos(ang));
            f.tab[L::kTw2 + (p * 16 + j) * 4 + 2 * half + 1] = static_cast<float>(std::sin(ang));
        }
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 3; ++s) {
        const int32_t span = 4 * f.q4[s];
        for (int j = 0; j < 16; ++j) {
            const size_t q = static_cast<size_t>(s) * 16 + j;
            start[q] = 0;
            filt[q] = -1;
            if (q >= M) continue;  // unused (slot, lane): zero weights -> 0 -> EPS -> ln, times a zero cosine column
            const int32_t m = order[q];
            filt[q] = m;
            int32_t st = t.bank.start[m];
            const int32_t len = t.bank.len[m];
            int32_t shift = 0;  // the lock-step loop reads `span` taps: keep st + span inside the row
            if (st + span > kRow) shift = st + span - kRow;
            st -= shift;
            start[q] = st;
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
            for (size_t c = 0; c < Cc; ++c) f.tab[L::kCos + c * 52 + q] = t.dct[c * M + m];
        }
        off += span;
    }
    if (!t.window_mfcc.empty()) {  // optional frame window (mfcc_window switch)
        f.windowed = true;
        const size_t base = f.tab.size();
        f.tab.resize(base + 256, 0.0f);
        for (size_t i = 0; i < t.window_mfcc.size() && i < 256; ++i) f.tab[base + i] = t.window_mfcc[i];
    }
    f.ok = true;
}



static void build_mel2048_bank(const HostTables &t, Mel2048Tables &f)
{
    namespace L = mel2048_layout;
    f = Mel2048Tables{};
    const size_t M = t.params.num_filters;
    if (t.d.n_fft != 2048 || !t.d.stft_ok || M > 128) return;
    if (t.bank.last_bin > 1025) return;
    f.fullp = t.bank.last_bin > 513;  // reference banks end at (F+1)/2 (P bins 0..512); others get rows of all 1025 bins
    const int32_t kRow = f.fullp ? 1028 : 516;  // P bins a tap may touch, including three zero pad bins
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    // taps are read as aligned float4s of the P row: a filter's span starts at its first bin rounded down to a multiple of 4
    auto alen = [&](int32_t m) { return t.bank.len[m] ? (t.bank.start[m] & 3) + t.bank.len[m] : 0; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return alen(a) > alen(b); });
    int32_t maxlen[4] = {0, 0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 32] = std::max(maxlen[q / 32], alen(order[q]));
    for (int s = 0; s < 4; ++s) f.q4[s] = (maxlen[s] + 3) / 4;
    balance_slots(t, 32, 4, f.q4, kRow, order);  // (the slots keep their spans; see balance_slots)
    f.wpitch = 4 * (f.q4[0] + f.q4[1] + f.q4[2] + f.q4[3]);
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: the lanes' ds_read_b128 of their rows spread over all banks
    if (f.wpitch > 256) return;
    // (+4 words behind the block: [0] = how many polls the whole-line tile build waits for a hand-off before it reports a
    // protocol error, as an integer -- device-resident so that the kernel's cold path reads it from L2, not over PCIe)
    f.tab.assign(static_cast<size_t>(L::kMelW) + 32 * f.wpitch + 4, 0.0f);
    {
        const uint32_t lim = 1u << 24;
        std::memcpy(&f.tab[static_cast<size_t>(L::kMelW) + 32 * f.wpitch], &lim, sizeof lim);
    }
    const double pi = 3.14159265358979323846;
    for (int r = 1; r < 32; ++r)
        for (int j = 0; j < 32; ++j) {  // exp(-2 pi i j r / 1024) = tw_c[j r]
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            f.tab[L::kTw2 + j * L::kTw2Pitch + p * 4 + 2 * half] = t.tw_c[2 * (j * r)];
            f.tab[L::kTw2 + j * L::kTw2Pitch + p * 4 + 2 * half + 1] = t.tw_c[2 * (j * r) + 1];
        }
    for (int r = 0; r < 16; ++r)
        for (int j = 0; j < 32; ++j) {  // exp(-2 pi i (j + 32 r) / 2048) = tw_n[j + 32 r]
            f.tab[L::kTwn + j * L::kTwnPitch + 2 * r] = t.tw_n[2 * (j + 32 * r)];
            f.tab[L::kTwn + j * L::kTwnPitch + 2 * r + 1] = t.tw_n[2 * (j + 32 * r) + 1];
        }
    (void)pi;
    for (int e = 0; e < 32; ++e)
        for (int j = 0; j < 32; ++j) {  // the sample pair of lane j, register e
            f.tab[L::kWin + j * L::kWinPitch + 2 * e] = t.window_stft[2 * (j + 32 * e)];
            f.tab[L::kWin + j * L::kWinPitch + 2 * e + 1] = t.window_stft[2 * (j + 32 * e) + 1];
        }
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 4; ++s) {
        const int32_t span = 4 * f.q4[s];
        // the slot's filters go to lanes and first float4 slots that keep the lock-step ds_read_b128 tap reads conflict-free
        // (32 lanes share a P row: two read groups of 16 lanes)
        std::vector<int32_t> lo4, hi4, lane_of, start4_of, idle;
        for (int j = 0; j < 32; ++j) {
            const size_t q = static_cast<size_t>(s) * 32 + j;
            if (q >= M) break;
            const int32_t m = order[q], len = t.bank.len[m], st = len ? t.bank.start[m] : 0;
            hi4.push_back(len ? std::min(st, kRow - span) / 4 : (kRow - span) / 4);  // an empty filter may read anywhere
            lo4.push_back(std::max<int32_t>(0, st + len - span + 3) / 4);
            if (lo4.back() > hi4.back()) lo4.back() = hi4.back();
        }
        place_taps_b128(32, lo4, hi4, lane_of, start4_of, idle, (kRow - span) / 4);
        for (int j = 0; j < 32; ++j) {
            start[s * 32 + j] = 4 * idle[j];
            filt[s * 32 + j] = -1;
        }
        for (size_t k = 0; k < lo4.size(); ++k) {
            const size_t q = static_cast<size_t>(s) * 32 + k;
            const int32_t m = order[q], len = t.bank.len[m], j = lane_of[k], st = 4 * start4_of[k];
            const int32_t shift = len ? t.bank.start[m] - st : 0;  // zero weights in front of the filter's first tap
            start[s * 32 + j] = st;
            filt[s * 32 + j] = m;
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
        }
        off += span;
    }
    f.ok = true;
}


void build_mel2048(const HostTables &t, Mel2048Tables &f)
{
    build_mel2048_bank(t, f);
    if (f.ok) return;
    HostTables e = without_bank(t);
    build_mel2048_bank(e, f);
    f.stft_only = f.ok;
    f.ok = false;
}

void build_mel512(const HostTables &t, Mel512Tables &f)
{
    build_mel512_bank(t, f);
    if (f.ok) return;
    HostTables e = without_bank(t);
    build_mel512_bank(e, f);
    f.stft_only = f.ok;
    f.ok = false;
}

// mel = false: the frame-path kernel (ss_mfcc_c512); mel = true: the mel-spectrogram kernel (ss_mel_c512) -- same FFT tables
// and bank layout, the Vorbis STFT window in kWin, no cosine rows
static void build_1024(const HostTables &t, Mfcc1024Tables &f, bool mel)
{
    namespace L = mfcc1024_layout;
    f = Mfcc1024Tables{};
    const size_t M = t.params.num_filters, Cc = mel ? 0 : t.params.num_cepstral;
    if (t.d.n_fft != 1024 || M > 128 || Cc > 64) return;
    if (mel && (!t.d.stft_ok || t.window_stft.size() != 1024)) return;
    if (t.bank.last_bin > 513) return;
    f.fullp = t.bank.last_bin > 257;  // reference banks end at (F+1)/2 (P bins 0..256); librosa-style ones need all 513
    const int32_t kRow = f.fullp ? 516 : 260;
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    auto alen = [&](int32_t m) { return t.bank.len[m] ? (t.bank.start[m] & 3) + t.bank.len[m] : 0; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return alen(a) > alen(b); });
    int32_t maxlen[4] = {0, 0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 32] = std::max(maxlen[q / 32], alen(order[q]));
    for (int s = 0; s < 4; ++s) f.q4[s] = (maxlen[s] + 3) / 4;
    balance_slots(t, 32, 4, f.q4, kRow, order);  // (the slots keep their spans; see balance_slots)
    f.wpitch = 4 * (f.q4[0] + f.q4[1] + f.q4[2] + f.q4[3]);
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: conflict-free ds_read_b128 of the lanes' rows
    if (f.wpitch > 256) return;
    f.tab.assign(static_cast<size_t>(L::kMelW) + 32 * static_cast<size_t>(f.wpitch), 0.0f);
    const double pi = 3.14159265358979323846;
    auto cis = [&](double num, double den, float *dst) {
        const double ang = -2.0 * pi * num / den;
        dst[0] = static_cast<float>(std::cos(ang));
        dst[1] = static_cast<float>(std::sin(ang));
    };
    for (int r = 1; r < 16; ++r)
        for (int k1 = 0; k1 < 16; ++k1) {
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            cis(static_cast<double>(k1 * r), 256.0, &f.tab[L::kT1 + (p * 16 + k1) * 4 + 2 * half]);
        }
    for (int i = 0; i < 8; ++i)
        for (int jj = 0; jj < 32; ++jj) {
            const int k1 = jj & 15, hh = jj >> 4;
            cis(static_cast<double>(k1 + 16 * (i + 8 * hh)), 512.0, &f.tab[L::kT2 + (i * 32 + jj) * 2]);
            cis(static_cast<double>(k1 + 16 * i + 128 * hh), 1024.0, &f.tab[L::kTwn + (i * 32 + jj) * 2]);
        }
    if (mel) {
        f.windowed = true;
        for (size_t i = 0; i < 1024; ++i) f.tab[L::kWin + i] = t.window_stft[i];
    } else if (!t.window_mfcc.empty()) {
        f.windowed = true;
        for (size_t i = 0; i < t.window_mfcc.size() && i < 1024; ++i) f.tab[L::kWin + i] = t.window_mfcc[i];
    }
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 4; ++s) {
        const int32_t span = 4 * f.q4[s];
        // the slot's filters go to lanes and first float4 slots that keep the lock-step ds_read_b128 tap reads conflict-free
        // (32 lanes share a P row: two read groups of 16 lanes; see place_taps_b128)
        std::vector<int32_t> lo4, hi4, lane_of, start4_of, idle;
        for (int j = 0; j < 32; ++j) {
            const size_t q = static_cast<size_t>(s) * 32 + j;
            if (q >= M) break;
            const int32_t m = order[q], len = t.bank.len[m], st = len ? t.bank.start[m] : 0;
            hi4.push_back(len ? std::min(st, kRow - span) / 4 : (kRow - span) / 4);  // an empty filter may read anywhere
            lo4.push_back(std::max<int32_t>(0, st + len - span + 3) / 4);
            if (lo4.back() > hi4.back()) lo4.back() = hi4.back();
        }
        place_taps_b128(32, lo4, hi4, lane_of, start4_of, idle, (kRow - span) / 4);
        for (int j = 0; j < 32; ++j) {
            start[s * 32 + j] = 4 * idle[j];
            filt[s * 32 + j] = -1;
        }
        for (size_t k = 0; k < lo4.size(); ++k) {
            const size_t q = static_cast<size_t>(s) * 32 + k;
            const int32_t m = order[q], len = t.bank.len[m], j = lane_of[k], st = 4 * start4_of[k];
            const int32_t shift = len ? t.bank.start[m] - st : 0;  // zero weights in front of the filter's first tap
            start[s * 32 + j] = st;
            filt[s * 32 + j] = m;
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
        }
        off += span;
    }
    for (size_t c = 0; c < Cc; ++c)
        for (size_t m = 0; m < (M + 1) / 2; ++m) f.tab[L::kCos + c * L::kCosPitch + m] = t.dct[c * M + m];
    f.ok = true;
}

void build_mfcc1024(const HostTables &t, Mfcc1024Tables &f) { build_1024(t, f, false); }
void build_mel1024(const HostTables &t, Mfcc1024Tables &f)
{
    build_1024(t, f, true);
    if (f.ok) return;
    HostTables e = without_bank(t);
    build_1024(e, f, true);
    f.stft_only = f.ok;
    f.ok = false;
}

void build_mfcc2048(const HostTables &t, Mfcc2048Tables &f)
{
    namespace L = mfcc2048_layout;
    f = Mfcc2048Tables{};
    const size_t M = t.params.num_filters, Cc = t.params.num_cepstral;
    if (t.d.n_fft != 2048 || M > 128 || Cc > 64) return;
    if (t.bank.last_bin > 1025) return;
    f.fullp = t.bank.last_bin > 513;  // reference banks end at (F+1)/2 (P bins 0..512); librosa-style ones need all 1025
    const int32_t kRow = f.fullp ? 1028 : 516;
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    auto alen = [&](int32_t m) { return t.bank.len[m] ? (t.bank.start[m] & 3) + t.bank.len[m] : 0; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return alen(a) > alen(b); });
    int32_t maxlen[4] = {0, 0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 32] = std::max(maxlen[q / 32], alen(order[q]));
    for (int s = 0; s < 4; ++s) f.q4[s] = (maxlen[s] + 3) / 4;
    balance_slots(t, 32, 4, f.q4, kRow, order);  // (the slots keep their spans; see balance_slots)
    f.wpitch = 4 * (f.q4[0] + f.q4[1] + f.q4[2] + f.q4[3]);
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: conflict-free ds_read_b128 of the lanes' rows
    if (f.wpitch > 256) return;
    f.tab.assign(static_cast<size_t>(L::kMelW) + 32 * static_cast<size_t>(f.wpitch), 0.0f);
    const double pi = 3.14159265358979323846;
    auto cis = [&](double num, double den, float *dst) {
        const double ang = -2.0 * pi * num / den;
        dst[0] = static_cast<float>(std::cos(ang));
        dst[1] = static_cast<float>(std::sin(ang));
    };
    for (int r = 1; r < 32; ++r)
        for (int j = 0; j < 32; ++j) {
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            cis(static_cast<double>(j * r), 1024.0, &f.tab[L::kTw2 + (p * 32 + j) * 4 + 2 * half]);
        }
    for (int r = 0; r < 16; ++r)
        for (int j = 0; j < 32; ++j) cis(static_cast<double>(j + 32 * r), 2048.0, &f.tab[L::kTwn + (r * 32 + j) * 2]);
    if (!t.window_mfcc.empty()) {
        f.windowed = true;
        for (size_t i = 0; i < t.window_mfcc.size() && i < 2048; ++i) f.tab[L::kWin + i] = t.window_mfcc[i];
    }
    int32_t *start = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
    int32_t *filt = reinterpret_cast<int32_t *>(f.tab.data() + L::kFilt);
    int32_t off = 0;
    for (int s = 0; s < 4; ++s) {
        const int32_t span = 4 * f.q4[s];
        // the slot's filters go to lanes and first float4 slots that keep the lock-step ds_read_b128 tap reads conflict-free
        // (32 lanes share a P row: two read groups of 16 lanes; see place_taps_b128)
        std::vector<int32_t> lo4, hi4, lane_of, start4_of, idle;
        for (int j = 0; j < 32; ++j) {
            const size_t q = static_cast<size_t>(s) * 32 + j;
            if (q >= M) break;
            const int32_t m = order[q], len = t.bank.len[m], st = len ? t.bank.start[m] : 0;
            hi4.push_back(len ? std::min(st, kRow - span) / 4 : (kRow - span) / 4);  // an empty filter may read anywhere
            lo4.push_back(std::max<int32_t>(0, st + len - span + 3) / 4);
            if (lo4.back() > hi4.back()) lo4.back() = hi4.back();
        }
        place_taps_b128(32, lo4, hi4, lane_of, start4_of, idle, (kRow - span) / 4);
        for (int j = 0; j < 32; ++j) {
            start[s * 32 + j] = 4 * idle[j];
            filt[s * 32 + j] = -1;
        }
        for (size_t k = 0; k < lo4.size(); ++k) {
            const size_t q = static_cast<size_t>(s) * 32 + k;
            const int32_t m = order[q], len = t.bank.len[m], j = lane_of[k], st = 4 * start4_of[k];
            const int32_t shift = len ? t.bank.start[m] - st : 0;  // zero weights in front of the filter's first tap
            start[s * 32 + j] = st;
            filt[s * 32 + j] = m;
            for (int32_t i = 0; i < len; ++i)
                f.tab[L::kMelW + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
        }
        off += span;
    }
    for (size_t c = 0; c < Cc; ++c)
        for (size_t m = 0; m < (M + 1) / 2; ++m) f.tab[L::kCos + c * L::kCosPitch + m] = t.dct[c * M + m];
    f.ok = true;
}

// mel = false: the frame-path kernel (ss_mfcc_c2048); mel = true: the mel-spectrogram kernel (ss_mel_c2048) -- same FFT
// tables and bank layout, no cosine rows, the Vorbis STFT window behind the mel rows
static void build_4096(const HostTables &t, Mfcc4096Tables &f, bool mel)
{
    namespace L = mfcc4096_layout;
    f = Mfcc4096Tables{};
    const size_t M = t.params.num_filters, Cc = mel ? 0 : t.params.num_cepstral;
    if (t.d.n_fft != 4096 || M > 256 || Cc > 64) return;
    if (mel && (!t.d.stft_ok || t.window_stft.size() != 4096)) return;
    if (t.bank.last_bin > 1025) return;  // the kernel keeps P bins 0..1024
    constexpr int32_t kRow = 1028;       // P bins a tap may touch: 0..1024 plus three zero pad bins
    std::vector<int32_t> order(M);
    for (size_t m = 0; m < M; ++m) order[m] = static_cast<int32_t>(m);
    // taps are read as aligned float4s of the P row: a filter's span starts at its first bin rounded down to a multiple of 4
    auto alen = [&](int32_t m) { return t.bank.len[m] ? (t.bank.start[m] & 3) + t.bank.len[m] : 0; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return alen(a) > alen(b); });
    int32_t maxlen[4] = {0, 0, 0, 0};
    for (size_t q = 0; q < M; ++q) maxlen[q / 64] = std::max(maxlen[q / 64], alen(order[q]));
    for (int s = 0; s < 4; ++s) f.q4[s] = (maxlen[s] + 3) / 4;
    balance_slots(t, 64, 4, f.q4, kRow, order);  // (the slots keep their spans; see balance_slots)
    f.wpitch = 4 * (f.q4[0] + f.q4[1] + f.q4[2] + f.q4[3]);
    if (f.wpitch == 0) f.wpitch = 4;
    if ((f.wpitch / 4) % 2 == 0) f.wpitch += 4;  // odd pitch in 16-byte units: the lanes' ds_read_b128 of their rows spread over all banks
    if (f.wpitch > 128) return;
    // DCT stage: with n_filters % 4 == 0 the 256-term product folds twice (an even coefficient is 64 terms, an odd one two
    // halves of 64) and 64 lanes cover up to 43 coefficients in one pass; the cosine block is then one 64-term row per lane
    f.dct_fold2 = !mel && M % 4 == 0 && Cc >= 1 && Cc <= 43;
    // (fold2: the last lane's row needs no pad behind it -- those 16 bytes are what lets a 12-wave workgroup fit 160 KB of LDS)
    f.cos_floats = static_cast<int32_t>(f.dct_fold2 ? 63 * L::kCosLanePitch + 64 : Cc * L::kCosPitch);
    const size_t melw0 = static_cast<size_t>(L::kCos) + f.cos_floats;
    f.tab.assign(melw0 + 64 * static_cast<size_t>(f.wpitch), 0.0f);
    const double pi = 3.14159265358979323846;
    auto cis = [&](double num, double den, float *dst) {
        const double ang = -2.0 * pi * num / den;
        dst[0] = static_cast<float>(std::cos(ang));
        dst[1] = static_cast<float>(std::sin(ang));
    };
    for (int r = 1; r < 32; ++r)
        for (int k1 = 0; k1 < 32; ++k1) {
            const int p = (r - 1) / 2, half = (r - 1) % 2;
            cis(static_cast<double>(k1 * r), 1024.0, &f.tab[L::kT1 + (p * 32 + k1) * 4 + 2 * half]);
        }
    for (int i = 0; i < 16; ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const int k1 = lane & 31, hh = lane >> 5;
            cis(static_cast<double>(k1 + 32 * (i + 16 * hh)), 2048.0, &f.tab[L::kT2 + (i * 64 + lane) * 2]);
            cis(static_cast<double>(k1 + 32 * i + 512 * hh), 4096.0, &f.tab[L::kTwn + (i * 64 + lane) * 2]);
        }
    // one word per (slot, lane): first P bin | filter index << 16 (one register in the kernel; one table less in its LDS)
    std::vector<int32_t> start(256, 0), filt(256, -1);
    int32_t off = 0;
    for (int s = 0; s < 4; ++s) {
        const int32_t span = 4 * f.q4[s];
        // the slot's filters go to lanes and first float4 slots that keep the lock-step ds_read_b128 tap reads conflict-free
        std::vector<int32_t> lo4, hi4, lane_of, start4_of, idle;
        for (int j = 0; j < 64; ++j) {
            const size_t q = static_cast<size_t>(s) * 64 + j;
            if (q >= M) break;
            const int32_t m = order[q], len = t.bank.len[m], st = len ? t.bank.start[m] : 0;
            hi4.push_back(len ? std::min(st, kRow - span) / 4 : (kRow - span) / 4);  // an empty filter may read anywhere
            lo4.push_back(std::max<int32_t>(0, st + len - span + 3) / 4);
            if (lo4.back() > hi4.back()) lo4.back() = hi4.back();
        }
        place_taps_b128(64, lo4, hi4, lane_of, start4_of, idle, (kRow - span) / 4);
        for (int j = 0; j < 64; ++j) {
            start[s * 64 + j] = 4 * idle[j];
            filt[s * 64 + j] = -1;
        }
        for (size_t k = 0; k < lo4.size(); ++k) {
            const size_t q = static_cast<size_t>(s) * 64 + k;
            const int32_t m = order[q], len = t.bank.len[m], j = lane_of[k], st = 4 * start4_of[k];
            const int32_t shift = len ? t.bank.start[m] - st : 0;  // zero weights in front of the filter's first tap
            start[s * 64 + j] = st;
            filt[s * 64 + j] = m;
            for (int32_t i = 0; i < len; ++i)
                f.tab[melw0 + static_cast<size_t>(j) * f.wpitch + off + shift + i] = t.bank.w[t.bank.off[m] + i];
        }
        off += span;
    }
    {
        int32_t *packed = reinterpret_cast<int32_t *>(f.tab.data() + L::kStart);
        for (int q = 0; q < 256; ++q) packed[q] = (start[q] & 0xffff) | static_cast<int32_t>(static_cast<uint32_t>(filt[q]) << 16);
    }
    if (f.dct_fold2) {
        // lane assignment of ss_mfcc_c2048's product stage: lanes 0 .. ne-1 the even coefficients 2 lane (filters 0 .. M/4-1),
        // then from the next even lane on pairs of lanes per odd coefficient (filters 0..63 and 64..127 of its M/2)
        const size_t ne = (Cc + 1) / 2, no = Cc / 2, nep = (ne + 1) & ~size_t(1);
        for (size_t lane = 0; lane < 64; ++lane) {
            float *row = &f.tab[L::kCos + lane * L::kCosLanePitch];
            if (lane < ne) {
                for (size_t m = 0; m < M / 4; ++m) row[m] = t.dct[2 * lane * M + m];
            } else if (lane >= nep && (lane - nep) / 2 < no) {
                const size_t cc = 2 * ((lane - nep) / 2) + 1, m0 = 64 * ((lane - nep) & 1);
                for (size_t i = 0; i < 64 && m0 + i < M / 2; ++i) row[i] = t.dct[cc * M + m0 + i];
            }
        }
    } else {
        for (size_t cc = 0; cc < Cc; ++cc)
            for (size_t m = 0; m < (M + 1) / 2; ++m) f.tab[L::kCos + cc * L::kCosPitch + m] = t.dct[cc * M + m];
    }
    if (mel) {
        const size_t base = f.tab.size();
        f.tab.resize(base + 4096);
        for (size_t i = 0; i < 4096; ++i) f.tab[base + i] = t.window_stft[i];
    }
    f.ok = true;
}

void build_mfcc4096(const HostTables &t, Mfcc4096Tables &f) { build_4096(t, f, false); }
void build_mel4096(const HostTables &t, Mfcc4096Tables &f)
{
    build_4096(t, f, true);
    if (f.ok) return;
    HostTables e = without_bank(t);
    build_4096(e, f, true);
    f.stft_only = f.ok;
    f.ok = false;
}

}  // namespace ss

// ---- host-only C ABI entry points ------------------------------------------------------------

namespace ss { const std::string &last_error(); }

extern "C" {

int ss_params_default(ss_params *p, uint32_t sample_rate)
{
    if (!p) return ss::fail(SS_ERR_ARG, "null params");
    std::memset(p, 0, sizeof(*p));
    p->struct_size = static_cast<uint32_t>(sizeof(ss_params));
    p->sample_rate = sample_rate;  // config.rs:35-47
    p->fft_points = 512;
    p->frame_length = 0.02f;
    p->frame_stride = 0.01f;
    p->num_cepstral = 13;
    p->num_filters = 40;
    p->low_frequency = 0.0f;
    p->high_frequency = static_cast<float>(sample_rate) / 2.0f;
    p->dc_elimination = 1;
    p->framing = SS_FRAMING_CONTRACT;
    p->spectrum_exponent = 1;
    p->dct_norm = SS_DCT_REFERENCE;
    p->dct2_gain = SS_DCT2_GAIN;
    p->mfcc_window = SS_WINDOW_RECT;
    p->preemph_coef = 0.0f;
    p->preemph_shift = 1;
    p->mel_scale = SS_MEL_REFERENCE;
    p->mel_norm = SS_MEL_NORM_NONE;
    p->pad_mode = SS_PAD_REFLECT;
    return SS_OK;
}

int ss_params_validate(const ss_params *p)
{
    if (!p) return ss::fail(SS_ERR_ARG, "null params");
    return ss::validate(*p);
}

int ss_frame_sizes(const ss_params *p, size_t *frame_len, size_t *frame_step)
{
    if (!p || !frame_len || !frame_step) return ss::fail(SS_ERR_ARG, "null argument");
    ss::Derived d;
    int rc = ss::derive(*p, d);
    if (rc) return rc;
    *frame_len = d.flen;
    *frame_step = d.step;
    return SS_OK;
}

int ss_num_frames(const ss_params *p, size_t n_samples, size_t *n_frames)
{
    if (!p || !n_frames) return ss::fail(SS_ERR_ARG, "null argument");
    size_t t = 0;
    int rc = ss::num_frames(*p, n_samples, t);
    if (rc) return rc;
    *n_frames = t;
    return SS_OK;
}

int ss_stft_sizes(const ss_params *p, size_t *hop, size_t *n_pad, float *wnorm)
{
    if (!p || !hop || !n_pad || !wnorm) return ss::fail(SS_ERR_ARG, "null argument");
    ss::Derived d;
    int rc = ss::derive(*p, d);
    if (rc) return rc;
    if (!d.stft_ok) return ss::fail(SS_ERR_BAD_CONFIG, "STFT path needs fft_points >= 2 * frame_size (functions.rs:136)");
    *hop = d.hop;
    *n_pad = d.n_pad;
    *wnorm = d.wnorm;
    return SS_OK;
}

int ss_stft_rows(const ss_params *p, size_t n_samples, size_t *rows, size_t *real_rows)
{
    if (!p || !rows || !real_rows) return ss::fail(SS_ERR_ARG, "null argument");
    return ss::stft_rows(*p, n_samples, *rows, *real_rows);
}

int ss_filterbank(const ss_params *p, float *fb, int32_t *idx)
{
    if (!p || !fb) return ss::fail(SS_ERR_ARG, "null argument");
    int rc = ss::validate(*p);
    if (rc) return rc;
    std::vector<float> dense;
    std::vector<int32_t> id;
    rc = ss::build_filterbank(*p, dense, id);
    if (rc) return rc;
    std::memcpy(fb, dense.data(), dense.size() * sizeof(float));
    if (idx) std::memcpy(idx, id.data(), id.size() * sizeof(int32_t));
    return SS_OK;
}

int ss_vorbis_window(size_t n, float *w)
{
    if (!w || n < 2) return ss::fail(SS_ERR_ARG, "bad window request");
    ss::vorbis_window(n, w);
    return SS_OK;
}

const char *ss_status_string(int status)
{
    switch (status) {
        case SS_OK: return "ok";
        case SS_ERR_SHORT_SIGNAL: return "signal too short for one frame";
        case SS_ERR_BAD_CONFIG: return "invalid configuration";
        case SS_ERR_ARG: return "invalid argument";
        case SS_ERR_HIP: return "HIP runtime error / no device";
        case SS_ERR_UNSUPPORTED: return "unsupported configuration";
        case SS_ERR_DEVICE: return "device-side protocol error reported by a kernel";
        default: return "unknown status";
    }
}

const char *ss_last_error_string(void) { return ss::last_error().c_str(); }
int ss_abi_version(void) { return SS_ABI_VERSION; }

}  // extern "C"

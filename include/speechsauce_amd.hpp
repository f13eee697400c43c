// speechsauce_amd.hpp -- header-only C++ mirror of the reference crate's Rust API over the C ABI
// (speechsauce_amd.h).  Same names, argument meaning and error behaviour as the reference:
//   speechsauce::config::{SpeechConfigBuilder, SpeechConfig}   (speechsauce/src/config.rs:10-190)
//   speechsauce::feature::{mfcc, mfe, mel_spectrogram1, mel_spectrogram2}  (feature.rs:99-233)
//   speechsauce::processing::preemphasis                        (processing.rs:31-53)
//   speechsauce::processing::{stack_frames, power_spectrum}     (processing.rs:65-129, :179-181)
//   speechsauce::functions::{stft1, stft2}                      (functions.rs:199-233, :86-123)
//   speechsauce::processing::{cmvn, cmvnw, derivative_extraction}, feature::extract_derivative_feature
//                                                               (processing.rs:222-371, feature.rs:253-269)
// Where the reference panics, these throw speechsauce::Error carrying the ss_status.
// Arrays are plain row-major std::vector<float> plus shapes (the reference returns ndarray::ArrayN<f32>).
#pragma once

#include <algorithm>
#include <complex>
#include <cstddef>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "speechsauce_amd.h"

namespace speechsauce {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(int status)
{
    if (status != SS_OK) {
        const char *d = ss_last_error_string();
        throw Error(status, (d && *d) ? d : ss_status_string(status));
    }
}

// 2-D / 3-D owning arrays, row-major
struct Array2 {
    std::size_t rows = 0, cols = 0;
    std::vector<float> data;
    float &operator()(std::size_t r, std::size_t c) { return data[r * cols + c]; }
    float operator()(std::size_t r, std::size_t c) const { return data[r * cols + c]; }
};
struct Array3 {
    std::size_t d0 = 0, d1 = 0, d2 = 0;
    std::vector<float> data;
};

class SpeechConfig;

// config.rs:10-97
class SpeechConfigBuilder {
public:
    explicit SpeechConfigBuilder(std::size_t sample_rate) { check(ss_params_default(&p_, static_cast<uint32_t>(sample_rate))); }
    SpeechConfigBuilder &high_freq(float v) { p_.high_frequency = v; return *this; }
    // librosa-compatible variants (ss_params switches): SS_FRAMING_CENTER + SS_PAD_*, SS_MEL_SLANEY / SS_MEL_HTK, SS_MEL_NORM_SLANEY
    SpeechConfigBuilder &framing(int v, int pad_mode = SS_PAD_REFLECT) { p_.framing = v; p_.pad_mode = pad_mode; return *this; }
    SpeechConfigBuilder &mel_scale(int scale, int norm = SS_MEL_NORM_NONE) { p_.mel_scale = scale; p_.mel_norm = norm; return *this; }
    SpeechConfigBuilder &low_freq(float v) { p_.low_frequency = v; return *this; }
    SpeechConfigBuilder &dc_elimination(bool v) { p_.dc_elimination = v; return *this; }
    SpeechConfigBuilder &num_cepstral(std::size_t v) { p_.num_cepstral = static_cast<uint32_t>(v); return *this; }
    SpeechConfigBuilder &frame_stride(float v) { p_.frame_stride = v; return *this; }
    SpeechConfigBuilder &frame_length(float v) { p_.frame_length = v; return *this; }
    SpeechConfigBuilder &fft_points(std::size_t v) { p_.fft_points = static_cast<uint32_t>(v); return *this; }
    // not settable in the reference builder (config.rs:49-82 has no num_filters setter); offered here
    SpeechConfigBuilder &num_filters(std::size_t v) { p_.num_filters = static_cast<uint32_t>(v); return *this; }
    // switches of include/speechsauce_amd.h (reference mode when untouched)
    ss_params &params() { return p_; }
    inline SpeechConfig build() const;

private:
    ss_params p_{};
};

// config.rs:98-190.  Immutable after creation: unlike the reference there is no STFT carry-over state.  Copyable like the
// reference's `#[derive(Clone)]` (config.rs:98): copies share one device handle, released with the last of them.
class SpeechConfig {
public:
    // SpeechConfig::new, config.rs:140-150
    SpeechConfig(std::size_t sample_rate, std::size_t fft_points, float frame_length, float frame_stride, std::size_t num_cepstral,
                 std::size_t num_filters, float low_frequency, float high_frequency, bool dc_elimination)
    {
        check(ss_params_default(&p_, static_cast<uint32_t>(sample_rate)));
        p_.fft_points = static_cast<uint32_t>(fft_points);
        p_.frame_length = frame_length;
        p_.frame_stride = frame_stride;
        p_.num_cepstral = static_cast<uint32_t>(num_cepstral);
        p_.num_filters = static_cast<uint32_t>(num_filters);
        p_.low_frequency = low_frequency;
        p_.high_frequency = high_frequency;
        p_.dc_elimination = dc_elimination;
        create_();
    }
    explicit SpeechConfig(const ss_params &p) : p_(p) { create_(); }
    SpeechConfig() : SpeechConfig(SpeechConfigBuilder(16000).build()) {}  // Default, config.rs:133-137

    const ss_params &params() const { return p_; }
    const ss_config *handle() const { return h_.get(); }
    // the reference's public plain-data fields (config.rs:100-126) as accessors
    std::size_t sample_rate() const { return p_.sample_rate; }
    std::size_t window_size() const { return p_.fft_points; }
    std::size_t window_size_half() const { return p_.fft_points / 2; }
    std::size_t freq_size() const { return p_.fft_points / 2 + 1; }
    std::size_t frame_size() const { return frame_size_; }    // trunc(frame_length * sample_rate), config.rs:154
    float frame_length() const { return p_.frame_length; }
    float frame_stride() const { return p_.frame_stride; }
    std::size_t num_cepstral() const { return p_.num_cepstral; }
    std::size_t num_filters() const { return p_.num_filters; }
    float low_frequency() const { return p_.low_frequency; }
    float high_frequency() const { return p_.high_frequency; }
    bool dc_elimination() const { return p_.dc_elimination != 0; }
    float wnorm() const { return wnorm_; }                    // 2 frame_size / fft_points^2, config.rs:178
    const std::vector<float> &window() const { return window_; }  // Vorbis window of window_size points, config.rs:151-160

private:
    void create_()
    {
        ss_config *h = nullptr;
        check(ss_config_create(&p_, &h));
        h_ = std::shared_ptr<ss_config>(h, [](ss_config *c) { ss_config_destroy(c); });
        // config.rs:154 / :178, the reference's own f32 arithmetic (every config has these two, not only the STFT-capable ones
        // ss_stft_sizes answers for): frame_size = trunc(frame_length * sample_rate), wnorm = 1 / (fft_points^2 / (2 frame_size))
        frame_size_ = static_cast<std::size_t>(p_.frame_length * static_cast<float>(p_.sample_rate));
        const std::size_t n2 = static_cast<std::size_t>(p_.fft_points) * p_.fft_points;
        wnorm_ = 1.0f / (static_cast<float>(n2) / static_cast<float>(2 * frame_size_));
        window_.resize(p_.fft_points);
        check(ss_vorbis_window(window_.size(), window_.data()));
    }
    ss_params p_{};
    std::shared_ptr<ss_config> h_;
    std::size_t frame_size_ = 0;
    float wnorm_ = 0.f;
    std::vector<float> window_;
};

inline SpeechConfig SpeechConfigBuilder::build() const { return SpeechConfig(p_); }

// feature.rs:99-148
inline Array2 mfcc(const float *signal, std::size_t n, const SpeechConfig &cfg)
{
    std::size_t t = 0;
    check(ss_num_frames(&cfg.params(), n, &t));
    Array2 out{t, cfg.num_cepstral(), std::vector<float>(t * cfg.num_cepstral())};
    check(ss_mfcc(cfg.handle(), signal, n, out.data.data()));
    return out;
}
inline Array2 mfcc(const std::vector<float> &signal, const SpeechConfig &cfg) { return mfcc(signal.data(), signal.size(), cfg); }

// feature.rs:200-233: (features [T x M], frame energies [T])
inline std::pair<Array2, std::vector<float>> mfe(const float *signal, std::size_t n, const SpeechConfig &cfg)
{
    std::size_t t = 0;
    check(ss_num_frames(&cfg.params(), n, &t));
    Array2 feat{t, cfg.num_filters(), std::vector<float>(t * cfg.num_filters())};
    std::vector<float> energy(t);
    check(ss_mfe(cfg.handle(), signal, n, feat.data.data(), energy.data()));
    return {std::move(feat), std::move(energy)};
}

// feature.rs:163-174: signal [channels x n] -> [channels x num_filters x rows]
inline Array3 mel_spectrogram2(const float *signal, std::size_t channels, std::size_t n, const SpeechConfig &cfg)
{
    std::size_t rows = 0, real_rows = 0;
    check(ss_stft_rows(&cfg.params(), n, &rows, &real_rows));
    Array3 out{channels, cfg.num_filters(), rows, std::vector<float>(channels * cfg.num_filters() * rows)};
    check(ss_mel_spectrogram(cfg.handle(), signal, channels, n, out.data.data()));
    return out;
}

// feature.rs:151-162 (one channel; the reference's axis mix-up is not reproduced)
inline Array2 mel_spectrogram1(const float *signal, std::size_t n, const SpeechConfig &cfg)
{
    Array3 a = mel_spectrogram2(signal, 1, n, cfg);
    return Array2{a.d1, a.d2, std::move(a.data)};
}

// functions.rs:86-123: Array3<Complex32> [channels x rows x freq_size]
struct ComplexArray3 {
    std::size_t d0 = 0, d1 = 0, d2 = 0;
    std::vector<std::complex<float>> data;
};
inline ComplexArray3 stft2(const float *signal, std::size_t channels, std::size_t n, const SpeechConfig &cfg)
{
    std::size_t rows = 0, real_rows = 0;
    check(ss_stft_rows(&cfg.params(), n, &rows, &real_rows));
    ComplexArray3 out{channels, rows, cfg.freq_size(), std::vector<std::complex<float>>(channels * rows * cfg.freq_size())};
    // std::complex<float> is layout-compatible with float[2] (re, im): the interleaved block the ABI writes
    check(ss_stft(cfg.handle(), signal, channels, n, reinterpret_cast<float *>(out.data.data())));
    return out;
}
// functions.rs:199-233 (one channel): [rows x freq_size]
inline ComplexArray3 stft1(const float *signal, std::size_t n, const SpeechConfig &cfg) { return stft2(signal, 1, n, cfg); }

// processing.rs:65-129 with the reference's own argument list:
//   stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding) -> frames [num_frames x frame_len]
// `filter` (may be null) is called with the frame length and returns the window as an Array2 of shape (1, frame_len) -- row 0
// multiplies every frame, what repeat_axis(filt, Axis(0), numframes) does in processing.rs:122-126 (a (frame_len, 1) column is
// read down its column).  No SpeechConfig and no FFT length are involved: any sampling rate and frame length.
using FrameFilter = Array2 (*)(std::size_t);
inline Array2 stack_frames(const float *signal, std::size_t n, std::size_t sample_rate, float frame_length, float frame_stride,
                           FrameFilter filter, bool zero_padding)
{
    std::size_t t = 0, flen = 0;
    check(ss_stack_frames_shape(n, static_cast<uint32_t>(sample_rate), frame_length, frame_stride, zero_padding ? 1 : 0, &t, &flen));
    std::vector<float> window;
    if (filter) {
        const Array2 w = filter(flen);
        if (w.rows * w.cols != flen || (w.rows != 1 && w.cols != 1)) throw Error(SS_ERR_ARG, "stack_frames: filter(frame_len) must give frame_len values");
        window = w.data;
    }
    Array2 out{t, flen, std::vector<float>(t * flen)};
    check(ss_stack_frames_signal(signal, n, static_cast<uint32_t>(sample_rate), frame_length, frame_stride,
                                 filter ? window.data() : nullptr, zero_padding ? 1 : 0, out.data.data()));
    return out;
}
inline Array2 stack_frames(const std::vector<float> &signal, std::size_t sample_rate, float frame_length, float frame_stride,
                           FrameFilter filter, bool zero_padding)
{
    return stack_frames(signal.data(), signal.size(), sample_rate, frame_length, frame_stride, filter, zero_padding);
}

// The same stage with the framing taken from a config (its framing / mfcc_window / pad_mode switches apply: literal and centred
// framing exist only in this form).  An extra beside the reference's signature above.
inline Array2 stack_frames(const float *signal, std::size_t n, const SpeechConfig &cfg)
{
    std::size_t t = 0, flen = 0, step = 0;
    check(ss_num_frames(&cfg.params(), n, &t));
    check(ss_frame_sizes(&cfg.params(), &flen, &step));
    Array2 out{t, flen, std::vector<float>(t * flen)};
    check(ss_stack_frames(cfg.handle(), signal, n, out.data.data()));
    return out;
}

// processing.rs:179-181 on an existing config (fft_points = the config's).  An extra beside the reference's signature below.
inline Array2 power_spectrum(const Array2 &frames, const SpeechConfig &cfg)
{
    Array2 out{frames.rows, cfg.freq_size(), std::vector<float>(frames.rows * cfg.freq_size())};
    check(ss_power_spectrum_frames(cfg.handle(), frames.data.data(), frames.rows, frames.cols, out.data.data()));
    return out;
}

// processing.rs:179-181 with the reference's own argument list: power_spectrum(frames, fft_points).  Only fft_points matters to
// this stage; the config it runs on is kept per fft_points in a small process-wide cache (the Python front memoises the same
// way), guarded by a mutex, so repeated calls cost one lookup.
inline Array2 power_spectrum(const Array2 &frames, std::size_t fft_points)
{
    static std::mutex mu;
    static std::map<std::size_t, std::unique_ptr<SpeechConfig>> cache;
    const SpeechConfig *cfg = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(fft_points);
        if (it == cache.end()) {
            // a config that validates for this FFT length: a frame of half its length, a bank that fits its spectrum
            SpeechConfigBuilder b(16000);
            const std::size_t nf = std::max<std::size_t>(1, std::min<std::size_t>(40, fft_points / 8));
            b.fft_points(fft_points).frame_length(static_cast<float>(fft_points) / 32000.0f).frame_stride(static_cast<float>(fft_points) / 64000.0f)
                .num_filters(nf).num_cepstral(std::min<std::size_t>(13, nf));
            it = cache.emplace(fft_points, std::make_unique<SpeechConfig>(b.build())).first;
        }
        cfg = it->second.get();  // configs are immutable and thread-safe; entries are never removed
    }
    return power_spectrum(frames, *cfg);
}

// processing.rs:31-53
inline std::vector<float> preemphasis(const std::vector<float> &signal, long shift = 1, float cof = 0.98f)
{
    std::vector<float> y(signal.size());
    check(ss_preemphasis(signal.data(), signal.size(), shift, cof, y.data()));
    return y;
}

// ---- post-processing on a row-major [rows x cols] feature matrix ----

// processing.rs:265-300
inline std::vector<float> cmvn(const std::vector<float> &vec, size_t rows, size_t cols, bool variance_normalization = false)
{
    if (vec.size() != rows * cols) throw Error(SS_ERR_ARG, "cmvn: shape mismatch");
    std::vector<float> out(vec.size());
    check(ss_cmvn(vec.data(), rows, cols, variance_normalization ? 1 : 0, out.data()));
    return out;
}

// processing.rs:315-371 (win_size must be odd, as the reference asserts)
inline std::vector<float> cmvnw(const std::vector<float> &vec, size_t rows, size_t cols, size_t win_size = 301,
                                bool variance_normalization = false)
{
    if (vec.size() != rows * cols) throw Error(SS_ERR_ARG, "cmvnw: shape mismatch");
    std::vector<float> out(vec.size());
    check(ss_cmvnw(vec.data(), rows, cols, win_size, variance_normalization ? 1 : 0, out.data()));
    return out;
}

// processing.rs:222-254
inline std::vector<float> derivative_extraction(const std::vector<float> &feat, size_t rows, size_t cols, size_t delta_windows)
{
    if (feat.size() != rows * cols) throw Error(SS_ERR_ARG, "derivative_extraction: shape mismatch");
    std::vector<float> out(feat.size());
    check(ss_derivative_extraction(feat.data(), rows, cols, delta_windows, out.data()));
    return out;
}

// feature.rs:253-269: [rows x cols x 3]
inline std::vector<float> extract_derivative_feature(const std::vector<float> &feature, size_t rows, size_t cols)
{
    if (feature.size() != rows * cols) throw Error(SS_ERR_ARG, "extract_derivative_feature: shape mismatch");
    std::vector<float> cube(3 * feature.size());
    check(ss_extract_derivative_feature(feature.data(), rows, cols, cube.data()));
    return cube;
}

}  // namespace speechsauce

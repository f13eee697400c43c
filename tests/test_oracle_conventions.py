"""The oracle's transform conventions against independent library implementations (CPU only).

The two scalars of the hot path that cannot be read off the reference's own sources are conventions of its dependencies:
`realfft`'s forward transform (unnormalised, e^{-2 pi i kn/N}; processing.rs:146-168 then divides by N) and `ndrustfft::nddct2`
(feature.rs:120-123), which follows scipy.fft.dct(type=2, norm=None): y[k] = 2 sum_n x[n] cos(pi k (2n+1) / 2N).  This file pins
the oracle's hand-written f64 transforms on numpy.fft.rfft and scipy.fft.dct, so that "the oracle agrees with the crate"
reduces to "the crates agree with numpy / scipy", which is what their documentation states.  (tools/ref_dump produces the
reference's own numbers where a Rust toolchain exists; tests/test_reference_vectors.py consumes them.)"""
import numpy as np
import pytest

scipy_fft = pytest.importorskip("scipy.fft")


def _signal(seed, n):
    return (np.random.default_rng(seed).standard_normal(n) * 0.1).astype(np.float32)


@pytest.mark.parametrize("sr,nfft,flen,step", [(16000, 512, 320, 160), (16000, 512, 400, 160), (8000, 256, 160, 80), (44100, 4096, 4096, 1024),
                                              (16000, 400, 400, 160)])
def test_power_spectrum_is_numpy_rfft_over_n(oracle, sr, nfft, flen, step):
    x = _signal(3, flen + 7 * step)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr)
    got = np.asarray(oracle.power_spectrum(p, x), dtype=np.float64)
    T = got.shape[0]
    frames = np.stack([x[t * step:t * step + flen] for t in range(T)]).astype(np.float64)
    want = np.abs(np.fft.rfft(frames, n=nfft, axis=1)) / nfft      # |X| / N: processing.rs:168, :180
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-7 * np.abs(want).max()   # f64 transform; the chirp-free direct DFT of other lengths ~ 2e-8


@pytest.mark.parametrize("M,C", [(40, 13), (256, 40), (26, 26), (128, 20)])
def test_mfcc_columns_are_scipy_dct2_of_the_log_mel_rows(oracle, M, C):
    """feature.rs:99-148 = ln(mfe) -> nddct2 -> scaling -> first C columns (-> column 0 := ln(energy))."""
    sr, nfft = (44100, 4096) if M > 128 else (16000, 512)
    flen, step = (4096, 1024) if M > 128 else (320, 160)
    x = _signal(5, flen + 20 * step)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_filters=M, num_cepstral=C,
                           dc_elimination=False)
    feat, _ = oracle.mfe(p, x)
    feat = np.asarray(feat, dtype=np.float64)
    got = np.asarray(oracle.mfcc(p, x), dtype=np.float64)
    T = feat.shape[0]
    y = scipy_fft.dct(np.log(feat), type=2, norm=None, axis=1)[:, :C]   # scipy: 2 sum x[n] cos(pi k (2n+1) / 2N)
    n = np.float32(T * M)
    y[:, 1:] *= float(np.float32(1.0) / np.sqrt(np.float32(2.0) * n))       # feature.rs:126-131, as written
    y[0, 0] *= float(np.float32(1.0) / np.sqrt(np.float32(4.0) * n))
    assert np.abs(got - y).max() <= 1e-9 * np.abs(y).max()
    # dct_norm = "ortho" is scipy's norm="ortho"
    po = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_filters=M, num_cepstral=C,
                            dc_elimination=False, dct_norm="ortho")
    yo = scipy_fft.dct(np.log(feat), type=2, norm="ortho", axis=1)[:, :C]
    go = np.asarray(oracle.mfcc(po, x), dtype=np.float64)
    assert np.abs(go - yo).max() <= 1e-9 * np.abs(yo).max()


@pytest.mark.parametrize("sr,nfft,flen,hop", [(22050, 2048, 2048, 512), (16000, 512, 400, 160), (44100, 1024, 1024, 256)])
def test_centred_reflect_padded_hann_frames_are_what_numpy_and_scipy_make(oracle, sr, nfft, flen, hop):
    """The librosa-style switches (framing="center", pad_mode="reflect", mfcc_window="hann", spectrum_exponent=2): frame t covers
    flen samples centred on t * hop of the reflect-padded clip (numpy.pad mode="reflect"), times the periodic Hann window
    (scipy.signal.get_window("hann", flen, fftbins=True)), zero-padded to n_fft: |rfft|^2 / N."""
    scipy_signal = pytest.importorskip("scipy.signal")
    x = _signal(7, sr // 4)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=hop / sr, framing="center", pad_mode="reflect",
                           mfcc_window="hann", spectrum_exponent=2)
    got = np.asarray(oracle.power_spectrum(p, x), dtype=np.float64)
    T = got.shape[0]
    assert T == 1 + x.size // hop
    xp = np.pad(x.astype(np.float64), flen // 2, mode="reflect")
    w = scipy_signal.get_window("hann", flen, fftbins=True)
    frames = np.stack([xp[t * hop:t * hop + flen] * w.astype(np.float32).astype(np.float64) for t in range(T)])
    want = np.abs(np.fft.rfft(frames, n=nfft, axis=1)) ** 2 / nfft
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()    # the window is an f32 table in the oracle as in the library

"""Pins the oracle against numbers produced by the REFERENCE crate itself, when they are available.

The reference is Rust and cannot be built in this image; tools/ref_dump/ is a Cargo project a maintainer runs next to a
checkout of the reference ("Pinning the oracle"); its output, packed by tools/ref_dump/pack.py, is
tests/golden/reference_v1.npz.  Until that file exists these tests skip and the oracle stays "parity unpinned" for the
third-party conventions (DCT-II gain of ndrustfft::nddct2, rustfft / realfft forward transforms).  With the file:
every vector below must match the oracle, i.e. the HIP path (which the GPU tests tie to the oracle) matches the reference.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "reference_v1.npz")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools", "ref_dump"))

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="tests/golden/reference_v1.npz not generated (needs cargo + the reference crate)")

CFG = {
    "cfg1": dict(sample_rate=16000),
    "cfg3": dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128, high_frequency=8000.0),
    "cfg5": dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40,
                 num_filters=256, high_frequency=22050.0),
}


@pytest.fixture(scope="module")
def ref():
    return np.load(REF)


def _close(got, want, tol=1e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30)


def test_inputs_are_the_documented_generator(ref):
    from pack import lcg_signal

    np.testing.assert_array_equal(ref["mfcc1_cfg1_in"], lcg_signal(21, 500, 0.1))
    np.testing.assert_array_equal(ref["mfcc_cfg1_in"], lcg_signal(31, 16000, 0.1))
    np.testing.assert_array_equal(ref["stft_cfg3_in"][1], lcg_signal(42, 16000, 0.1))


@pytest.mark.parametrize("name", ["dct2_ramp8", "dct2_ramp40"])
def test_dct2_gain(ref, name):
    """ndrustfft::nddct2 (feature.rs:120-123): y[k] = g * sum x[n] cos(pi k (2n+1) / 2N) with the oracle's g (SS_DCT2_GAIN = 2)."""
    x = ref[name + "_in"][0].astype(np.float64)
    n = len(x)
    k = np.arange(n)[:, None]
    want = 2.0 * (np.cos(np.pi * k * (2 * np.arange(n)[None, :] + 1) / (2 * n)) @ x)
    _close(want, ref[name + "_out"][0], 1e-5)


def test_power_spectrum_frames(ref, oracle):
    """processing::power_spectrum (processing.rs:143-181): |rfft(frame, 512)| / 512 -- rustfft's forward transform is
    unnormalised (the impulse row is exactly 1/512 everywhere)."""
    frames = ref["pspec_in"].astype(np.float64)
    want = np.abs(np.fft.rfft(frames, 512, axis=1)) / 512.0
    _close(want, ref["pspec_out"], 1e-5)


def test_column_zero_scaling_without_dc_elimination(ref, oracle):
    """num_cepstral == num_filters = 40, dc_elimination = false: column 0 keeps its DCT value, scaled by 1/sqrt(4n) in `[0, 0]`
    only (feature.rs:126-131) -- the quirk the default configuration hides behind the ln(energy) replacement."""
    if "mfcc1_full_in" not in ref:
        pytest.skip("reference_v1.npz predates the `full` vector (re-run tools/ref_dump)")
    p = oracle.make_params(sample_rate=16000, fft_points=512, frame_length=0.02, frame_stride=0.01, num_cepstral=40, num_filters=40,
                           high_frequency=8000.0, dc_elimination=False)
    _close(oracle.mfcc(p, ref["mfcc1_full_in"]), ref["mfcc1_full_out"], 1e-4)


@pytest.mark.parametrize("cfg", ["cfg1", "cfg5"])
def test_single_frame_clips_pin_the_whole_chain(ref, oracle, cfg):
    """One frame is the one case where the reference's stack_frames copies the signal (processing.rs:110-120): filterbank,
    zero handling, ln, DCT-II, scaling and the column-0 replacement on real data."""
    p = oracle.make_params(**CFG[cfg])
    x = ref[f"mfcc1_{cfg}_in"]
    _close(oracle.mfcc(p, x), ref[f"mfcc1_{cfg}_out"], 1e-4)
    feat, en = oracle.mfe(p, x)
    _close(feat, ref[f"mfe1_{cfg}_feat"], 1e-5)
    _close(en, ref[f"mfe1_{cfg}_energy"], 1e-5)


@pytest.mark.parametrize("cfg", ["cfg1", "cfg5"])
def test_literal_framing_of_long_clips(ref, oracle, cfg):
    """More than two frames: the reference's frames stay zero; the oracle's `literal` framing
    restates that, the product's default is the documented contract (deviation D1)."""
    p = oracle.make_params(**CFG[cfg], framing="literal")
    _close(oracle.mfcc(p, ref[f"mfcc_{cfg}_in"]), ref[f"mfcc_{cfg}_out"], 1e-5)


def test_stft_and_mel_spectrogram_first_channel(ref, oracle):
    """functions::stft2 / feature::mel_spectrogram2 on a fresh SpeechConfig: channel 0 starts from zero state, which is
    what the product defines for every clip (deviation D3; channel 1 of the reference starts with channel 0's tail)."""
    p = oracle.make_params(**CFG["cfg3"])
    x = ref["stft_cfg3_in"]
    s = oracle.stft(p, x[0])[0]
    _close(np.stack([s.real, s.imag], axis=-1), ref["stft_cfg3_out"][0], 1e-5)
    _close(oracle.mel_spectrogram(p, x[0]), ref["mel_cfg3_out"][0], 1e-4)

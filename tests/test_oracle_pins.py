"""Pins the CPU oracle (oracle/ss_oracle.c) before anything is checked against it:
  * every shape the reference's own tests assert (speechsauce/src/lib.rs:50-68, 93-134),
  * analytic known answers,
  * the committed fixtures generated from the independent numpy restatement (tests/golden/).
The reference publishes no golden vectors, so numeric parity is "unpinned" at the third-party
FFT/DCT conventions (see oracle/ss_oracle.h); these tests pin everything else.
"""
import numpy as np
import pytest

from common import CONFIGS, N_SAMPLES, golden_signals, load_golden, rel


def test_reference_test_shapes(oracle):
    """lib.rs:93-134: 1e6-sample N(0, 0.1) signal, defaults -> mfcc (6248, 13) no NaN; mfe (6248, 40), (6248,)."""
    x = (np.random.default_rng(7).standard_normal(1_000_000) * 0.1).astype(np.float32)
    p = oracle.make_params()
    out = oracle.port_mfcc(p, x)
    assert out.shape == (6248, 13) and not np.isnan(out).any()
    feat, en = oracle.port_mfe(p, x)
    assert feat.shape == (6248, 40) and en.shape == (6248,)
    # lib.rs:50-68: stack_frames(zero_padding=true, stride 0.02) -> ceil((1e6 - 320) / 320) = 3124 frames
    assert oracle.num_frames_padded(oracle.make_params(frame_stride=0.02), 1_000_000) == 3124
    assert oracle.num_frames(p, 1_000_000) == 6248


def test_frame_count_table(oracle):
    g = load_golden()
    p = oracle.make_params()
    got = [oracle.num_frames(p, int(n)) for n in g["frames/lengths"]]
    assert got == list(g["frames/cfg1"])
    for n in (0, 10, 319, 320, 479):  # usize underflow / zero frames in the reference -> error code
        with pytest.raises(oracle.OracleError) as e:
            oracle.num_frames(p, n)
        assert e.value.code == oracle.ORC_ERR_SHORT_SIGNAL


def test_derived_sizes(oracle):
    assert oracle.frame_sizes(oracle.make_params(**CONFIGS["cfg1"])) == (320, 160)
    assert oracle.frame_sizes(oracle.make_params(**CONFIGS["cfg5"])) == (4096, 1024)
    assert oracle.num_frames(oracle.make_params(**CONFIGS["cfg5"]), 44100) == 39
    hop, n_pad, wnorm = oracle.stft_sizes(oracle.make_params(**CONFIGS["cfg3"]))
    assert (hop, n_pad) == (512, 3) and wnorm == pytest.approx(2 * 512 / 2048**2)
    assert oracle.stft_rows(oracle.make_params(**CONFIGS["cfg3"]), 16000) == (32, 29)
    with pytest.raises(oracle.OracleError):  # defaults: fft 512 < 2 * 320 (functions.rs:136 underflow)
        oracle.stft_sizes(oracle.make_params())


@pytest.mark.parametrize("name", ["cfg1", "cfg3", "cfg5"])
def test_filterbank_fixture(oracle, name):
    g = load_golden()
    fb, idx = oracle.filterbank(oracle.make_params(**CONFIGS[name]))
    np.testing.assert_array_equal(idx, g[f"{name}/fb_idx"])
    rows, cols = np.nonzero(fb)
    np.testing.assert_array_equal(rows, g[f"{name}/fb_nz_rows"])
    np.testing.assert_array_equal(cols, g[f"{name}/fb_nz_cols"])
    np.testing.assert_array_equal(fb[rows, cols], g[f"{name}/fb_nz_vals"])
    assert np.isfinite(fb).all() and fb.max() == 1.0


def test_filterbank_known_structure(oracle):
    fb, idx = oracle.filterbank(oracle.make_params())
    # default-config indices; only the lower half of the spectrum is weighted (Q4)
    assert list(idx[:6]) == [0, 0, 1, 2, 3, 4] and idx[-1] == 129
    assert np.count_nonzero(fb) == 210 and not fb[:, 130:].any()
    fb5, idx5 = oracle.filterbank(oracle.make_params(**CONFIGS["cfg5"]))
    assert (np.count_nonzero(fb5, axis=1) == 0).sum() == 19  # empty filters -> EPS after zero handling


def test_vorbis_window(oracle):
    g = load_golden()
    for name in CONFIGS:
        n = CONFIGS[name].get("fft_points", 512)
        np.testing.assert_array_equal(oracle.vorbis_window(n), g[f"{name}/vorbis_window"])
    w = oracle.vorbis_window(512).astype(np.float64)
    np.testing.assert_allclose(w[:256] ** 2 + w[256:] ** 2, 1.0, atol=1e-6)  # power-complementary


@pytest.mark.parametrize("name", ["cfg1", "cfg5"])
def test_mfcc_against_fixtures(oracle, name):
    g = load_golden()
    p = oracle.make_params(**CONFIGS[name])
    for sname, x in golden_signals(N_SAMPLES[name], CONFIGS[name]["sample_rate"]).items():
        got = oracle.mfcc(p, x)
        assert rel(got, g[f"{name}/{sname}/mfcc"]) < 1e-9, sname
        feat, en = oracle.mfe(p, x)
        assert rel(en, g[f"{name}/{sname}/energy"]) < 1e-9
        assert rel(feat[[0, feat.shape[0] // 2, -1]], g[f"{name}/{sname}/feat_rows"]) < 1e-9
        assert rel(oracle.power_spectrum(p, x)[1], g[f"{name}/{sname}/P_row1"]) < 1e-9
        # the reference-shaped f32 port stays within the path's tolerance of the f64 oracle
        assert rel(oracle.port_mfcc(p, x), got) < 2e-5, sname


def test_mel_against_fixtures(oracle):
    g = load_golden()
    p = oracle.make_params(**CONFIGS["cfg3"])
    for sname, x in golden_signals(16000, 16000).items():
        got = oracle.mel_spectrogram(p, x)
        assert got.shape == (128, 32)
        assert rel(got, g[f"cfg3/{sname}/mel"]) < 1e-9, sname
        assert not got[:, 29:].any()  # trailing n_pad rows never written (functions.rs:121)
        S = oracle.stft(p, x)[0]
        assert rel(np.stack([S[5].real, S[5].imag]), g[f"cfg3/{sname}/stft_row5"]) < 1e-9
        assert rel(oracle.port_mel_spectrogram(p, x), got) < 2e-5


def test_switch_fixtures(oracle):
    g = load_golden()
    x = golden_signals(16000, 16000)["noise"]
    for tag, sw in {"pow2": dict(spectrum_exponent=2), "ortho": dict(dct_norm="ortho"), "hann": dict(mfcc_window="hann"),
                    "preemph": dict(preemph_coef=0.97), "nodc": dict(dc_elimination=False),
                    "literal": dict(framing="literal")}.items():
        got = oracle.mfcc(oracle.make_params(**sw), x)
        assert rel(got, g[f"switch/{tag}"]) < 1e-9, tag
    assert rel(oracle.preemphasis(x[:1000], 1, 0.98), g["preemphasis/shift1_cof0.98"]) < 1e-7


def test_known_answer_impulse(oracle):
    """One unit impulse at the start of every frame: |X[k]| = 1 for all k -> P = 1/N, E = F/N."""
    p = oracle.make_params()
    x = np.zeros(16000, np.float32)
    x[::160] = 1.0
    P = oracle.power_spectrum(p, x)
    # frame t holds impulses at offsets 0 and 160: |X[k]| = |1 + e^{-2 pi i 160 k / 512}|
    k = np.arange(257)
    np.testing.assert_allclose(P[3], np.abs(1 + np.exp(-2j * np.pi * 160 * k / 512)) / 512, atol=1e-12)
    x1 = np.zeros(16000, np.float32)
    x1[::320] = 1.0  # even frames see one impulse at offset 0, odd frames one at offset 160
    P1 = oracle.power_spectrum(p, x1)
    np.testing.assert_allclose(P1[0], 1.0 / 512, atol=1e-12)
    feat, en = oracle.mfe(p, x1)
    np.testing.assert_allclose(en[0], 257 / 512, rtol=1e-12)


def test_known_answer_zero_signal_and_literal_framing(oracle):
    """P = 0 -> zero_handling -> f32::EPSILON -> ln(EPS) = -15.942385 in column 0, DCT of a constant row = 0
    elsewhere.  The literal exact_chunks framing (processing.rs:110-120) gives the same for ANY signal (Q1)."""
    p = oracle.make_params()
    out = oracle.mfcc(p, np.zeros(16000, np.float32))
    np.testing.assert_allclose(out[:, 0], np.log(np.float32(1.1920929e-7)), rtol=1e-7)
    assert np.abs(out[:, 1:]).max() < 1e-12
    x = golden_signals(16000, 16000)["noise"]
    lit = oracle.mfcc(oracle.make_params(framing="literal"), x)
    np.testing.assert_allclose(lit, out, atol=1e-12)
    # with <= 2 frames the literal code copies x[0:flen] into every row
    short = x[:640]  # 2 frames
    lit2 = oracle.mfcc(oracle.make_params(framing="literal"), short)
    np.testing.assert_allclose(lit2[0, 1:], lit2[1, 1:], atol=1e-12)


def test_dct_scaling_quirk(oracle):
    """feature.rs:126-131: n = T*M, [[0,0]] *= 1/sqrt(4n), columns 1.. *= 1/sqrt(2n), column 0 of rows >= 1 unscaled."""
    x = golden_signals(16000, 16000)["noise"]
    a = oracle.mfcc(oracle.make_params(dc_elimination=False), x)
    b = oracle.mfcc(oracle.make_params(dc_elimination=False, dct_norm="ortho"), x)
    T, M = 98, 40
    np.testing.assert_allclose(a[:, 1:] * np.sqrt(2 * T * M), b[:, 1:] * np.sqrt(2 * M), rtol=1e-6)
    np.testing.assert_allclose(a[0, 0] * np.sqrt(4 * T * M), b[0, 0] * np.sqrt(4 * M), rtol=1e-6)
    np.testing.assert_allclose(a[1:, 0], b[1:, 0] * np.sqrt(4 * M), rtol=1e-6)  # unscaled in reference mode
    # gain is one named constant
    c = oracle.mfcc(oracle.make_params(dct2_gain=1.0), x)
    np.testing.assert_allclose(c[:, 1:] * 2, oracle.mfcc(oracle.make_params(), x)[:, 1:], rtol=1e-6)


def test_c_oracle_matches_numpy_twin(oracle):
    import oracle_np as on

    rng = np.random.default_rng(3)
    for kw, n in [(dict(sample_rate=8000, fft_points=256, num_filters=26, num_cepstral=12), 5000),
                  (dict(sample_rate=16000, fft_points=1024, frame_length=0.025, frame_stride=0.0125, num_filters=64,
                        num_cepstral=20, low_frequency=100.0, high_frequency=7000.0), 9000),
                  (dict(sample_rate=22050, fft_points=96 * 4, frame_length=0.01, frame_stride=0.005), 4000)]:
        x = (rng.standard_normal(n) * 0.2).astype(np.float32)
        assert rel(oracle.mfcc(oracle.make_params(**kw), x), on.mfcc(on.Params(**kw), x)) < 1e-9
    kw = dict(sample_rate=16000, fft_points=1024, frame_length=0.016, frame_stride=0.016, num_filters=80)
    x = (rng.standard_normal((3, 7000)) * 0.2).astype(np.float32)
    assert rel(oracle.mel_spectrogram(oracle.make_params(**kw), x), on.mel_spectrogram(on.Params(**kw), x)) < 1e-9


def test_bad_configs(oracle):
    for kw in (dict(high_frequency=9000.0), dict(low_frequency=-1.0)):  # feature.rs:47-51 asserts
        with pytest.raises(oracle.OracleError) as e:
            oracle.filterbank(oracle.make_params(**kw))
        assert e.value.code == oracle.ORC_ERR_BAD_CONFIG
    with pytest.raises(oracle.OracleError):  # num_cepstral > num_filters: slice panic (feature.rs:133)
        oracle.mfcc(oracle.make_params(num_cepstral=41), np.zeros(16000, np.float32))
    with pytest.raises(oracle.OracleError):  # frame longer than fft_points (processing.rs:146-164)
        oracle.mfcc(oracle.make_params(frame_length=0.04), np.zeros(16000, np.float32))


def test_the_timed_cpu_port_is_built_for_this_host_and_checked_first(oracle):
    """bench.py's cpu_baseline leg: the reference-shaped f32 port is compiled ON the host that times it (`make -C oracle native`:
    -O3 -march=native, no fast-math, no FMA contraction; oracle/_native/ is git- and gpurun-ignored so it never travels), compared
    with the f64 checker on one clip before any timing, and the flags are named in the line."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    res = bench.cpu_baseline("mfcc", dict(sample_rate=16000), 16000, budget_s=0.3)
    assert res["kind"] == "port" and res["cores"] == 1 and res["value"] > 1e3
    assert ("-O3 -march=native" in res["sample"] and "-ffp-contract=off" in res["sample"]) or "portable" in res["sample"]
    nat, flags = oracle.native_port()
    if nat is not None:
        x = (np.random.default_rng(2).standard_normal(16000) * 0.1).astype(np.float32)
        p = oracle.make_params()
        np.testing.assert_array_equal(oracle.port_mfcc(p, x, from_lib=nat).shape, (98, 13))
        want = oracle.mfcc(p, x)
        assert np.abs(oracle.port_mfcc(p, x, from_lib=nat) - want).max() <= 1e-4 * np.abs(want).max()
    ign = open(os.path.join(root, ".gpurunignore")).read()
    assert "oracle/_native/" in ign and "oracle/_native/" in open(os.path.join(root, ".gitignore")).read()

"""Round-2 exports next to the hot path: lmfe (feature.rs:242-245), stack_frames(zero_padding = true) as a framing switch
(processing.rs:85-97; the reference's own test_stack_frames, lib.rs:50-68) and librosa's power_to_db."""
import numpy as np
import pytest

from common import rel as _rel


def _signal(seed, shape, scale=0.1):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def _power_to_db_ref(S, ref=1.0, amin=1e-10, top_db=80.0):
    """librosa.power_to_db restated (librosa/core/spectrum.py): magnitude = S, ref scalar."""
    S = np.asarray(S, np.float64)
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, abs(ref)))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


# ---- CPU: frame counts of the padded framing (oracle and host library agree with the reference's test) ----
def test_padded_frame_count_of_the_reference_test(oracle, sslib):
    import ctypes as C

    from speechsauce_amd import make_params

    # lib.rs:50-68: 1e6 samples @16 kHz, 20 ms / 20 ms, zero_padding = true -> ceil((1e6 - 320) / 320) = 3124 frames
    p = oracle.make_params(sample_rate=16000, frame_length=0.02, frame_stride=0.02, framing="padded")
    assert oracle.num_frames(p, 1_000_000) == 3124 == oracle.num_frames_padded(p, 1_000_000)
    sp = make_params(sample_rate=16000, frame_length=0.02, frame_stride=0.02, framing="padded")
    t = C.c_size_t()
    assert sslib.ss_num_frames(C.byref(sp), 1_000_000, C.byref(t)) == 0 and t.value == 3124
    # defaults: 16000 samples, 20 ms / 10 ms -> floor gives 98, ceil gives 98 as well ((16000-320)/160 = 98 exactly); 16001 -> 99
    sp = make_params(sample_rate=16000, framing="padded")
    assert sslib.ss_num_frames(C.byref(sp), 16001, C.byref(t)) == 0 and t.value == 99
    assert sslib.ss_num_frames(C.byref(sp), 100, C.byref(t)) == 1  # SS_ERR_SHORT_SIGNAL: shorter than one frame


def test_padded_oracle_last_frame_reads_zeros(oracle):
    x = _signal(3, 16100)
    p = oracle.make_params(sample_rate=16000, framing="padded")
    T = oracle.num_frames(p, 16100)
    assert T == 99
    P = oracle.power_spectrum(p, x)
    xz = np.concatenate([x, np.zeros(400, np.float32)])
    pc = oracle.make_params(sample_rate=16000)  # contract framing of the explicitly padded signal
    Pc = oracle.power_spectrum(pc, xz)
    np.testing.assert_allclose(P, Pc[:T], rtol=0, atol=0)


# ---- GPU ----
@pytest.mark.gpu
def test_lmfe_matches_ln_of_mfe(ss, oracle):
    import torch

    x = _signal(21, (5, 16000))
    got = ss.lmfe_batch(torch.from_numpy(x).cuda(), 16000).cpu().numpy()
    assert got.shape == (5, 98, 40)
    p = oracle.make_params(sample_rate=16000)
    for b in range(5):
        feat, _ = oracle.mfe(p, x[b])
        want = np.log(feat)
        assert np.abs(got[b] - want).max() <= 1e-4 * np.abs(want).max()
    one = ss.lmfe(x[0], 16000)  # host-array entry point
    np.testing.assert_allclose(one, got[0], rtol=0, atol=1e-6)
    # an all-zero clip: every energy is f32::EPSILON after zero_handling, so every value is ln(EPS)
    z = ss.lmfe(np.zeros(16000, np.float32), 16000)
    np.testing.assert_allclose(z, np.log(np.float32(1.1920929e-7)), rtol=0, atol=2e-6)
    # the wide configuration (cfg5) goes through its own mfe build
    x5 = _signal(22, 44100)
    kw = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_filters=256, fft_length=4096, high_frequency=22050.0)
    got5 = ss.lmfe(x5, 44100, **kw)
    f5, _ = oracle.mfe(oracle.make_params(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                                          num_filters=256, high_frequency=22050.0), x5)
    assert np.abs(got5 - np.log(f5)).max() <= 1e-4 * np.abs(np.log(f5)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [16100, 16001, 1_000_000])
def test_padded_framing_mfcc(ss, oracle, n):
    x = _signal(n, n)
    kw = dict(frame_length=0.02, frame_stride=0.02) if n == 1_000_000 else {}
    got = ss.mfcc(x, 16000, framing="padded", **kw)
    p = oracle.make_params(sample_rate=16000, framing="padded", **kw)
    want = oracle.mfcc(p, x)
    assert got.shape == want.shape
    if n == 1_000_000:
        assert got.shape == (3124, 13)  # lib.rs:50-68
    assert np.isfinite(got).all()
    assert np.abs(got[:, 0] - want[:, 0]).max() <= 1e-4 * np.abs(want[:, 0]).max()
    assert np.abs(got[:, 1:] - want[:, 1:]).max() <= 1e-4 * np.abs(want[:, 1:]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("top_db", [80.0, 20.0, None])
def test_power_to_db(ss, top_db):
    import torch

    rng = np.random.default_rng(5)
    S = (rng.standard_normal((128, 301)) ** 2 * 10.0 ** rng.uniform(-12, 2, (128, 301))).astype(np.float32)
    S[3, 7] = 0.0
    want = _power_to_db_ref(S, ref=1.0, amin=1e-10, top_db=top_db)
    got = ss.power_to_db(torch.from_numpy(S).cuda(), top_db=top_db).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
    host = ss.power_to_db(S, top_db=top_db)
    np.testing.assert_array_equal(host, got)
    got2 = ss.power_to_db(S, ref=float(S.max()), amin=1e-6, top_db=top_db)
    np.testing.assert_allclose(got2, _power_to_db_ref(S, ref=float(S.max()), amin=1e-6, top_db=top_db), rtol=0, atol=2e-4)
    # |ref| is what counts (librosa: np.abs(ref)); a negative reference is legal
    np.testing.assert_array_equal(ss.power_to_db(S, ref=-float(S.max()), amin=1e-6, top_db=top_db), got2)
    with pytest.raises(ss.SpeechSauceError):
        ss.power_to_db(S, amin=0.0)


@pytest.mark.gpu
def test_mel_spectrogram_in_db(ss, oracle):
    """power_to_db on the mel-spectrogram the device just produced (no host round trip): a librosa-style log-mel front end."""
    import torch

    x = _signal(9, (3, 16000))
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    mel = ss.mel_spectrogram(torch.from_numpy(x).cuda(), 16000, **kw)
    db = ss.power_to_db(mel, ref=1.0, top_db=80.0).cpu().numpy()
    p = oracle.make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128, high_frequency=8000.0)
    want = _power_to_db_ref(oracle.mel_spectrogram(p, x), top_db=80.0)
    np.testing.assert_allclose(db, want, rtol=0, atol=5e-4)


@pytest.mark.gpu
def test_shader_clock_diagnostic_is_per_call_and_leaves_results_alone(ss, sslib):
    """ss_mfcc_shader_clock (the product library's own diagnostic, what bench.py turns into roofline.clock_ghz_measured): a
    plausible clock for the 512-point kernel, SS_ERR_UNSUPPORTED for a configuration on another kernel, and no trace left
    behind -- launches before and after give the same bits."""
    import ctypes as C

    import torch

    from speechsauce_amd import SpeechConfig, make_params

    x = torch.randn((1024, 16000), device="cuda") * 0.1
    want = ss.mfcc_batch(x, 16000)
    cfg = SpeechConfig(make_params(sample_rate=16000))
    out = torch.empty_like(want)
    ghz = C.c_float(0.0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert sslib.ss_mfcc_shader_clock(cfg.handle, x.data_ptr(), 1024, 16000, 16000, out.data_ptr(), stream, 5, C.byref(ghz)) == 0
    assert 1.0 < ghz.value < 3.0, ghz.value
    assert torch.equal(out, want)                                   # the stamped launches compute the same features
    assert torch.equal(ss.mfcc_batch(x, 16000), want)               # ... and nothing stays switched on
    cfg5 = SpeechConfig(make_params(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                                    num_cepstral=40, num_filters=256))
    x5 = torch.randn((4, 44100), device="cuda") * 0.1
    o5 = torch.empty((4, 39, 40), device="cuda")
    assert sslib.ss_mfcc_shader_clock(cfg5.handle, x5.data_ptr(), 4, 44100, 44100, o5.data_ptr(), stream, 2, C.byref(ghz)) == 5  # SS_ERR_UNSUPPORTED
    assert sslib.ss_mfcc_shader_clock(cfg.handle, x.data_ptr(), 1024, 16000, 16000, out.data_ptr(), stream, 0, C.byref(ghz)) == 3  # SS_ERR_ARG


@pytest.mark.gpu
def test_shader_clock_probe_beside_a_workload(ss, sslib):
    """ss_shader_clock_probe / _async (ABI 6): one sleeping wave that reads the shader clock beside whatever else runs.  Alone on
    the device it reads a plausible clock; queued ahead of cfg5 launches (whose workgroups take all of a CU's LDS: the probe needs
    none) it still runs beside them and the launches' results are untouched; argument checks."""
    import ctypes as C

    import torch

    ghz = C.c_float(0.0)
    side = torch.cuda.Stream()
    assert sslib.ss_shader_clock_probe(C.c_void_p(side.cuda_stream), 300, C.byref(ghz)) == 0
    assert 0.05 < ghz.value < 2.6, ghz.value   # (an idle part may sit far below its peak clock)
    assert sslib.ss_shader_clock_probe(C.c_void_p(side.cuda_stream), 5, C.byref(ghz)) == 3       # SS_ERR_ARG: too short
    assert sslib.ss_shader_clock_probe(C.c_void_p(side.cuda_stream), 300, None) == 3
    assert sslib.ss_shader_clock_probe_async(C.c_void_p(side.cuda_stream), 300, None) == 3
    kw = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096, high_frequency=22050.0)
    x = torch.randn((512, 44100), device="cuda") * 0.1
    want = ss.mfcc_batch(x, 44100, **kw)
    torch.cuda.synchronize()
    words = torch.zeros(2, dtype=torch.int64, device="cuda")
    assert sslib.ss_shader_clock_probe_async(C.c_void_p(side.cuda_stream), 1000, C.c_void_p(words.data_ptr())) == 0
    outs = [ss.mfcc_batch(x, 44100, **kw) for _ in range(40)]   # ~2.4 ms of launches behind the probe
    torch.cuda.synchronize()
    cyc, ticks = words.tolist()
    assert ticks >= 100000 and 0.3 < cyc / (10.0 * ticks) < 2.6, (cyc, ticks)   # >= 1000 us on the 100 MHz counter
    assert all(torch.equal(o, want) for o in outs)


@pytest.mark.gpu
def test_lab_stamp_buffer_describes_a_sane_launch(ss, sslab):
    """ss_debug_stamp_buffer (LAB library; tools/prof2.py, tools/dbg_times.py): while set, launches of the 512-point MFCC kernel
    write per-wave stamps; the features must not change, and the stamps must describe a sane launch."""
    import torch

    with ss._lib.use_library(sslab):
        x = torch.randn((1024, 16000), device="cuda") * 0.1
        want = ss.mfcc_batch(x, 16000)
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        stamps = torch.zeros((ncu * 16, 6), dtype=torch.int64, device="cuda")
        assert sslab.ss_debug_stamp_buffer(stamps.data_ptr()) == 0
        try:
            got = ss.mfcc_batch(x, 16000)
            torch.cuda.synchronize()
        finally:
            assert sslab.ss_debug_stamp_buffer(None) == 0
        assert torch.equal(got, want)
        w = stamps.cpu().numpy()
        ran = w[:, 2] != 0
        assert ran.sum() == ncu * 12                                  # twelve waves per workgroup, one workgroup per CU
        assert int((w[ran, 3] >> 32).sum()) == 1024 * 98 // 4          # every quad was claimed exactly once
        stamps.zero_()
        ss.mfcc_batch(x, 16000)                                         # switched off again: nothing is written
        torch.cuda.synchronize()
        assert int(stamps.abs().sum().item()) == 0


@pytest.mark.gpu
def test_device_entry_points_are_graph_capturable(ss, oracle, sslib):
    """The device-pointer entry points queue work on the caller's stream and nothing else (no synchronisation, no allocation,
    no host read of device results), so a caller can capture a whole feature pipeline into a HIP graph and replay it:
    MFCC (the 512-point kernel) and the 2048-point mel spectrogram, captured once, replayed on new inputs."""
    import torch

    x = torch.from_numpy(_signal(71, (8, 16000))).cuda()
    ss.mfcc_batch(x, 16000)  # configs and tables are created outside the capture (the lru-cached config does allocate)
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048)
    ss.mel_spectrogram(x, 16000, **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        feat = ss.mfcc_batch(x, 16000)
        mel = ss.mel_spectrogram(x, 16000, **kw)
    names = set()
    for seed in (72, 73):
        x.copy_(torch.from_numpy(_signal(seed, (8, 16000))).cuda())
        g.replay()
        torch.cuda.synchronize()
        xn = x.cpu().numpy()
        want = oracle.mfcc(oracle.make_params(), xn[3])
        assert _rel(feat[3].cpu().numpy(), want) <= 1e-4
        pm = oracle.make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128)
        wm = oracle.mel_spectrogram(pm, xn[5:6])[0]
        assert _rel(mel[5].cpu().numpy(), wm) <= 1e-4
        names.add(sslib.ss_last_kernel_name())
    assert feat.shape == (8, 98, 13) and mel.shape == (8, 128, 32)

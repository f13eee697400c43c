"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star: "within 1e-4 relative tolerance"): per clip,
max|got - want| <= 1e-4 * max|want|  (cepstra cross zero, so element-wise relative error is
ill-posed).
"""
import ctypes as C

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-4

CFG1 = dict(sample_rate=16000)  # SpeechConfig defaults: n_fft 512, hop 160, 40 mels, 13 ceps
CFG3 = dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_cepstral=13,
            num_filters=128, low_frequency=0.0, high_frequency=8000.0)
CFG5 = dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
            num_cepstral=40, num_filters=256, low_frequency=0.0, high_frequency=22050.0)


def _signal(seed, shape, scale=0.1):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def _rel(got, want):
    return float(np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def _cfg(ss, **kw):
    from speechsauce_amd import SpeechConfig, make_params

    return SpeechConfig(make_params(**kw))


def test_cfg1_single_clip_defaults(ss, oracle):
    x = _signal(0, 16000)
    got = ss.mfcc(x, 16000)
    want = oracle.mfcc(oracle.make_params(**CFG1), x)
    assert got.shape == (98, 13) and got.dtype == np.float32
    assert _rel(got, want) <= RTOL
    # sanity values recorded in SURVEY.md 8c for this seed
    np.testing.assert_allclose(got[0, :3], [-0.19977, -0.45368, -0.00118], atol=2e-4)


@pytest.mark.parametrize("n_fft", [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_mfcc_all_fft_sizes(ss, oracle, n_fft):
    sr = 16000
    fl = min(0.02, n_fft / sr)
    kw = dict(sample_rate=sr, fft_points=n_fft, frame_length=fl, frame_stride=fl / 2, num_filters=20, num_cepstral=12)
    x = _signal(n_fft, 9000)
    got = ss.mfcc(x, sr, frame_length=fl, frame_stride=fl / 2, num_cepstral=12, num_filters=20, fft_length=n_fft)
    want = oracle.mfcc(oracle.make_params(**kw), x)
    assert got.shape == want.shape
    assert _rel(got, want) <= RTOL


def test_power_spectrum_stage(ss, oracle, sslib):
    import torch

    x = _signal(3, (3, 16000))
    cfg = _cfg(ss, **CFG1)
    T = cfg.num_frames(16000)
    xd = torch.from_numpy(x).cuda()
    P = torch.empty((3, T, 257), dtype=torch.float32, device="cuda")
    from speechsauce_amd import _lib

    _lib.check(sslib.ss_power_spectrum_batch_device(cfg.handle, xd.data_ptr(), 3, 16000, 16000, P.data_ptr(), None))
    torch.cuda.synchronize()
    p = oracle.make_params(**CFG1)
    for b in range(3):
        assert _rel(P[b].cpu().numpy(), oracle.power_spectrum(p, x[b])) <= 1e-5
    assert b"power" in sslib.ss_last_kernel_name()  # the power-spectrum build of the 512-point kernel
    # same rows through the generic kernel's configuration space (power = 2 switch)
    cfg2 = _cfg(ss, **CFG1, spectrum_exponent=2)
    P.fill_(7.0)
    _lib.check(sslib.ss_power_spectrum_batch_device(cfg2.handle, xd.data_ptr(), 3, 16000, 16000, P.data_ptr(), None))
    torch.cuda.synchronize()
    p2 = oracle.make_params(**CFG1, spectrum_exponent=2)
    assert _rel(P[1].cpu().numpy(), oracle.power_spectrum(p2, x[1])) <= 1e-5


def test_mfe(ss, oracle):
    x = _signal(4, 16000)
    feat, en = ss.mfe(x, 16000)
    wf, we = oracle.mfe(oracle.make_params(**CFG1), x)
    assert feat.shape == (98, 40) and en.shape == (98,)
    assert _rel(feat, wf) <= 1e-5 and _rel(en, we) <= 1e-5


def test_mfe_batch_fast_path(ss, oracle, sslib):
    """mfe (feature.rs:200-233) over a batch: served by the mfe-output build of the 512-point kernel."""
    import torch

    x = _signal(8, (96, 16000))
    feat, en = ss.mfe_batch(torch.from_numpy(x).cuda(), 16000)
    assert feat.shape == (96, 98, 40) and en.shape == (96, 98)
    assert b"mfe" in sslib.ss_last_kernel_name()
    p = oracle.make_params(**CFG1)
    for b in (0, 50, 95):
        wf, we = oracle.mfe(p, x[b])
        assert _rel(feat[b].cpu().numpy(), wf) <= RTOL and _rel(en[b].cpu().numpy(), we) <= RTOL
    # all-zero clip: every value is exactly f32::EPSILON (functions.rs:66-71)
    z = torch.zeros((3, 16000), dtype=torch.float32, device="cuda")
    feat, en = ss.mfe_batch(z, 16000)
    assert torch.all(feat == 1.1920929e-7) and torch.all(en == 1.1920929e-7)


# kernel builds that `bench.py --workload cfgN` launches (the names ss_last_kernel_name() reports; the rocprofv3 traces under
# profiles/ carry the same builds under their template spelling, profiles/pmc_traffic.json "kernel_full")
from common import BENCH_KERNELS  # noqa: E402  (the builds bench.py's workloads run on)


def test_cfg2_batch_1024(ss, oracle):
    import torch

    x = _signal(1, (1024, 16000))
    got = ss.mfcc_batch(torch.from_numpy(x).cuda(), 16000).cpu().numpy()
    assert got.shape == (1024, 98, 13)
    # the build bench.py times and profiles/ traces for this workload: a dispatch change must not move the bench onto a
    # build this comparison does not see
    assert ss._lib.lib().ss_last_kernel_name() == BENCH_KERNELS["cfg2"], ss._lib.lib().ss_last_kernel_name()
    p = oracle.make_params(**CFG1)
    worst = 0.0
    for b in list(range(0, 1024, 37)) + [1023]:
        worst = max(worst, _rel(got[b], oracle.mfcc(p, x[b])))
    assert worst <= RTOL
    # host-pointer entry point gives the same bits as the device one
    host = ss.mfcc_batch(x[:8], 16000)
    np.testing.assert_array_equal(host, got[:8])


def test_cfg3_mel_spectrogram(ss, oracle):
    import torch

    x = _signal(1, (16, 16000))
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    got = ss.mel_spectrogram(torch.from_numpy(x).cuda(), 16000, **kw).cpu().numpy()
    want = oracle.mel_spectrogram(oracle.make_params(**CFG3), x)
    assert got.shape == (16, 128, 32)
    for b in range(16):
        assert _rel(got[b], want[b]) <= RTOL
    assert np.all(got[:, :, 29:] == 0.0)  # trailing n_pad rows are never written (functions.rs:121)
    one = ss.mel_spectrogram(x[0], 16000, **kw)
    assert one.shape == (128, 32)
    np.testing.assert_array_equal(one, got[0])


def test_mel_spectrogram_512_kernel(ss, oracle, sslib):
    """mel_spectrogram at fft_points = 512 (16 kHz, 16 ms / 8 ms chunks, 40 / 64 / 80 mels, reference and Slaney banks): the
    four-rows-per-wave kernel; clip lengths that leave partial last chunks, row counts that are not multiples of 4."""
    import torch

    for n, frame, M, sw in ((16000, 0.016, 40, {}), (15872, 0.008, 64, {}), (16000, 0.016, 80, dict(mel_scale="slaney", mel_norm="slaney")),
                            (4100, 0.010, 23, dict(high_frequency=3800.0)), (700, 0.016, 40, {})):
        x = _signal(41, (7, n))
        kw = dict(frame_length=frame, frame_stride=frame, num_filters=M, fft_length=512)
        got = ss.mel_spectrogram(torch.from_numpy(x).cuda(), 16000, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name() == b"ss_mel_c256", (sslib.ss_last_kernel_name(), n, frame, M)
        p = oracle.make_params(sample_rate=16000, fft_points=512, frame_length=frame, frame_stride=frame, num_filters=M, **sw)
        want = oracle.mel_spectrogram(p, x)
        assert got.shape == want.shape
        for b in range(7):
            assert _rel(got[b], want[b]) <= RTOL, (n, frame, M, sw, b)
        one = ss.mel_spectrogram(x[3], 16000, **kw, **sw)
        np.testing.assert_array_equal(one, got[3])


def test_mel_spectrogram_1024_kernel(ss, oracle, sslib):
    """mel_spectrogram at fft_points = 1024 (32 / 16 ms chunks at 16 kHz, 22.05 kHz with a Slaney bank over the whole spectrum):
    two rows per wave on the 1024-point FFT mapping; partial last chunks, odd row counts, odd filter counts."""
    import torch

    for sr, n, frame, M, sw in ((16000, 16000, 0.032, 80, {}), (16000, 15872, 0.016, 64, {}), (22050, 22050, 512 / 22050, 128, dict(mel_scale="slaney", mel_norm="slaney")),
                                (16000, 4100, 0.020, 23, dict(high_frequency=3800.0)), (16000, 1500, 0.032, 41, {})):
        x = _signal(43, (5, n))
        kw = dict(frame_length=frame, frame_stride=frame, num_filters=M, fft_length=1024)
        got = ss.mel_spectrogram(torch.from_numpy(x).cuda(), sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name() == b"ss_mel_c512", (sslib.ss_last_kernel_name(), n, frame, M)
        p = oracle.make_params(sample_rate=sr, fft_points=1024, frame_length=frame, frame_stride=frame, num_filters=M, **sw)
        want = oracle.mel_spectrogram(p, x)
        assert got.shape == want.shape
        for b in range(5):
            assert _rel(got[b], want[b]) <= RTOL, (sr, n, frame, M, sw, b)
        one = ss.mel_spectrogram(x[2], sr, **kw, **sw)
        np.testing.assert_array_equal(one, got[2])


def test_mel_spectrogram_2048_full_spectrum_bank(ss, oracle, sslib):
    """mel_spectrogram at fft_points = 2048 with banks that reach past (F+1)/2 (Slaney / HTK scale up to fs/2): the fullp build of
    the cfg3 kernel keeps all 1025 bins of a row."""
    import torch

    sr = 22050
    x = _signal(47, (4, sr))
    for sw, M in ((dict(mel_scale="slaney", mel_norm="slaney"), 128), (dict(mel_scale="htk"), 64)):
        kw = dict(frame_length=512 / sr, frame_stride=512 / sr, num_filters=M, fft_length=2048)
        got = ss.mel_spectrogram(torch.from_numpy(x).cuda(), sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name() == b"ss_mel_c1024<fullp>", sslib.ss_last_kernel_name()
        p = oracle.make_params(sample_rate=sr, fft_points=2048, frame_length=512 / sr, frame_stride=512 / sr, num_filters=M, **sw)
        want = oracle.mel_spectrogram(p, x)
        assert got.shape == want.shape
        for b in range(4):
            assert _rel(got[b], want[b]) <= RTOL, (sw, b)


def test_host_calls_small_and_chunked_give_the_device_bits(ss):
    """Host-pointer calls take pinned device-mapped staging up to 1 MB of samples (one launch, no copy commands) and the chunked
    two-stream pipeline beyond; both must return exactly what the device-pointer path computes, for every output kind, for
    batch sizes on either side of the limit and when small and large calls alternate on one config (the staging is reused)."""
    import torch

    x = _signal(91, (40, 16000))
    xd = torch.from_numpy(x).cuda()
    want = ss.mfcc_batch(xd, 16000).cpu().numpy()
    wf, we = [t.cpu().numpy() for t in ss.mfe_batch(xd, 16000)]
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    wm = ss.mel_spectrogram(xd, 16000, **kw).cpu().numpy()
    for nb in (1, 40, 2, 16, 17, 15, 40, 3):   # 16 clips = 1 024 000 bytes: the last mapped size; 17 is chunked
        np.testing.assert_array_equal(ss.mfcc_batch(x[:nb], 16000), want[:nb])
        f, e = ss.mfe_batch(x[:nb], 16000)
        np.testing.assert_array_equal(f, wf[:nb])
        np.testing.assert_array_equal(e, we[:nb])
        np.testing.assert_array_equal(ss.mel_spectrogram(x[:nb], 16000, **kw), wm[:nb])
    np.testing.assert_array_equal(ss.mfcc(x[7], 16000), want[7])
    # a strided view (every second clip) is made contiguous by the front; a clip shorter than the others on the same config
    np.testing.assert_array_equal(ss.mfcc_batch(x[::2][:5], 16000), want[::2][:5])
    short = ss.mfcc(x[3, :8000], 16000)
    np.testing.assert_array_equal(short, ss.mfcc_batch(xd[3:4, :8000].contiguous(), 16000).cpu().numpy()[0])


def _on_lab_test_mel_spectrogram_2048_whole_line_tile(ss, oracle, sslib):
    """fft_points = 2048, three builds of one kernel family: eight waves per CU with the CU-wide whole-line tile (a clip's [mel][row]
    block collected in LDS, ss_mel_c1024<tile>; needs at least one clip per CU), eight waves with direct stores, and twelve waves
    (three per SIMD) with direct stores.  Shapes that give a CU one clip, an uneven number of clips, few row pairs per clip (most
    waves never get a unit and only write their share), more clips than tile buffers; every clip must come out bit for bit the
    same from the two eight-wave builds and within f32 rounding from the twelve-wave kernel, and a sample of clips is checked
    against the oracle."""
    import torch

    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    try:
        for n, M, B in ((16000, 128, ncu), (16000, 128, 5 * ncu + 37), (16000, 64, ncu + 19), (3584, 128, 2 * ncu + 1), (7680, 40, ncu + 3),
                        (16000, 100, ncu + 5), (13312, 128, ncu + 7)):
            x = _signal(71, (B, n))
            kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=M, fft_length=2048, high_frequency=8000.0)
            xd = torch.from_numpy(x).cuda()
            sslib.ss_debug_mel_tile(2)  # eight-wave builds: the tile wherever the shape allows it
            got = ss.mel_spectrogram(xd, 16000, **kw)
            name = sslib.ss_last_kernel_name()
            R = got.shape[2]
            tiled = R % 4 == 0 and R <= 32 and M % 8 == 0 and M <= 128
            assert name == (b"ss_mel_c1024<tile>" if tiled else b"ss_mel_c1024"), (name, n, M, B, R)
            for lo in range(0, B, 100):  # fewer clips than CUs: direct stores
                part = ss.mel_spectrogram(xd[lo:lo + 100].contiguous(), 16000, **kw)
                assert sslib.ss_last_kernel_name() == b"ss_mel_c1024"
                assert torch.equal(part, got[lo:lo + 100]), (n, M, B, lo)
            sslib.ss_debug_mel_tile(0)  # eight waves, direct stores, whole batch
            direct = ss.mel_spectrogram(xd, 16000, **kw)
            assert sslib.ss_last_kernel_name() == b"ss_mel_c1024" and torch.equal(direct, got), (n, M, B)
            sslib.ss_debug_mel_tile(3)  # twelve waves
            w12 = ss.mel_spectrogram(xd, 16000, **kw)
            assert sslib.ss_last_kernel_name().startswith(b"ss_mel_c1024<w12"), sslib.ss_last_kernel_name()
            # (a kernel of its own: the compiler fuses some products and sums into FMAs differently, so last-bit differences
            # are expected; 1e-6 of the clip's largest value is f32 rounding of the transform)
            scale = got.abs().amax(dim=(1, 2), keepdim=True)
            assert ((w12 - got).abs() <= 1e-6 * scale).all(), (n, M, B)
            sslib.ss_debug_mel_tile(1)  # automatic: one of the builds above
            auto = ss.mel_spectrogram(xd, 16000, **kw)
            assert torch.equal(auto, w12) or torch.equal(auto, got), (n, M, B, sslib.ss_last_kernel_name())
            pick = sorted({0, 1, B // 2, B - 2, B - 1})
            p = oracle.make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=M, high_frequency=8000.0)
            want = oracle.mel_spectrogram(p, x[pick])
            g = got[pick].cpu().numpy()
            for i in range(len(pick)):
                assert _rel(g[i], want[i]) <= RTOL, (n, M, B, pick[i])
    finally:
        sslib.ss_debug_mel_tile(1)


def test_mel_spectrogram_2048_whole_line_tile(ss, oracle, sslab):
    """Runs on the LAB library (the build selection / fault aids it needs are not in the product library): the front is
    switched to it for the duration."""
    with ss._lib.use_library(sslab):
        _on_lab_test_mel_spectrogram_2048_whole_line_tile(ss, oracle, sslab)


def test_mel_spectrogram_4096_kernel(ss, oracle, sslib):
    """mel_spectrogram at fft_points = 4096 (44.1 kHz, 1024- and 2048-sample chunks, 256 / 128 / 100 mels): one row per wave on
    the 4096-point FFT mapping; partial last chunks, clips shorter than a window."""
    import torch

    sr = 44100
    for n, hop, M in ((44100, 1024, 256), (44100, 2048, 128), (30000, 1024, 100), (3000, 1024, 64)):
        x = _signal(45, (3, n))
        kw = dict(frame_length=hop / sr, frame_stride=hop / sr, num_filters=M, fft_length=4096)
        got = ss.mel_spectrogram(torch.from_numpy(x).cuda(), sr, **kw).cpu().numpy()
        assert sslib.ss_last_kernel_name() == b"ss_mel_c2048", (sslib.ss_last_kernel_name(), n, hop, M)
        p = oracle.make_params(sample_rate=sr, fft_points=4096, frame_length=hop / sr, frame_stride=hop / sr, num_filters=M)
        want = oracle.mel_spectrogram(p, x)
        assert got.shape == want.shape
        for b in range(3):
            assert _rel(got[b], want[b]) <= RTOL, (n, hop, M, b)


def test_cfg5_highres(ss, oracle):
    import torch

    x = _signal(5, (8, 44100))
    got = ss.mfcc_batch(torch.from_numpy(x).cuda(), 44100, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                        num_cepstral=40, num_filters=256, fft_length=4096).cpu().numpy()
    assert got.shape == (8, 39, 40)
    p = oracle.make_params(**CFG5)
    for b in range(8):
        assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL


def test_stft_stage(ss, oracle, sslib):
    import torch
    from speechsauce_amd import _lib

    x = _signal(7, (2, 5000))
    cfg = _cfg(ss, **CFG3)
    R, Rreal = cfg.stft_rows(5000)
    xd = torch.from_numpy(x).cuda()
    out = torch.full((2, R, 1025, 2), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_stft_device(cfg.handle, xd.data_ptr(), 2, 5000, 5000, out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = oracle.stft(oracle.make_params(**CFG3), x)
    g = got[..., 0] + 1j * got[..., 1]
    assert np.abs(g - want).max() <= 1e-5 * np.abs(want).max()
    assert np.all(g[:, Rreal:] == 0)
    assert sslib.ss_last_kernel_name() == b"ss_mel_c1024<w12,stft>"  # input + output inside the Infinity Cache: the twelve-wave build
    # an odd number of rows (the last row pair is half empty) and a batch whose units span clip boundaries
    x = _signal(71, (37, 16100))
    R, Rreal = cfg.stft_rows(16100)
    xd = torch.from_numpy(x).cuda()
    out = torch.full((37, R, 1025, 2), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_stft_device(cfg.handle, xd.data_ptr(), 37, 16100, 16100, out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    g = got[..., 0] + 1j * got[..., 1]
    for b in (0, 18, 36):
        want = oracle.stft(oracle.make_params(**CFG3), x[b:b + 1])[0]
        assert np.abs(g[b] - want).max() <= 1e-5 * np.abs(want).max()
    assert np.all(g[:, Rreal:] == 0)
    # the 1024-clip batch tools/stage_rate.py times (268.7 MB of output: beyond the Infinity Cache, the eight-wave build that asks
    # for the next unit's samples ahead of its stores): that build, by name, against the oracle on sampled clips -- and the
    # same clips through the twelve-wave build (a small batch) agree to f32 rounding of the transform
    xb = _signal(72, (1024, 16000))
    xbd = torch.from_numpy(xb).cuda()
    outb = torch.empty((1024, 32, 1025, 2), dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_stft_device(cfg.handle, xbd.data_ptr(), 1024, 16000, 16000, outb.data_ptr(), None))
    torch.cuda.synchronize()
    assert sslib.ss_last_kernel_name() == b"ss_mel_c1024<stft>"
    pick = [0, 300, 1023]
    sub = torch.empty((3, 32, 1025, 2), dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_stft_device(cfg.handle, xbd[pick].contiguous().data_ptr(), 3, 16000, 16000, sub.data_ptr(), None))
    torch.cuda.synchronize()
    assert sslib.ss_last_kernel_name() == b"ss_mel_c1024<w12,stft>"
    want = oracle.stft(oracle.make_params(**CFG3), xb[pick])
    for i, b in enumerate(pick):
        gb = outb[b].cpu().numpy()
        gb = gb[..., 0] + 1j * gb[..., 1]
        assert np.abs(gb - want[i]).max() <= 1e-5 * np.abs(want[i]).max(), b
        assert np.all(gb[29:] == 0)
        assert (sub[i] - outb[b]).abs().max().item() <= 1e-6 * outb[b].abs().max().item()


@pytest.mark.parametrize("nfft,sr,hop,kernel", [(512, 16000, 256, b"ss_mel_c256<stft>"), (512, 16000, 160, b"ss_mel_c256<stft>"),
                                                 (1024, 16000, 512, b"ss_mel_c512<stft>"), (1024, 22050, 256, b"ss_mel_c512<stft>"),
                                                 (4096, 44100, 1024, b"ss_mel_c2048<stft>")])
def test_stft_stage_other_sizes(ss, oracle, sslib, nfft, sr, hop, kernel):
    """stft2's output (functions.rs:86-123) from the stft builds of the 512-, 1024- and 4096-point mel kernels: all bins of
    every row, trailing n_pad rows zero, row counts that do not fill the last wave, partial last chunks."""
    import torch
    from speechsauce_amd import _lib

    kw = dict(sample_rate=sr, fft_points=nfft, frame_length=hop / sr, frame_stride=hop / sr, num_filters=128)  # 512 points: more filters than the mel stage takes -> a table block for the stft build only
    cfg = _cfg(ss, **kw)
    n = 3 * nfft + 6 * hop + 88
    x = _signal(73, (5, n))
    R, Rreal = cfg.stft_rows(n)
    F = nfft // 2 + 1
    out = torch.full((5, R, F, 2), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_stft_device(cfg.handle, torch.from_numpy(x).cuda().data_ptr(), 5, n, n, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert sslib.ss_last_kernel_name() == kernel, sslib.ss_last_kernel_name()
    got = out.cpu().numpy()
    g = got[..., 0] + 1j * got[..., 1]
    want = oracle.stft(oracle.make_params(**kw), x)
    assert g.shape == want.shape
    assert np.abs(g - want).max() <= 1e-5 * np.abs(want).max()
    assert np.all(g[:, Rreal:] == 0)


def test_preemphasis(ss, oracle):
    x = _signal(8, 4001)
    for shift, cof in [(1, 0.98), (3, 0.5), (4001, 0.9)]:
        got = ss.preemphasis(x, shift=shift, cof=cof)
        assert got.shape == x.shape
        assert _rel(got, oracle.preemphasis(x, shift, cof)) <= 1e-6


def test_switches(ss, oracle):
    x = _signal(9, 16000)
    for sw in [dict(spectrum_exponent=2), dict(dct_norm="ortho"), dict(mfcc_window="hann"), dict(mfcc_window="vorbis"),
               dict(preemph_coef=0.97), dict(preemph_coef=0.9, preemph_shift=2), dict(dct2_gain=1.0)]:
        got = ss.mfcc(x, 16000, **sw)
        want = oracle.mfcc(oracle.make_params(**CFG1, **sw), x)
        assert _rel(got, want) <= RTOL, sw
    got = ss.mfcc(x, 16000, dc_elimination=False)
    want = oracle.mfcc(oracle.make_params(**CFG1, dc_elimination=False), x)
    assert _rel(got, want) <= RTOL


def test_front_end_fast_path(ss, oracle, sslib):
    """Frame window and fused pre-emphasis (the north-star's "Hann window + pre-emphasis framing loop") are served by
    builds of the 512-point kernel, for mfcc and mfe, over batches whose quads span clip boundaries."""
    import torch

    x = _signal(14, (21, 16000))
    xd = torch.from_numpy(x).cuda()
    for sw, tag in [(dict(mfcc_window="hann"), b"win"), (dict(preemph_coef=0.97), b"pre"),
                    (dict(mfcc_window="vorbis", preemph_coef=0.9, preemph_shift=3), b"win,pre")]:
        got = ss.mfcc_batch(xd, 16000, **sw).cpu().numpy()
        name = sslib.ss_last_kernel_name()
        assert name.startswith(b"ss_mfcc_c256<") and tag in name, name
        p = oracle.make_params(**CFG1, **sw)
        for b in (0, 10, 20):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (sw, b)
        feat, en = ss.mfe_batch(xd, 16000, **sw)
        assert b"mfe" in sslib.ss_last_kernel_name() and tag in sslib.ss_last_kernel_name()
        wf, we = oracle.mfe(p, x[20])
        assert _rel(feat[20].cpu().numpy(), wf) <= RTOL and _rel(en[20].cpu().numpy(), we) <= RTOL


def test_25ms_frames(ss, oracle, sslib):
    """The classic 25 ms / 10 ms speech front end (400-sample frames in a 512-point FFT): 13-input build of the kernel."""
    import torch

    x = _signal(15, (33, 16000))
    got = ss.mfcc_batch(torch.from_numpy(x).cuda(), 16000, frame_length=0.025).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256<13")
    p = oracle.make_params(**dict(CFG1, frame_length=0.025))
    assert got.shape == (33, oracle.num_frames(p, 16000), 13)
    for b in (0, 16, 32):
        assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL
    got26 = ss.mfcc_batch(torch.from_numpy(x).cuda(), 16000, frame_length=0.025, num_filters=26).cpu().numpy()
    p26 = oracle.make_params(**dict(CFG1, frame_length=0.025, num_filters=26))
    assert _rel(got26[5], oracle.mfcc(p26, x[5])) <= RTOL


def test_literal_framing_known_answer(ss, oracle):
    """processing.rs:110-120 as written copies nothing for > 2 frames: output is signal-independent."""
    x = _signal(10, 16000)
    got = ss.mfcc(x, 16000, framing="literal")
    assert np.allclose(got[:, 0], np.log(np.float32(1.1920929e-7)), rtol=1e-6)  # ln(EPS) = -15.942385
    assert np.abs(got[:, 1:]).max() < 1e-4  # DCT of a constant row
    want = oracle.mfcc(oracle.make_params(**CFG1, framing="literal"), x)
    assert np.abs(got - want).max() <= 1e-4


def test_edge_signals(ss, oracle):
    p = oracle.make_params(**CFG1)
    zero = np.zeros(16000, np.float32)
    got = ss.mfcc(zero, 16000)
    assert np.abs(got - oracle.mfcc(p, zero)).max() <= 1e-4
    sq = np.where(np.arange(16000) % 64 < 32, 1.0, -1.0).astype(np.float32)
    assert _rel(ss.mfcc(sq, 16000), oracle.mfcc(p, sq)) <= RTOL
    imp = np.zeros(16000, np.float32)
    imp[::160] = 1.0  # one unit impulse per frame hop
    assert _rel(ss.mfcc(imp, 16000), oracle.mfcc(p, imp)) <= RTOL


def test_a_nan_sample_stays_in_its_own_frames(ss):
    """One NaN (and one +inf) sample in one clip of a batch: exactly the frames / rows whose window covers it come out non-finite
    (as in the reference, whose arithmetic propagates them), every other frame of that clip and every other clip keeps its bits --
    frames share a wave, an exchange region and a reduction tree, and none of that may leak."""
    import torch

    for bad in (float("nan"), float("inf")):
        x = torch.from_numpy(_signal(93, (6, 16000))).cuda()
        clean = ss.mfcc_batch(x, 16000)
        xb = x.clone()
        xb[2, 5000] = bad
        got = ss.mfcc_batch(xb, 16000)
        t = torch.arange(98, device="cuda")
        hit = (t * 160 <= 5000) & (5000 < t * 160 + 320)      # frames whose 320 samples cover sample 5000
        assert int(hit.sum()) == 2
        assert not torch.isfinite(got[2, hit]).all(dim=1).any()
        keep = torch.ones((6, 98), dtype=torch.bool, device="cuda")
        keep[2, hit] = False
        assert torch.equal(got[keep], clean[keep])
        # mel spectrogram, 2048-point windows with hop 512: row r covers samples [512 r, 512 r + 2048)
        kw3 = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
        mclean, mgot = ss.mel_spectrogram(x, 16000, **kw3), ss.mel_spectrogram(xb, 16000, **kw3)
        r = torch.arange(32, device="cuda")
        mhit = (r * 512 <= 5000) & (5000 < r * 512 + 2048) & (r < 29)
        mkeep = torch.ones((6, 32), dtype=torch.bool, device="cuda")
        mkeep[2, mhit] = False
        assert torch.equal(mgot.permute(0, 2, 1)[mkeep], mclean.permute(0, 2, 1)[mkeep])
        assert not torch.isfinite(mgot.permute(0, 2, 1)[2, mhit]).all(dim=1).any()
        x5 = torch.from_numpy(_signal(94, (3, 44100))).cuda()
        kw5 = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096, high_frequency=22050.0)
        c5 = ss.mfcc_batch(x5, 44100, **kw5)
        x5b = x5.clone()
        x5b[1, 20000] = bad
        g5 = ss.mfcc_batch(x5b, 44100, **kw5)
        t5 = torch.arange(39, device="cuda")
        h5 = (t5 * 1024 <= 20000) & (20000 < t5 * 1024 + 4096)
        k5 = torch.ones((3, 39), dtype=torch.bool, device="cuda")
        k5[1, h5] = False
        assert int(h5.sum()) == 4 and torch.equal(g5[k5], c5[k5]) and not torch.isfinite(g5[1, h5]).all(dim=1).any()


@pytest.mark.parametrize("amp", [1e-6, 1e-3, 1.0, 32768.0])
def test_amplitude_range_on_the_bench_kernels(ss, oracle, amp):
    """The amplitudes a front end meets -- a near-silent recording (1e-6), quiet speech, full scale, and floats that still carry the
    int16 scale (32768) -- through the three bench configurations against the oracle.  The kernels take `ln` of values pre-scaled
    by 2^32 and `v_sqrt_f32` / `v_log_f32` without denormal handling: both are exact in this range (|X|^2 leaves the normal f32
    range only below ~1e-19 / above ~1e19 of amplitude, where the reference's own f32 arithmetic underflows / overflows too)."""
    import torch

    x1 = (_signal(91, (3, 16000)) * (amp / 0.1)).astype(np.float32)
    got = ss.mfcc_batch(torch.from_numpy(x1).cuda(), 16000).cpu().numpy()
    p = oracle.make_params(**CFG1)
    for b in range(3):
        assert _rel(got[b], oracle.mfcc(p, x1[b])) <= RTOL, ("cfg1", amp, b)
    kw3 = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    mel = ss.mel_spectrogram(torch.from_numpy(x1).cuda(), 16000, **kw3).cpu().numpy()
    want = oracle.mel_spectrogram(oracle.make_params(**CFG3), x1)
    for b in range(3):
        assert _rel(mel[b], want[b]) <= RTOL, ("cfg3", amp, b)
    x5 = (_signal(92, (2, 44100)) * (amp / 0.1)).astype(np.float32)
    kw5 = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096, high_frequency=22050.0)
    got5 = ss.mfcc_batch(torch.from_numpy(x5).cuda(), 44100, **kw5).cpu().numpy()
    p5 = oracle.make_params(**CFG5)
    for b in range(2):
        assert _rel(got5[b], oracle.mfcc(p5, x5[b])) <= RTOL, ("cfg5", amp, b)


@pytest.mark.parametrize("sr,nfft,flen,hop,M,C", [(8000, 256, 160, 80, 40, 13), (16000, 512, 400, 160, 80, 13), (22050, 1024, 1024, 256, 64, 20),
                                                  (44100, 2048, 2048, 512, 128, 20), (44100, 4096, 4096, 1024, 256, 40)])
def test_edge_signals_every_frame_kernel(ss, oracle, sslib, sr, nfft, flen, hop, M, C):
    """All-zero clip (zero_handling everywhere, functions.rs:66-71), full-scale square wave and an impulse train through each
    frame-path kernel: MFCC and mfe."""
    kw = dict(frame_length=flen / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M)
    n = flen + 11 * hop + 2
    zero = np.zeros(n, np.float32)
    sq = np.where(np.arange(n) % 64 < 32, 1.0, -1.0).astype(np.float32)
    imp = np.zeros(n, np.float32)
    imp[::hop] = 1.0
    mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
    # (the 256-point kernel transforms two frames at once; bins that cancel exactly -- the square wave has exact zeros in most
    # bands -- trip its tiny-bin check and the oct is run again one frame per transform: no signal class is left out)
    for x in (zero, sq, imp):
        got = ss.mfcc(x, sr, **kw)
        assert not sslib.ss_last_kernel_name().startswith(b"ss_front_generic")
        want = oracle.mfcc(p, x)
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
        feat, en = ss.mfe(x, sr, **mkw)
        wf, we = oracle.mfe(p, x)
        assert _rel(feat, wf) <= RTOL and _rel(en, we) <= RTOL
    feat, en = ss.mfe(zero, sr, **mkw)
    assert np.all(feat == np.float32(1.1920929e-7)) and np.all(en == np.float32(1.1920929e-7))


@pytest.mark.parametrize("sr,nfft,hop,M", [(16000, 512, 256, 40), (16000, 1024, 512, 80), (16000, 2048, 512, 128), (44100, 4096, 1024, 256)])
def test_edge_signals_every_mel_kernel(ss, oracle, sslib, sr, nfft, hop, M):
    """All-zero clip (no zero handling on this path: exact zeros), full-scale square wave and an impulse train through each
    mel-spectrogram kernel."""
    kw = dict(frame_length=hop / sr, frame_stride=hop / sr, num_filters=M, fft_length=nfft)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=hop / sr, frame_stride=hop / sr, num_filters=M)
    n = nfft + 12 * hop
    sq = np.where(np.arange(n) % 64 < 32, 1.0, -1.0).astype(np.float32)
    imp = np.zeros(n, np.float32)
    imp[::hop] = 1.0
    got = ss.mel_spectrogram(np.zeros(n, np.float32), sr, **kw)
    assert sslib.ss_last_kernel_name().startswith(b"ss_mel_c") and np.all(got == 0.0)
    for x in (sq, imp):
        got = ss.mel_spectrogram(x, sr, **kw)
        want = oracle.mel_spectrogram(p, x)
        assert got.shape == want.shape and _rel(got, want) <= RTOL


@pytest.mark.parametrize("sr,nfft,flen,step,M,C,kernel", [
    (16000, 512, 320, 161, 40, 13, b"ss_mfcc_c256<"), (22050, 512, 441, 221, 40, 13, b"ss_mfcc_c256<"), (16000, 512, 399, 160, 80, 13, b"ss_mfcc_c256w<"),
    (22050, 1024, 883, 221, 64, 20, b"ss_mfcc_c512"), (44100, 2048, 1765, 441, 128, 20, b"ss_mfcc_c1024"), (44100, 4096, 4095, 1023, 256, 40, b"ss_mfcc_c2048"),
    (44100, 4096, 4096, 1025, 128, 20, b"ss_mfcc_c2048")])
def test_odd_lengths_hops_and_offsets(ss, oracle, sslib, sr, nfft, flen, step, M, C, kernel):
    """Odd frame lengths (22.05 kHz x 20 ms = 441 samples), odd hops, odd leading dimensions and odd base offsets reach the
    dedicated frame kernels: sample pairs load at dword alignment, an odd frame ends in a half pair.  The last frame of the
    last clip ends exactly at the end of the buffer (no pair may read past it)."""
    import torch

    T = 7
    n = flen + (T - 1) * step  # the last frame ends with the clip
    x = _signal(51, (3, n + 1))
    xd = torch.from_numpy(x).cuda()[:, 1:]  # odd base offset; the leading dimension n + 1 has the other parity of n
    kw = dict(frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M)
    # (the headline kernel's window builds exist for the default frame shape only)
    for sw in ({},) if kernel == b"ss_mfcc_c256<" else ({}, dict(mfcc_window="hann")):
        q = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(kernel), sslib.ss_last_kernel_name()
        assert got.shape[1] == oracle.num_frames(q, n)
        for b in range(3):
            assert _rel(got[b], oracle.mfcc(q, x[b, 1:])) <= RTOL, (sw, b)
    mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
    feat, en = ss.mfe_batch(xd, sr, **mkw)
    wf, we = oracle.mfe(p, x[2, 1:])
    assert _rel(feat[2].cpu().numpy(), wf) <= RTOL and _rel(en[2].cpu().numpy(), we) <= RTOL
    # a contiguous exact-size tensor: the last pair of the last frame of the last clip is the end of the allocation
    xe = torch.from_numpy(np.ascontiguousarray(x[:, 1:])).cuda()
    got = ss.mfcc_batch(xe, sr, **kw).cpu().numpy()
    assert _rel(got[2], oracle.mfcc(p, x[2, 1:])) <= RTOL


@pytest.mark.parametrize("sr,nfft,hop,M,kernel", [(22050, 512, 221, 40, b"ss_mel_c256"), (22050, 1024, 441, 80, b"ss_mel_c512"),
                                                  (22050, 2048, 441, 128, b"ss_mel_c1024"), (44100, 4096, 1103, 128, b"ss_mel_c2048")])
def test_mel_odd_hops_lengths_and_offsets(ss, oracle, sslib, sr, nfft, hop, M, kernel):
    """Odd chunk sizes (22.05 kHz x 20 ms = 441), odd clip lengths, odd leading dimensions and odd base offsets on the
    mel-spectrogram kernels: pairs load at dword alignment, pairs that straddle a clip edge are bounded per sample."""
    import torch

    kw = dict(frame_length=hop / sr, frame_stride=hop / sr, num_filters=M, fft_length=nfft)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=hop / sr, frame_stride=hop / sr, num_filters=M)
    for n in (nfft + 9 * hop + 1, nfft + 7 * hop + 2, 3 * hop + 1):
        x = _signal(53, (4, n + 1))
        xd = torch.from_numpy(x).cuda()[:, 1:]
        got = ss.mel_spectrogram(xd, sr, **kw).cpu().numpy()
        assert sslib.ss_last_kernel_name() == kernel, sslib.ss_last_kernel_name()
        want = oracle.mel_spectrogram(p, x[:, 1:])
        assert got.shape == want.shape
        for b in range(4):
            assert _rel(got[b], want[b]) <= RTOL, (n, b)


@pytest.mark.parametrize("sr,nfft,flen,step,M,C", [(16000, 400, 400, 160, 40, 13), (16000, 400, 320, 160, 80, 13), (22050, 441, 441, 220, 40, 13),
                                                  (44100, 1000, 882, 441, 64, 20), (44100, 1323, 1323, 441, 128, 20), (8000, 100, 80, 40, 20, 12),
                                                  (8000, 48, 48, 16, 10, 5), (16000, 1365, 1200, 400, 40, 13), (44100, 1764, 1764, 441, 128, 20),
                                                  (44100, 2730, 2205, 1024, 128, 13)])
def test_fft_lengths_that_are_not_powers_of_two(ss, oracle, sslib, sr, nfft, flen, step, M, C):
    """The reference takes any fft_points (rustfft); here every length that is not a power of two runs the chirp-z build of the
    generic kernel: n_fft = 400 (25 ms at 16 kHz), 441, 1000, odd lengths, tiny ones; MFCC, mfe, windows, centred frames."""
    import torch

    x = _signal(61, (5, flen + 9 * step + 3))
    xd = torch.from_numpy(x).cuda()
    for sw in ({}, dict(mfcc_window="hann", spectrum_exponent=2, dct_norm="ortho"), dict(framing="center", preemph_coef=0.97)):
        kw = dict(frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
        p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert b"chirpz" in sslib.ss_last_kernel_name(), sslib.ss_last_kernel_name()
        assert got.shape == (5, oracle.num_frames(p, x.shape[1]), C)
        for b in (0, 4):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        msw = {k: v for k, v in sw.items() if k != "dct_norm"}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **msw)
        wf, we = oracle.mfe(p, x[4])
        assert _rel(feat[4].cpu().numpy(), wf) <= RTOL and _rel(en[4].cpu().numpy(), we) <= RTOL, sw


@pytest.mark.parametrize("sr,nfft,hop,M", [(16000, 400, 200, 40), (16000, 400, 160, 80), (22050, 441, 147, 40), (44100, 1000, 500, 64), (8000, 101, 50, 20)])
def test_mel_spectrogram_fft_lengths_that_are_not_powers_of_two(ss, oracle, sslib, sr, nfft, hop, M):
    """mel_spectrogram / stft2 at fft_points that are not powers of two (chirp-z build of the generic kernel)."""
    import torch
    from speechsauce_amd import _lib

    kw = dict(frame_length=hop / sr, frame_stride=hop / sr, num_filters=M, fft_length=nfft)
    pk = dict(sample_rate=sr, fft_points=nfft, frame_length=hop / sr, frame_stride=hop / sr, num_filters=M)
    p = oracle.make_params(**pk)
    n = 3 * nfft + 7 * hop + 5
    x = _signal(63, (3, n))
    got = ss.mel_spectrogram(torch.from_numpy(x).cuda(), sr, **kw).cpu().numpy()
    assert b"chirpz" in sslib.ss_last_kernel_name()
    want = oracle.mel_spectrogram(p, x)
    assert got.shape == want.shape and _rel(got, want) <= RTOL
    cfg = _cfg(ss, **pk)
    R, Rreal = cfg.stft_rows(n)
    F = nfft // 2 + 1
    out = torch.full((3, R, F, 2), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_stft_device(cfg.handle, torch.from_numpy(x).cuda().data_ptr(), 3, n, n, out.data_ptr(), None))
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    g = g[..., 0] + 1j * g[..., 1]
    ws = oracle.stft(p, x)
    assert g.shape == ws.shape and np.abs(g - ws).max() <= 2e-5 * np.abs(ws).max()
    assert np.all(g[:, Rreal:] == 0)


def test_fft_length_limits(ss):
    from speechsauce_amd import SpeechSauceError

    x = _signal(64, 16000)
    for nfft in (2731, 3000, 16384, 15):
        with pytest.raises(SpeechSauceError):
            ss.mfcc(x, 16000, fft_length=nfft, frame_length=10 / 16000 if nfft == 15 else 0.02)


def test_fft_8192(ss, oracle, sslib):
    """fft_points = 8192 (the generic kernel with one workgroup per frame): 48 kHz, full-length frames, MFCC and mel-spectrogram."""
    import torch

    sr = 48000
    x = _signal(65, (2, 8192 + 5 * 2048 + 7))
    kw = dict(frame_length=8192 / sr, frame_stride=2048 / sr, num_cepstral=20, num_filters=128, fft_length=8192)
    p = oracle.make_params(sample_rate=sr, fft_points=8192, frame_length=8192 / sr, frame_stride=2048 / sr, num_cepstral=20, num_filters=128)
    got = ss.mfcc_batch(torch.from_numpy(x).cuda(), sr, **kw).cpu().numpy()
    assert sslib.ss_last_kernel_name() == b"ss_front_generic<12>"
    for b in range(2):
        assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL
    mk = dict(frame_length=2048 / sr, frame_stride=2048 / sr, num_filters=128, fft_length=8192)
    pm = oracle.make_params(sample_rate=sr, fft_points=8192, frame_length=2048 / sr, frame_stride=2048 / sr, num_filters=128)
    gm = ss.mel_spectrogram(x, sr, **mk)
    wm = oracle.mel_spectrogram(pm, x)
    assert gm.shape == wm.shape and _rel(gm, wm) <= RTOL


def test_256_kernel_pair_guard(ss, oracle, sslib):
    """The two-frames-per-transform kernel must not let a loud frame's rounding noise into its silent partner: zero-padded
    clips (all-zero frames next to speech: exact f32::EPSILON energies, as in the reference), digital silence followed by a
    loud onset, and a 70 dB level step."""
    import torch

    sr, n = 8000, 8000
    rng = np.random.default_rng(77)
    x = np.zeros((6, n), np.float32)
    x[0, :3000] = rng.standard_normal(3000) * 0.3                      # speech, then zero padding
    x[1, 5000:] = rng.standard_normal(3000) * 0.3                      # silence, then an onset
    x[2] = rng.standard_normal(n) * 1e-4
    x[2, 4000:] = rng.standard_normal(4000) * 0.4                      # 72 dB step
    x[3, 1234:1300] = 0.9                                              # a click in silence
    x[4] = rng.standard_normal(n) * 0.1                                # ordinary clip (single-pass octs)
    x[5, ::2] = 1e-3                                                   # quiet everywhere
    p = oracle.make_params(sample_rate=sr, fft_points=256)
    xd = torch.from_numpy(x).cuda()
    got = ss.mfcc_batch(xd, sr, fft_length=256).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256x2<")
    feat, en = ss.mfe_batch(xd, sr, fft_length=256)
    feat, en = feat.cpu().numpy(), en.cpu().numpy()
    eps = np.float32(1.1920929e-7)
    for b in range(6):
        want = oracle.mfcc(p, x[b])
        assert np.abs(got[b] - want).max() <= 1e-4 * np.abs(want).max(), b
        wf, we = oracle.mfe(p, x[b])
        # per frame: each frame's mel energies against that frame's own scale
        for t in range(wf.shape[0]):
            assert np.abs(feat[b, t] - wf[t]).max() <= 1e-4 * max(np.abs(wf[t]).max(), eps), (b, t)
        assert _rel(en[b], we) <= RTOL
    assert np.all(feat[0, 40:] == eps) and np.all(en[0, 40:] == eps)   # frames wholly inside the zero padding


def test_ragged_lengths_and_errors(ss, oracle):
    from speechsauce_amd import SpeechSauceError

    p = oracle.make_params(**CFG1)
    for n in (480, 481, 639, 640, 641, 8191, 16001):
        x = _signal(n, n)
        got = ss.mfcc(x, 16000)
        want = oracle.mfcc(p, x)
        assert got.shape == want.shape
        assert _rel(got, want) <= RTOL
    for n in (10, 319, 320, 479):  # fewer samples than a frame, or zero frames: the reference panics
        with pytest.raises(SpeechSauceError) as e:
            ss.mfcc(_signal(1, n), 16000)
        assert e.value.status == 1
    with pytest.raises(SpeechSauceError):  # default config: fft 512 < 2*320 -> STFT path underflows (functions.rs:136)
        ss.mel_spectrogram(_signal(1, 16000), 16000)
    with pytest.raises(TypeError):
        ss.mfcc(np.zeros(16000, np.float64), 16000)


def test_short_clip_batches(ss, oracle):
    """Batches of very short clips: a quad of the 512-point kernel then spans several clips (frame -> (clip, t) wraps),
    with and without the [0,0]-scaled DCT row that depends on t == 0 (feature.rs:126-146)."""
    import torch

    for n in (480, 640, 800, 960, 1120, 1600):  # 1, 2, 3, 4, 5, 8 frames per clip
        x = _signal(40 + n, (37, n))
        for dc in (True, False):
            got = ss.mfcc_batch(torch.from_numpy(x).cuda(), 16000, dc_elimination=dc).cpu().numpy()
            p = oracle.make_params(**dict(CFG1, dc_elimination=int(dc)))
            for b in (0, 1, 17, 36):
                assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (n, dc, b)


def test_reference_test_shapes(ss):
    """speechsauce/src/lib.rs:93-134: mfcc (6248, 13), mfe (6248, 40)/(6248,), no NaN, for 1e6 samples."""
    x = _signal(11, 1_000_000)
    out = ss.mfcc(x, 16000)
    assert out.shape == (6248, 13) and not np.isnan(out).any()
    feat, en = ss.mfe(x, 16000)
    assert feat.shape == (6248, 40) and en.shape == (6248,)


def test_strided_batch_ld(ss, oracle, sslib):
    import torch
    from speechsauce_amd import _lib

    x = _signal(12, (5, 20000))
    xd = torch.from_numpy(x).cuda()
    cfg = _cfg(ss, **CFG1)
    T = cfg.num_frames(16000)
    out = torch.empty((5, T, 13), dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_mfcc_batch_device(cfg.handle, xd.data_ptr(), 5, 16000, 20000, out.data_ptr(), None))
    torch.cuda.synchronize()
    p = oracle.make_params(**CFG1)
    for b in range(5):
        assert _rel(out[b].cpu().numpy(), oracle.mfcc(p, x[b, :16000])) <= RTOL


@pytest.mark.parametrize("ld", [(1 << 30) - 1000, (1 << 30) + 20000, (1 << 32) + 2])
def test_huge_row_stride_is_addressed_in_64_bits(ss, oracle, sslib, ld):
    """A row stride of >= 2^30 floats (>= 4 GB between clips): the headline kernel addresses a lane's frame by a 32-bit byte offset
    from its quad's base and must hand such batches to a kernel that forms 64-bit addresses instead of wrapping (round-4
    advisor finding).  Three clips at that stride, compared bit for bit with the same clips at a small stride."""
    import torch
    from speechsauce_amd import _lib

    B, n = 3, 16000
    x = _signal(21, (B, n))
    big = torch.zeros(((B - 1) * ld + n,), dtype=torch.float32, device="cuda")
    for b in range(B):
        big[b * ld: b * ld + n] = torch.from_numpy(x[b]).cuda()
    cfg = _cfg(ss, **CFG1)
    T = cfg.num_frames(n)
    out = torch.empty((B, T, 13), dtype=torch.float32, device="cuda")
    _lib.check(sslib.ss_mfcc_batch_device(cfg.handle, big.data_ptr(), B, n, ld, out.data_ptr(), None))
    torch.cuda.synchronize()
    kernel_big = sslib.ss_last_kernel_name()
    del big
    small = torch.from_numpy(x).cuda()
    want = torch.empty_like(out)
    _lib.check(sslib.ss_mfcc_batch_device(cfg.handle, small.data_ptr(), B, n, n, want.data_ptr(), None))
    torch.cuda.synchronize()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256<")
    p = oracle.make_params(**CFG1)
    for b in range(B):
        assert _rel(out[b].cpu().numpy(), oracle.mfcc(p, x[b])) <= RTOL, (b, kernel_big)
    if kernel_big.startswith(b"ss_mfcc_c256<"):  # strides the 32-bit offsets still cover stay on the headline kernel: same bits
        assert torch.equal(out, want)


# ---------------------------------------------------------------------------------------------------------
# committed golden fixtures (tests/golden/golden_v1.npz, generated by the numpy restatement)
# ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["cfg1", "cfg5"])
def test_golden_fixtures_mfcc(ss, name):
    from common import CONFIGS, N_SAMPLES, golden_signals, load_golden

    g = load_golden()
    kw = dict(CONFIGS[name])
    sr = kw.pop("sample_rate")
    pykw = dict(frame_length=kw.get("frame_length", 0.02), frame_stride=kw.get("frame_stride", 0.01),
                num_cepstral=kw.get("num_cepstral", 13), num_filters=kw.get("num_filters", 40),
                fft_length=kw.get("fft_points", 512), high_frequency=kw.get("high_frequency"))
    for sname, x in golden_signals(N_SAMPLES[name], sr).items():
        got = ss.mfcc(x, sr, **pykw)
        assert _rel(got, g[f"{name}/{sname}/mfcc"]) <= RTOL, sname
        feat, en = ss.mfe(x, sr, frame_length=pykw["frame_length"], frame_stride=pykw["frame_stride"],
                          num_filters=pykw["num_filters"], fft_length=pykw["fft_length"], high_frequency=pykw["high_frequency"])
        assert _rel(en, g[f"{name}/{sname}/energy"]) <= 1e-5
        assert _rel(feat[[0, feat.shape[0] // 2, -1]], g[f"{name}/{sname}/feat_rows"]) <= 1e-5


def test_golden_fixtures_mel(ss):
    from common import golden_signals, load_golden

    g = load_golden()
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    for sname, x in golden_signals(16000, 16000).items():
        got = ss.mel_spectrogram(x, 16000, **kw)
        assert _rel(got, g[f"cfg3/{sname}/mel"]) <= RTOL, sname


def test_golden_switches(ss):
    from common import golden_signals, load_golden

    g = load_golden()
    x = golden_signals(16000, 16000)["noise"]
    for tag, sw in {"pow2": dict(spectrum_exponent=2), "ortho": dict(dct_norm="ortho"), "hann": dict(mfcc_window="hann"),
                    "preemph": dict(preemph_coef=0.97), "nodc": dict(dc_elimination=False)}.items():
        assert _rel(ss.mfcc(x, 16000, **sw), g[f"switch/{tag}"]) <= RTOL, tag
    assert np.abs(ss.mfcc(x, 16000, framing="literal") - g["switch/literal"]).max() <= 1e-4


# ---------------------------------------------------------------------------------------------------------
# size-independent properties at BASELINE.json's full batch sizes
# ---------------------------------------------------------------------------------------------------------

def test_cfg2_full_batch_properties(ss, sslib):
    import torch
    from speechsauce_amd import _lib

    x = torch.from_numpy(_signal(21, (1024, 16000))).cuda()
    a = ss.mfcc_batch(x, 16000)
    b = ss.mfcc_batch(x, 16000)
    assert torch.equal(a, b)                                   # deterministic: no atomics, fixed reduction orders
    perm = torch.randperm(1024, device="cuda")
    assert torch.equal(ss.mfcc_batch(x[perm].contiguous(), 16000), a[perm])   # clips are independent units
    one = ss.mfcc_batch(x[517:518].contiguous(), 16000)
    assert torch.equal(one[0], a[517])                         # batch == single clip, bit for bit
    assert torch.isfinite(a).all()
    # |X|/N and the mel energies are homogeneous of degree 1: mfe(4x) = 4 mfe(x) exactly (power of two)
    f1, e1 = ss.mfe_batch(x[:64].contiguous(), 16000)
    f4, e4 = ss.mfe_batch((4.0 * x[:64]).contiguous(), 16000)
    assert torch.equal(f4, 4.0 * f1) and torch.equal(e4, 4.0 * e1)
    # Parseval on the squared spectrum: sum_k w_k |X[k]|^2 / N = sum_n x_n^2 over each (zero-padded) frame
    cfg = _cfg(ss, sample_rate=16000, spectrum_exponent=2)
    T = cfg.num_frames(16000)
    P = torch.empty((8, T, 257), dtype=torch.float32, device="cuda")
    xs = x[:8].contiguous()
    _lib.check(sslib.ss_power_spectrum_batch_device(cfg.handle, xs.data_ptr(), 8, 16000, 16000, P.data_ptr(), None))
    w = torch.full((257,), 2.0, device="cuda")
    w[0] = w[256] = 1.0
    lhs = (P * w).sum(-1)
    frames = xs.unfold(1, 320, 160)[:, :T]
    rhs = (frames.double() ** 2).sum(-1)
    assert ((lhs.double() - rhs).abs() / rhs).max().item() < 1e-5


def test_cfg3_full_batch_properties(ss, oracle):
    import torch

    xh = _signal(22, (1024, 16000))
    x = torch.from_numpy(xh).cuda()
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    a = ss.mel_spectrogram(x, 16000, **kw)
    assert a.shape == (1024, 128, 32)
    # the 1024-clip batch is what bench.py --workload cfg3 times: that build, by name, against the oracle on sampled clips
    # (feature.rs:163-174, functions.rs:86-170) -- the 16-clip test above lands on the eight-wave build
    assert ss._lib.lib().ss_last_kernel_name() == BENCH_KERNELS["cfg3"], ss._lib.lib().ss_last_kernel_name()
    pick = [0, 300, 511, 1023]
    want = oracle.mel_spectrogram(oracle.make_params(**CFG3), xh[pick])
    for i, b in enumerate(pick):
        assert _rel(a[b].cpu().numpy(), want[i]) <= RTOL, b
    assert torch.equal(a, ss.mel_spectrogram(x, 16000, **kw))
    # one clip alone takes another build of the kernel (eight waves per CU) than the full batch (twelve): same arithmetic, but
    # the compiler's FMA fusion differs in the last bit -- f32 rounding of the transform, far inside the parity tolerance
    one = ss.mel_spectrogram(x[300], 16000, **kw)
    assert ((one - a[300]).abs() <= 1e-6 * a[300].abs().max()).all()
    assert (a[:, :, 29:] == 0).all() and (a >= 0).all() and torch.isfinite(a).all()
    # |X|^2 is homogeneous of degree 2
    # (exact within one build of the kernel: powers of two commute with every rounding; x[:32] alone takes the eight-wave build)
    assert torch.equal(ss.mel_spectrogram((2.0 * x[:32]).contiguous(), 16000, **kw), 4.0 * ss.mel_spectrogram(x[:32].contiguous(), 16000, **kw))
    assert torch.equal(ss.mel_spectrogram(2.0 * x, 16000, **kw), 4.0 * a)


def test_cfg5_full_batch(ss, oracle):
    import torch

    x = _signal(23, (512, 44100))
    kw = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096)
    a = ss.mfcc_batch(torch.from_numpy(x).cuda(), 44100, **kw)
    assert a.shape == (512, 39, 40) and torch.isfinite(a).all()
    assert ss._lib.lib().ss_last_kernel_name() == BENCH_KERNELS["cfg5"], ss._lib.lib().ss_last_kernel_name()
    assert torch.equal(a, ss.mfcc_batch(torch.from_numpy(x).cuda(), 44100, **kw))
    p = oracle.make_params(**CFG5)
    for b in (0, 255, 511):
        assert _rel(a[b].cpu().numpy(), oracle.mfcc(p, x[b])) <= RTOL


def test_cfg4_corpus_properties(ss, oracle):
    """BASELINE config 4 at full size (360 000 x 1 s clips, one launch): the corpus is 7 distinct clips repeated, so the
    feature block must repeat bit-exactly with period 7 (large-batch indexing: 35 M frames, 23 GB of input), and the
    first period must match the oracle."""
    import torch

    base = _signal(44, (7, 16000))
    n = 360_000
    idx = torch.arange(n, device="cuda") % 7
    x = torch.from_numpy(base).cuda()[idx]  # [360000, 16000]
    out = ss.mfcc_batch(x, 16000)
    del x
    assert out.shape == (n, 98, 13)
    first = out[:7]
    # 360 000 = 7 * 51 428 + 4
    body = out[: 7 * 51_428].view(51_428, 7, 98, 13)
    assert bool((body == first.unsqueeze(0)).all())
    assert bool((out[7 * 51_428:] == first[:4]).all())
    p = oracle.make_params(**CFG1)
    for b in range(7):
        assert _rel(first[b].cpu().numpy(), oracle.mfcc(p, base[b])) <= RTOL


def test_bench_default_line_carries_the_contract_and_the_secondary_configs():
    """The driver's command (`python bench.py --steps 20 --warmup 5`, here with a short CPU leg): ONE JSON line with the contract's
    keys, `roofline` and `cpu_baseline` (the port built on this host, its flags named), the other BASELINE configurations
    (`secondary`: the kernels of BENCH_KERNELS, the ones the parity tests above compare with the oracle), the four-stream figure with
    its one-stream twin, and -- round 6 -- `secondary.cfg2` (time and clock from the same 1000 launches) and `secondary.cfg2_x4`."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "cfg2" in d["config"]["workload"] and d["value"] > 1e9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["kernel"].encode() == BENCH_KERNELS["cfg2"]
    assert r["algorithmic_bytes_per_launch"] == 70754304 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["achieved"] - 70754304 / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert 0.9 < r["traffic"] / 70754304 < 1.1 and 1.0 < r["clock_ghz_measured"] < 2.6
    # the 20-step headline prints no cycle count: its clock comes from later launches (round 6) -- secondary.cfg2 carries the coherent one
    assert r["cycles_per_launch"] is None and "secondary.cfg2" in r["cycles_per_launch_see"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1e4 and ("-O3 -march=native" in c["sample"] or "portable" in c["sample"])
    sec = d["secondary"]
    s2 = sec["cfg2"]  # 1000 single-stream steps inside the library: events and in-kernel stamps over the same launches
    assert "error" not in s2, s2
    assert s2["kernel"].encode() == BENCH_KERNELS["cfg2"] and s2["steps"] == 1000 and s2["streams"] == 1
    assert "timed launches themselves" in s2["clock_source"] and 1.0 < s2["clock_ghz_measured"] < 2.6
    assert abs(s2["cycles_per_launch"] - s2["avg_launch_us"] * 1e3 * s2["clock_ghz_measured"]) < 1.0
    assert 0.2 < s2["frac"] < 0.6 and 0.3 < s2["valu_floor_frac"] < 0.8
    x4 = sec["cfg2_x4"]  # four batches per launch (ss_mfcc_batches_device): per-batch figures
    assert "error" not in x4, x4
    assert x4["kernel"] == "ss_mfcc_c256m<10,exact,bank421,sym>" and x4["batches_per_launch"] == 4
    assert x4["algorithmic_bytes_per_launch"] == 70754304 and abs(x4["launch_us"] - 4 * x4["avg_launch_us"]) < 1e-6
    assert 0.9 * s2["frac"] < x4["frac"] < 0.6
    x8 = sec["cfg2_x8"]  # eight batches per launch: the one-launch corpus rate
    assert "error" not in x8 and x8["kernel"] == x4["kernel"] and x8["batches_per_launch"] == 8 and 0.95 * x4["frac"] < x8["frac"] < 0.6
    for wl, kern, bytes_ in (("cfg3", "ss_mel_c1024m<w12,mel6321>", 82313216), ("cfg5", "ss_mfcc_c2048m<exact,mel8321,w12>", 93511680)):
        y4 = sec[wl + "_x4"]
        assert "error" not in y4, y4
        assert y4["kernel"] == kern and y4["batches_per_launch"] == 4 and y4["algorithmic_bytes_per_launch"] == bytes_
        assert 0.9 * sec[wl]["frac"] < y4["frac"] < 0.6
    assert d["value_pipelined"] > 0.9 * d["value"] and d["pipelined"]["streams"] == 4
    assert d["pipelined"]["value_one_stream"] == s2["value"] and d["pipelined"]["steps"] == 1000
    for wl, bytes_ in (("cfg3", 82313216), ("cfg5", 93511680), ("cfg4", 24874560000)):
        e = sec[wl]
        assert "error" not in e, e
        assert e["kernel"].encode() == BENCH_KERNELS["cfg2" if wl == "cfg4" else wl]
        assert e["algorithmic_bytes_per_launch"] == bytes_ and abs(e["frac"] - bytes_ / (e["avg_launch_us"] * 1e-6) / 8e12) < 1e-9
        assert 0.05 < e["frac"] < 0.6
        if e.get("clock_ghz_measured"):
            assert 1.0 < e["clock_ghz_measured"] < 2.6 and abs(e["cycles_per_launch"] - e["avg_launch_us"] * 1e3 * e["clock_ghz_measured"]) < 1.0
        if wl in ("cfg3", "cfg5"):  # round 6: their cycle counts share launches with their times, like secondary.cfg2's
            assert "timed launches themselves" in e["clock_source"] and 1.0 < e["clock_ghz_measured"] < 2.6


def test_cpp_mirror_parity(tmp_path, oracle):
    """The header-only C++ mirror of the crate's API (include/speechsauce_amd.hpp), built with plain g++ against the
    library: mfcc -> cmvn on a seeded clip, compared with the oracle."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    x = _signal(51, 16000)
    x.tofile(tmp_path / "x.f32")
    src = tmp_path / "t.cpp"
    src.write_text(
        '#include "speechsauce_amd.hpp"\n#include <cstdio>\n'
        "int main(int argc, char **argv) {\n"
        "  std::vector<float> x(16000); FILE *f = std::fopen(argv[1], \"rb\");\n"
        "  if (!f || std::fread(x.data(), 4, x.size(), f) != x.size()) return 2; std::fclose(f);\n"
        "  speechsauce::SpeechConfig cfg0 = speechsauce::SpeechConfigBuilder(16000).build();\n"
        "  speechsauce::SpeechConfig cfg = cfg0;  // Clone (config.rs:98): the copy shares the handle and outlives the original\n"
        "  cfg0 = speechsauce::SpeechConfig(16000, 2048, 0.032f, 0.032f, 13, 128, 0.f, 8000.f, true);\n"
        "  if (cfg.frame_size() != 320 || cfg.window_size_half() != 256 || cfg.window().size() != 512 || cfg.handle() == cfg0.handle()) return 5;\n"
        "  if (cfg0.frame_size() != 512 || cfg0.wnorm() != 1.0f / (4194304.0f / 1024.0f) || cfg0.window()[0] <= 0.f) return 6;\n"
        "  auto m = speechsauce::mfcc(x, cfg);\n"
        "  auto n = speechsauce::cmvn(m.data, m.rows, m.cols, true);\n"
        "  f = std::fopen(argv[2], \"wb\"); std::fwrite(m.data.data(), 4, m.data.size(), f); std::fwrite(n.data(), 4, n.size(), f);\n"
        "  std::fclose(f); return m.rows == 98 && m.cols == 13 ? 0 : 3; }\n")
    libdir = os.path.join(root, "mfcc-rust_amd", "lib")
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), str(src), "-L", libdir, "-lspeechsauce_amd",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    subprocess.run([str(exe), str(tmp_path / "x.f32"), str(tmp_path / "o.f32")], check=True)
    o = np.fromfile(tmp_path / "o.f32", dtype=np.float32)
    m, n = o[:98 * 13].reshape(98, 13), o[98 * 13:].reshape(98, 13)
    want = oracle.mfcc(oracle.make_params(**CFG1), x)
    assert _rel(m, want) <= RTOL
    assert _rel(n, oracle.cmvn(m, True)) <= RTOL


def test_mfcc_4096_other_filter_counts(ss, oracle, sslib):
    """The 4096-point kernel with fewer than 256 filters (any even count): 128 mels / 20 cepstra, 100 mels with a window, mfe."""
    import torch

    sr = 44100
    x = _signal(26, (4, sr))
    xd = torch.from_numpy(x).cuda()
    for flen, M, C, sw in ((4096, 128, 20, {}), (3000, 100, 13, dict(mfcc_window="hann")), (4096, 64, 40, dict(dc_elimination=False)),
                           (4096, 128, 20, dict(mfcc_window="hann", spectrum_exponent=2, dct_norm="ortho")), (3001, 256, 13, dict(mfcc_window="vorbis", spectrum_exponent=2)),
                           # the DCT stage's lane layouts: twice folded up to 43 coefficients when the filter count is a multiple
                           # of 4 (41..43: the odd coefficients start on the next even lane), once folded in two passes otherwise
                           (4096, 256, 41, {}), (4096, 256, 42, {}), (4096, 256, 43, dict(dc_elimination=False)), (4096, 256, 44, {}), (4096, 256, 64, {}),
                           (4096, 254, 36, {}), (4096, 130, 33, dict(dc_elimination=False)), (4096, 255, 40, {}), (4096, 60, 5, {}), (4096, 252, 1, {})):
        kw = dict(frame_length=flen / sr, frame_stride=1024 / sr, num_cepstral=C, num_filters=M, fft_length=4096)
        p = oracle.make_params(sample_rate=sr, fft_points=4096, frame_length=flen / sr, frame_stride=1024 / sr, num_cepstral=C,
                               num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c2048<"), sslib.ss_last_kernel_name()
        for b in (0, 3):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, M, C, sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        msw = {k: v for k, v in sw.items() if k not in ("dc_elimination", "dct_norm")}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **msw)
        assert b"mfe" in sslib.ss_last_kernel_name()
        wf, we = oracle.mfe(p, x[3])
        assert _rel(feat[3].cpu().numpy(), wf) <= RTOL and _rel(en[3].cpu().numpy(), we) <= RTOL


def test_cfg5_mfe(ss, oracle, sslib):
    """mfe at the high-resolution configuration: the mfe build of the 4096-point kernel (256 filters, 19 of them empty)."""
    import torch

    x = _signal(24, (5, 44100))
    kw = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_filters=256, fft_length=4096)
    feat, en = ss.mfe_batch(torch.from_numpy(x).cuda(), 44100, **kw)
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c2048<") and b"mfe" in sslib.ss_last_kernel_name()
    assert feat.shape == (5, 39, 256) and en.shape == (5, 39)
    p = oracle.make_params(**CFG5)
    for b in (0, 4):
        wf, we = oracle.mfe(p, x[b])
        assert _rel(feat[b].cpu().numpy(), wf) <= RTOL and _rel(en[b].cpu().numpy(), we) <= RTOL
    assert (feat.cpu().numpy() == np.float32(1.1920929e-7)).sum() >= 5 * 39 * 19  # the empty filters: exactly EPS


def test_cfg5_windowed(ss, oracle, sslib):
    """Frame window at the high-resolution configuration: windowed builds of the 4096-point kernel (mfcc and mfe),
    full frames and frames shorter than the FFT."""
    import torch

    x = _signal(25, (4, 44100))
    xd = torch.from_numpy(x).cuda()
    for flen in (4096, 3000):
        kw = dict(frame_length=flen / 44100, frame_stride=1024 / 44100, num_filters=256, fft_length=4096)
        p = oracle.make_params(**dict(CFG5, frame_length=flen / 44100, mfcc_window="hann"))
        got = ss.mfcc_batch(xd, 44100, num_cepstral=40, mfcc_window="hann", **kw).cpu().numpy()
        name = sslib.ss_last_kernel_name()
        assert name.startswith(b"ss_mfcc_c2048<") and b"win" in name, name
        assert _rel(got[3], oracle.mfcc(p, x[3])) <= RTOL
        feat, en = ss.mfe_batch(xd, 44100, mfcc_window="hann", **kw)
        assert b"mfe,win" in sslib.ss_last_kernel_name()
        wf, we = oracle.mfe(p, x[0])
        assert _rel(feat[0].cpu().numpy(), wf) <= RTOL and _rel(en[0].cpu().numpy(), we) <= RTOL


def test_mfcc_2048_kernel(ss, oracle, sslib):
    """MFCC / mfe at fft_points = 2048 (e.g. 22.05 kHz, 2048-sample frames, hop 512, 128 mels, 20 cepstra): the
    two-frames-per-wave kernel, with and without a frame window, odd frame counts, frames shorter than the FFT."""
    import torch

    sr = 22050
    x = _signal(26, (7, sr))
    xd = torch.from_numpy(x).cuda()
    for flen, sw in ((2048, {}), (2048, dict(mfcc_window="hann")), (1764, dict(spectrum_exponent=2)), (1500, dict(mfcc_window="vorbis", dct_norm="ortho"))):
        kw = dict(frame_length=flen / sr, frame_stride=512 / sr, num_cepstral=20, num_filters=128, fft_length=2048)
        p = oracle.make_params(sample_rate=sr, fft_points=2048, frame_length=flen / sr, frame_stride=512 / sr, num_cepstral=20,
                               num_filters=128, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c1024"), sslib.ss_last_kernel_name()
        assert got.shape == (7, oracle.num_frames(p, sr), 20)
        for b in (0, 3, 6):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **sw)
        assert b"mfe" in sslib.ss_last_kernel_name()
        wf, we = oracle.mfe(p, x[6])
        assert _rel(feat[6].cpu().numpy(), wf) <= RTOL and _rel(en[6].cpu().numpy(), we) <= RTOL
    # 26 filters (13 sum/difference terms: the product runs over whole float4s), 13 cepstra, no dc elimination
    kw = dict(frame_length=2048 / sr, frame_stride=441 * 2 / sr, num_cepstral=13, num_filters=26, fft_length=2048, dc_elimination=False)
    p = oracle.make_params(sample_rate=sr, fft_points=2048, frame_length=2048 / sr, frame_stride=882 / sr, num_cepstral=13,
                           num_filters=26, dc_elimination=False)
    got = ss.mfcc_batch(xd, sr, **kw).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c1024")
    assert _rel(got[2], oracle.mfcc(p, x[2])) <= RTOL


def test_mfcc_1024_kernel(ss, oracle, sslib):
    """MFCC / mfe at fft_points = 1024 (e.g. 22.05 kHz, 1024-sample frames, hop 256, 64 mels, 20 cepstra): the
    permlane16 kernel, with and without a frame window, short frames, odd frame counts, power spectrum."""
    import torch

    sr = 22050
    x = _signal(27, (9, sr))
    xd = torch.from_numpy(x).cuda()
    for flen, M, C, sw in ((1024, 64, 20, {}), (1024, 128, 32, dict(mfcc_window="hann")), (882, 40, 13, dict(spectrum_exponent=2)),
                           (700, 26, 13, dict(mfcc_window="vorbis", dct_norm="ortho", dc_elimination=False))):
        kw = dict(frame_length=flen / sr, frame_stride=256 / sr, num_cepstral=C, num_filters=M, fft_length=1024)
        p = oracle.make_params(sample_rate=sr, fft_points=1024, frame_length=flen / sr, frame_stride=256 / sr, num_cepstral=C,
                               num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c512"), sslib.ss_last_kernel_name()
        assert got.shape == (9, oracle.num_frames(p, sr), C)
        for b in (0, 4, 8):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, M, sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        msw = {k: v for k, v in sw.items() if k != "dc_elimination"}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **msw)
        assert b"mfe" in sslib.ss_last_kernel_name()
        wf, we = oracle.mfe(p, x[8])
        assert _rel(feat[8].cpu().numpy(), wf) <= RTOL and _rel(en[8].cpu().numpy(), we) <= RTOL


def test_mfcc_512_wide_bank_kernel(ss, oracle, sslib):
    """fft_points = 512 with more than 48 filters (log-mel front ends: 25 ms frames, 64 / 80 mels): the wide-bank kernel;
    mfe and MFCC, window, power spectrum, Slaney bank over the whole spectrum, 20 ms frames, short clips."""
    import torch

    sr = 16000
    x = _signal(35, (9, sr))
    xd = torch.from_numpy(x).cuda()
    for flen, M, C, sw in ((400, 80, 13, {}), (400, 64, 16, dict(mfcc_window="hann", spectrum_exponent=2)), (320, 64, 13, {}),
                           (512, 80, 13, dict(mel_scale="slaney", mel_norm="slaney", mfcc_window="hann", spectrum_exponent=2, dct_norm="ortho")),
                           (400, 57, 12, dict(dc_elimination=False)), (400, 80, 30, {}), (320, 40, 20, dict(dc_elimination=False)),
                           (400, 64, 32, dict(dct_norm="ortho"))):  # more than 16 cepstra: two coefficients per lane
        kw = dict(frame_length=flen / sr, frame_stride=0.01, num_cepstral=C, num_filters=M, fft_length=512)
        p = oracle.make_params(sample_rate=sr, fft_points=512, frame_length=flen / sr, frame_stride=0.01, num_cepstral=C, num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256w<"), sslib.ss_last_kernel_name()
        assert got.shape == (9, oracle.num_frames(p, sr), C)
        for b in (0, 4, 8):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, M, sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        msw = {k: v for k, v in sw.items() if k not in ("dc_elimination", "dct_norm")}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **msw)
        # (mfe takes no cepstrum count: with at most 48 filters the headline kernel's mfe build serves it)
        assert b"mfe" in sslib.ss_last_kernel_name() and sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256w<" if M > 48 else b"ss_mfcc_c256<")
        for b in (0, 8):
            wf, we = oracle.mfe(p, x[b])
            assert _rel(feat[b].cpu().numpy(), wf) <= RTOL and _rel(en[b].cpu().numpy(), we) <= RTOL
    # three frames per clip (the per-lane division path), odd clip count
    p = oracle.make_params(sample_rate=sr, fft_points=512, frame_length=0.025, num_filters=80)
    xs = _signal(36, (5, 400 + 160 * 3 + 2))
    got = ss.mfcc_batch(torch.from_numpy(xs).cuda(), sr, frame_length=0.025, num_filters=80).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256w<") and got.shape[1] == oracle.num_frames(p, xs.shape[1]) < 4
    for b in range(5):
        assert _rel(got[b], oracle.mfcc(p, xs[b])) <= RTOL, b


def test_512_point_combinations_without_a_headline_build(ss, oracle, sslib):
    """At 512 points the headline kernel has mfe / window / power builds for the default frame shape only; other shapes (25 ms
    frames with 40 mels: the log-mel front end) fall to the wide-bank kernel, not to the generic one."""
    import torch

    sr = 16000
    x = _signal(37, (7, sr))
    xd = torch.from_numpy(x).cuda()
    for flen, M, sw in ((400, 40, {}), (400, 40, dict(mfcc_window="hann")), (512, 26, dict(mfcc_window="vorbis", spectrum_exponent=2)),
                        (256, 40, dict(framing="center", mfcc_window="hann"))):
        kw = dict(frame_length=flen / sr, frame_stride=0.01, num_filters=M, fft_length=512)
        p = oracle.make_params(sample_rate=sr, fft_points=512, frame_length=flen / sr, frame_stride=0.01, num_filters=M, **sw)
        feat, en = ss.mfe_batch(xd, sr, **kw, **sw)
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256w<") and b"mfe" in sslib.ss_last_kernel_name(), sslib.ss_last_kernel_name()
        for b in (0, 6):
            wf, we = oracle.mfe(p, x[b])
            assert _rel(feat[b].cpu().numpy(), wf) <= RTOL and _rel(en[b].cpu().numpy(), we) <= RTOL, (flen, M, sw, b)
        if sw:
            got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
            assert not sslib.ss_last_kernel_name().startswith(b"ss_front_generic"), sslib.ss_last_kernel_name()
            for b in (0, 6):
                assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, M, sw, b)


@pytest.mark.parametrize("sr,nfft,hop,M,C,kernel", [(22050, 1024, 256, 41, 13, b"ss_mfcc_c512"), (22050, 1024, 256, 127, 32, b"ss_mfcc_c512"),
                                                    (44100, 2048, 512, 127, 20, b"ss_mfcc_c1024"), (44100, 2048, 512, 25, 13, b"ss_mfcc_c1024"),
                                                    (44100, 4096, 1024, 255, 40, b"ss_mfcc_c2048"), (44100, 4096, 1024, 101, 13, b"ss_mfcc_c2048")])
def test_odd_filter_counts_on_the_symmetric_dct_kernels(ss, oracle, sslib, sr, nfft, hop, M, C, kernel):
    """The sum / difference DCT of the 1024-, 2048- and 4096-point kernels with an odd number of filters: the middle filter pairs
    with itself."""
    import torch

    x = _signal(39, (3, nfft + 8 * hop))
    kw = dict(frame_length=nfft / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
    for sw in ({}, dict(dct_norm="ortho", dc_elimination=False)):
        p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=nfft / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, **sw)
        got = ss.mfcc_batch(torch.from_numpy(x).cuda(), sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(kernel), sslib.ss_last_kernel_name()
        for b in range(3):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (sw, b)


@pytest.mark.parametrize("sr,nfft,flen,step,M,C,kernel", [
    (8000, 256, 200, 80, 26, 13, b"ss_mfcc_c256x2<"), (16000, 512, 400, 160, 40, 13, b"ss_mfcc_c256w<"), (16000, 512, 401, 160, 80, 20, b"ss_mfcc_c256w<"),
    (22050, 1024, 883, 221, 64, 20, b"ss_mfcc_c512"), (44100, 2048, 2048, 512, 128, 20, b"ss_mfcc_c1024"), (44100, 4096, 4096, 1024, 256, 40, b"ss_mfcc_c2048"),
    (44100, 4096, 3001, 1000, 100, 13, b"ss_mfcc_c2048")])
def test_fused_preemphasis_on_the_dedicated_kernels(ss, oracle, sslib, sr, nfft, flen, step, M, C, kernel):
    """y[i] = x[i] - c x[(i - shift) mod L] (processing.rs:31-53) fused into the loaders of the dedicated frame kernels: shifts 1
    and 3, with and without a window, centred frames where the kernel has them; the first frames wrap to the end of the clip."""
    import torch

    x = _signal(71, (4, flen + 9 * step + 1))
    xd = torch.from_numpy(x).cuda()
    kw = dict(frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
    cases = [dict(preemph_coef=0.97), dict(preemph_coef=0.9, preemph_shift=3, mfcc_window="hann", spectrum_exponent=2)]
    if nfft in (512, 1024, 2048) and flen % 4 == 0:
        cases.append(dict(preemph_coef=0.97, framing="center", pad_mode="reflect"))
    for sw in cases:
        p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(kernel), (sslib.ss_last_kernel_name(), sw)
        for b in (0, 3):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **sw)
        wf, we = oracle.mfe(p, x[3])
        assert _rel(feat[3].cpu().numpy(), wf) <= RTOL and _rel(en[3].cpu().numpy(), we) <= RTOL, sw


def test_configuration_that_outgrows_a_dedicated_kernel_falls_back(ss, oracle, sslib):
    """A windowed 4096-point configuration whose table block (47 long filters, 45 cosine rows, the window: 167 636 bytes) does not fit the
    dedicated kernel's LDS budget: its launcher declines before launching and the dispatcher moves on to the generic kernel."""
    import torch

    sr = 22050
    kw = dict(frame_length=0.13506802721088434, frame_stride=0.029489795918367347, num_cepstral=45, num_filters=47, fft_length=4096)
    sw = dict(mfcc_window="hann", spectrum_exponent=2)
    x = _signal(75, (3, 21076))
    got = ss.mfcc_batch(torch.from_numpy(x).cuda(), sr, **kw, **sw).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_front_generic<11>")
    p = oracle.make_params(sample_rate=sr, fft_points=4096, frame_length=kw["frame_length"], frame_stride=kw["frame_stride"], num_cepstral=45,
                           num_filters=47, **sw)
    for b in range(3):
        assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL


@pytest.mark.parametrize("sr,nfft,hop,M,C,kernel", [(22050, 1024, 256, 64, 40, b"ss_mfcc_c512"), (44100, 2048, 512, 128, 64, b"ss_mfcc_c1024"),
                                                    (44100, 2048, 512, 127, 33, b"ss_mfcc_c1024")])
def test_more_than_32_cepstra_on_the_1024_and_2048_point_kernels(ss, oracle, sslib, sr, nfft, hop, M, C, kernel):
    """33..64 cepstra: lane c also forms coefficient c + 32."""
    import torch

    x = _signal(77, (3, nfft + 9 * hop))
    kw = dict(frame_length=nfft / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
    for sw in ({}, dict(dct_norm="ortho", dc_elimination=False, mfcc_window="hann")):
        p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=nfft / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, **sw)
        got = ss.mfcc_batch(torch.from_numpy(x).cuda(), sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(kernel), sslib.ss_last_kernel_name()
        for b in range(3):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (sw, b)


def test_mfcc_256_kernel(ss, oracle, sslib):
    """MFCC / mfe at fft_points = 256 (8 kHz telephony front ends): two frames per complex transform.  20 ms and 25 ms frames,
    odd hops (scalar loads: no alignment assumptions), window, power spectrum, filter counts up to 48, batches whose frame count
    is not a multiple of 8, clips with fewer than 8 frames (the per-lane division path), a single clip."""
    import torch

    sr = 8000
    x = _signal(31, (11, 2 * sr))
    xd = torch.from_numpy(x).cuda()
    for flen, step, M, C, sw in ((160, 80, 40, 13, {}), (200, 80, 26, 13, dict(mfcc_window="vorbis")), (256, 81, 48, 16, dict(spectrum_exponent=2)),
                                 (160, 80, 40, 20, {}), (200, 80, 48, 32, dict(mfcc_window="hann", dc_elimination=False)),
                                 (161, 77, 23, 12, dict(mfcc_window="hann", dct_norm="ortho", dc_elimination=False)),
                                 (160, 80, 40, 13, dict(mel_scale="slaney", mel_norm="slaney"))):
        kw = dict(frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, fft_length=256)
        p = oracle.make_params(sample_rate=sr, fft_points=256, frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C,
                               num_filters=M, **sw)
        got = ss.mfcc_batch(xd, sr, **kw, **sw).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256x2<"), sslib.ss_last_kernel_name()
        assert got.shape == (11, oracle.num_frames(p, 2 * sr), C)
        for b in (0, 5, 10):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, step, M, sw, b)
        mkw = {k: v for k, v in kw.items() if k != "num_cepstral"}
        msw = {k: v for k, v in sw.items() if k != "dc_elimination"}
        feat, en = ss.mfe_batch(xd, sr, **mkw, **msw)
        assert b"mfe" in sslib.ss_last_kernel_name() and sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256x2<")
        for b in (0, 10):
            wf, we = oracle.mfe(p, x[b])
            assert _rel(feat[b].cpu().numpy(), wf) <= RTOL and _rel(en[b].cpu().numpy(), we) <= RTOL
    # fewer than 8 frames per clip, odd clip count; one clip through the 1-D entry point (views at odd offsets: unaligned starts)
    p = oracle.make_params(sample_rate=sr, fft_points=256)
    xs = _signal(32, (5, 160 + 80 * 5 + 3))
    got = ss.mfcc_batch(torch.from_numpy(xs).cuda(), sr, fft_length=256).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256x2<")
    assert got.shape[1] == oracle.num_frames(p, xs.shape[1]) and got.shape[1] < 8
    for b in range(5):
        assert _rel(got[b], oracle.mfcc(p, xs[b])) <= RTOL, b
    x1 = _signal(33, 4001)
    assert _rel(ss.mfcc(x1[1:], sr, fft_length=256), oracle.mfcc(p, x1[1:])) <= RTOL


def _on_lab_test_kernel_variants_agree(ss, sslib):
    """The generic kernel and the production kernel compute the same MFCCs (ss_debug_force_generic, the test aid of
    include/speechsauce_amd_debug.h, routes every configuration to ss_front_generic)."""
    import torch

    x = torch.from_numpy((np.random.default_rng(5).standard_normal((37, 16000)) * 0.1).astype(np.float32)).cuda()
    outs, names = [], []
    try:
        for force in (0, 1):
            sslib.ss_debug_force_generic(force)
            outs.append(ss.mfcc_batch(x, 16000).cpu().numpy())
            names.append(sslib.ss_last_kernel_name().decode())
    finally:
        sslib.ss_debug_force_generic(0)
    assert names[0].startswith("ss_mfcc_c256<") and names[1].startswith("ss_front_generic")
    assert _rel(outs[1], outs[0]) <= 2e-5


def test_kernel_variants_agree(ss, sslib, sslab):
    """The dedicated kernel runs on the PRODUCT library (the code that ships: its kernels are compiled without the lab
    switches), only the forced-generic reference on the LAB library (which owns ss_debug_force_generic); the product build's
    dedicated kernel must also give the lab build's dedicated kernel's bits -- a miscompile or a dispatch difference of the
    product objects would show here (round-4 advisor finding)."""
    import torch

    x = torch.from_numpy((np.random.default_rng(5).standard_normal((37, 16000)) * 0.1).astype(np.float32)).cuda()
    prod = ss.mfcc_batch(x, 16000).cpu().numpy()
    assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256<")
    with ss._lib.use_library(sslab):
        lab = ss.mfcc_batch(x, 16000).cpu().numpy()
        assert sslab.ss_last_kernel_name() == sslib.ss_last_kernel_name()
        try:
            sslab.ss_debug_force_generic(1)
            gen = ss.mfcc_batch(x, 16000).cpu().numpy()
            assert sslab.ss_last_kernel_name().startswith(b"ss_front_generic")
        finally:
            sslab.ss_debug_force_generic(0)
        _on_lab_test_kernel_variants_agree(ss, sslab)
    assert np.array_equal(prod, lab)
    assert _rel(gen, prod) <= 2e-5



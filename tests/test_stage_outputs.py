"""The stage outputs the reference exposes as pub fns -- processing::stack_frames (processing.rs:65-129),
processing::power_spectrum(frames, fft_points) (:179-181), functions::stft1 / stft2 (functions.rs:199-233, :86-123) -- through the
same boundary as mfcc / mfe: host-pointer entry points of the C ABI, the Python front, and the device-pointer forms, which
must agree bit for bit.  Plus the failure path of the whole-line tile: a lost hand-off becomes SS_ERR_DEVICE, not maybe-NaNs.
CPU part: the oracle restatements against independent numpy."""
import ctypes as C

import numpy as np
import pytest

from common import rel


def _signal(seed, shape, scale=0.1):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


# ---- CPU: the oracle's restatements ----

def test_oracle_power_spectrum_frames_is_rfft_magnitude_over_n(oracle):
    """processing.rs:143-181: zero-pad to fft_points, R2C FFT, magnitude, times 1/fft_points."""
    for rows, cols, n in ((7, 320, 512), (3, 512, 512), (5, 400, 400), (4, 100, 256), (2, 1, 64)):
        f = _signal(rows + cols, (rows, cols))
        want = np.abs(np.fft.rfft(f.astype(np.float64), n=n, axis=1)) / n
        got = oracle.power_spectrum_frames(f, n)
        assert got.shape == (rows, n // 2 + 1)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))


def test_python_stack_frames_rejects_the_old_positional_flag(sslib):
    """Round 3's stack_frames had zero_padding where the reference has `filter` (processing.rs:65-76): an old positional call
    gets a clear TypeError before anything is indexed or launched (no device needed)."""
    import speechsauce_amd as ss

    x = _signal(1, 4000)
    with pytest.raises(TypeError, match="filter"):
        ss.stack_frames(x, 16000, 0.02, 0.01, True)


def test_oracle_stack_frames_contract_and_padded(oracle):
    x = _signal(3, 16000)
    p = oracle.make_params()
    fr = oracle.stack_frames(p, x)
    assert fr.shape == (98, 320)
    for t in (0, 1, 50, 97):
        np.testing.assert_array_equal(fr[t], x[t * 160: t * 160 + 320].astype(np.float64))
    # the power spectrum of those frames is the signal-level stage
    np.testing.assert_allclose(oracle.power_spectrum_frames(fr.astype(np.float32), 512), oracle.power_spectrum(p, x), rtol=0, atol=1e-12)
    # zero_padding = true: the reference's own test shape (lib.rs:50-68)
    pp = oracle.make_params(frame_length=0.02, frame_stride=0.02, framing="padded")
    big = _signal(4, 100_000)
    frp = oracle.stack_frames(pp, big)
    T = oracle.num_frames(pp, big.size)
    assert frp.shape == (T, 320)
    assert T == 312  # ceil((100000 - 320) / 320): one more than the floor of the default framing
    np.testing.assert_array_equal(frp[-1], big[(T - 1) * 320: T * 320].astype(np.float64))
    assert oracle.stack_frames(oracle.make_params(frame_length=0.02, frame_stride=0.02), big).shape == (311, 320)
    # the literal exact_chunks copy: all-zero rows for more than two frames
    assert not oracle.stack_frames(oracle.make_params(framing="literal"), x).any()
    # the `filter` argument
    ph = oracle.make_params(mfcc_window="hann")
    np.testing.assert_allclose(oracle.stack_frames(ph, x)[5], x[800:1120].astype(np.float64) * oracle.hann_window(320), rtol=0, atol=0)


# ---- GPU ----

@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,n", [(98, 320, 512), (33, 512, 512), (50, 400, 512), (17, 441, 1024), (9, 2048, 2048), (40, 400, 400),
                                         (5, 64, 64), (1000, 320, 512)])
def test_power_spectrum_of_a_frames_matrix(ss, oracle, sslib, rows, cols, n):
    """speechsauce.power_spectrum(frames, fft_points): host form == device form bit for bit, both within 1e-5 of the oracle."""
    import torch

    f = _signal(rows * 7 + cols, (rows, cols))
    want = oracle.power_spectrum_frames(f, n)
    host = ss.power_spectrum(f, n)
    assert host.shape == want.shape and host.dtype == np.float32
    dev = ss.power_spectrum(torch.from_numpy(f).cuda(), n)
    np.testing.assert_array_equal(dev.cpu().numpy(), host)
    assert rel(host, want) <= 1e-5, (rows, cols, n, sslib.ss_last_kernel_name())
    if (cols, n) == (320, 512):
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256<10,exact,power>")
    with pytest.raises(ss.SpeechSauceError):
        ss.power_spectrum(np.zeros((2, n + 1), np.float32), n)  # ndfft_r2c would assert
    with pytest.raises(TypeError):
        ss.power_spectrum(np.zeros((2, 8)), n)
    with pytest.raises(ValueError):
        ss.power_spectrum(np.zeros(8, np.float32), n)


@pytest.mark.gpu
def test_power_spectrum_of_a_signal_host_and_device(ss, oracle, sslib):
    """The signal-level stage (stack_frames + power_spectrum, as mfe uses them) through the host-pointer entry points."""
    import torch

    x = _signal(11, (5, 16000))
    p = oracle.make_params()
    host = ss.power_spectrum_of_signal(x, 16000)
    assert host.shape == (5, 98, 257)
    dev = ss.power_spectrum_of_signal(torch.from_numpy(x).cuda(), 16000).cpu().numpy()
    np.testing.assert_array_equal(host, dev)
    for b in range(5):
        assert rel(host[b], oracle.power_spectrum(p, x[b])) <= 1e-5
    one = ss.power_spectrum_of_signal(x[2], 16000)
    np.testing.assert_array_equal(one, host[2])
    # a batch large enough for the chunked two-stream pipeline (> 1 MB of samples), other sizes on the generic kernel
    xb = _signal(12, (40, 22050))
    kw = dict(frame_length=0.025, frame_stride=0.010, fft_length=1024)
    hb = ss.power_spectrum_of_signal(xb, 22050, **kw)
    db = ss.power_spectrum_of_signal(torch.from_numpy(xb).cuda(), 22050, **kw).cpu().numpy()
    np.testing.assert_array_equal(hb, db)
    pb = oracle.make_params(sample_rate=22050, fft_points=1024, frame_length=0.025, frame_stride=0.010)
    assert rel(hb[7], oracle.power_spectrum(pb, xb[7])) <= 1e-5
    # frames cut on the device, then the frames-matrix entry point: the same numbers as the fused stage
    fr = ss.stack_frames(x[3], 16000, frame_length=0.020, frame_stride=0.010)
    np.testing.assert_array_equal(ss.power_spectrum(fr, 512), host[3])


@pytest.mark.gpu
def test_stack_frames(ss, oracle):
    import torch

    x = _signal(21, 16000)
    for kw, okw in ((dict(frame_length=0.020, frame_stride=0.010), {}),
                    (dict(frame_length=0.020, frame_stride=0.020, zero_padding=True), dict(frame_stride=0.02, framing="padded")),
                    (dict(frame_length=0.025, frame_stride=0.010, mfcc_window="hann"), dict(frame_length=0.025, mfcc_window="hann")),
                    (dict(frame_length=0.020, frame_stride=0.010, framing="literal"), dict(framing="literal")),
                    (dict(frame_length=0.032, frame_stride=0.010, framing="center"), dict(frame_length=0.032, framing="center"))):
        want = oracle.stack_frames(oracle.make_params(**okw), x)
        host = ss.stack_frames(x, 16000, **kw)
        assert host.shape == want.shape, kw
        dev = ss.stack_frames(torch.from_numpy(x).cuda(), 16000, **kw).cpu().numpy()
        np.testing.assert_array_equal(host, dev)
        if "mfcc_window" in kw:
            np.testing.assert_allclose(host, want, rtol=0, atol=1e-7)
        else:
            np.testing.assert_array_equal(host.astype(np.float64), want)
    # the reference's own test_stack_frames shape (lib.rs:50-68): 1e6 samples, 20 ms / 20 ms, zero_padding = true
    big = _signal(22, 1_000_000)
    fr = ss.stack_frames(big, 16000, frame_length=0.02, frame_stride=0.02, zero_padding=True)
    assert fr.shape == (3124, 320)
    np.testing.assert_array_equal(fr[100], big[32000:32320])
    with pytest.raises(ss.SpeechSauceError):
        ss.stack_frames(x[:100], 16000)
    # stack_frames has no FFT dependency (processing.rs:65-129): frames longer than any FFT length of the MFCC path are fine
    # (round 3 built its config with fft_length = 512 and rejected these)
    for sr, fl, st in ((44100, 0.025, 0.010), (48000, 0.025, 0.010), (48000, 0.2, 0.1)):
        sig = _signal(23, sr)
        flen, step = int(np.floor(np.float32(sr) * np.float32(fl) + np.float32(0.5))), int(np.floor(np.float32(sr) * np.float32(st) + np.float32(0.5)))
        T = int(np.floor(np.float32(sr - flen) / np.float32(step)))
        host = ss.stack_frames(sig, sr, frame_length=fl, frame_stride=st)
        assert host.shape == (T, flen), (sr, fl, host.shape)
        idx = np.arange(T)[:, None] * step + np.arange(flen)[None, :]
        np.testing.assert_array_equal(host, sig[idx])
        np.testing.assert_array_equal(ss.stack_frames(torch.from_numpy(sig).cuda(), sr, frame_length=fl, frame_stride=st).cpu().numpy(), host)
    # `filter`, as in the reference: a callable frame_len -> (1, frame_len) window (processing.rs:122-126)
    ham = lambda n: np.hamming(n).astype(np.float32)[None, :]
    np.testing.assert_array_equal(ss.stack_frames(x, 16000, 0.02, 0.01, ham), ss.stack_frames(x, 16000, 0.02, 0.01) * ham(320))
    # the switch path of long frames picks an FFT length that holds the frame
    assert ss.stack_frames(_signal(24, 44100), 44100, 0.025, 0.010, framing="center").shape[1] == 1103


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_fft,hop,n", [(16000, 2048, 512, 16000), (16000, 512, 256, 16000), (16000, 1024, 512, 9000),
                                            (44100, 4096, 1024, 30000), (16000, 400, 160, 8000), (8000, 256, 64, 4000)])
def test_stft_host_and_device(ss, oracle, sslib, sr, n_fft, hop, n):
    """speechsauce.stft (stft1 for 1-D, stft2 for [C, L]): complex64 rows; host == device bit for bit; <= 1e-5 of the oracle;
    the trailing n_pad rows are exact zeros (functions.rs:121)."""
    import torch

    x = _signal(n + n_fft, (3, n))
    kw = dict(frame_length=hop / sr, fft_length=n_fft)
    host = ss.stft(x, sr, **kw)
    p = oracle.make_params(sample_rate=sr, fft_points=n_fft, frame_length=hop / sr, frame_stride=hop / sr)
    want = oracle.stft(p, x)
    assert host.dtype == np.complex64 and host.shape == want.shape
    dev = ss.stft(torch.from_numpy(x).cuda(), sr, **kw)
    assert dev.dtype == torch.complex64
    np.testing.assert_array_equal(dev.cpu().numpy(), host)
    scale = np.abs(want).max()
    assert np.abs(host - want).max() <= 1e-5 * scale, sslib.ss_last_kernel_name()
    rows, real_rows = oracle.stft_rows(p, n)
    assert host.shape[1] == rows and not host[:, real_rows:].any()
    one = ss.stft(x[1], sr, **kw)  # stft1
    assert one.shape == want.shape[1:]
    np.testing.assert_array_equal(one, host[1])


def _on_lab_test_a_lost_tile_hand_off_becomes_a_status(ss, sslib):
    """ss_mel_c1024<tile>: with wave 0's row pairs withheld (ss_debug_tile_fault) the waiting waves run into their bound,
    set the config's device error word and stop.  The launch ends, the host-pointer call returns SS_ERR_DEVICE, the
    device-pointer path reports it at ss_config_device_status and refuses further launches until it has been read; after
    that the same config computes correct results again."""
    import torch

    from speechsauce_amd import SpeechConfig, make_params

    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    cfg = SpeechConfig(make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128,
                                   high_frequency=8000.0))
    x = torch.from_numpy(_signal(5, (ncu + 3, 16000))).cuda()
    rows, _ = cfg.stft_rows(16000)
    out = torch.zeros((x.shape[0], 128, rows), dtype=torch.float32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        return sslib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), x.shape[0], 16000, 16000, out.data_ptr(), stream)

    sslib.ss_debug_mel_tile(2)  # the eight-wave builds: this batch takes the whole-line tile
    assert launch() == 0 and sslib.ss_last_kernel_name() == b"ss_mel_c1024<tile>"
    torch.cuda.synchronize()
    assert sslib.ss_config_device_status(cfg.handle) == 0
    good = out.clone()
    sslib.ss_debug_tile_fault(1)
    try:
        assert launch() == 0  # asynchronous: the launch itself succeeds
        torch.cuda.synchronize()  # ... and ends (bounded waits), it does not hang
        assert launch() == 6  # SS_ERR_DEVICE: no more work behind a broken launch (this call reads and clears the word)
        assert b"protocol error" in sslib.ss_last_error_string()
        assert launch() == 0
        torch.cuda.synchronize()
        assert sslib.ss_config_device_status(cfg.handle) == 6
        assert sslib.ss_config_device_status(cfg.handle) == 0  # cleared by the read
        # host-pointer entry point: synchronous, so the status comes back from the call itself
        xh = x.cpu().numpy()
        oh = np.empty((x.shape[0], 128, rows), np.float32)
        assert sslib.ss_mel_spectrogram(cfg.handle, xh.ctypes.data, x.shape[0], 16000, oh.ctypes.data) == 6
    finally:
        sslib.ss_debug_tile_fault(0)
        sslib.ss_debug_mel_tile(1)
    sslib.ss_debug_mel_tile(2)
    out.zero_()
    assert launch() == 0
    torch.cuda.synchronize()
    assert sslib.ss_config_device_status(cfg.handle) == 0
    assert torch.equal(out, good)
    sslib.ss_debug_mel_tile(1)


@pytest.mark.gpu
def test_a_lost_tile_hand_off_becomes_a_status(ss, sslab):
    """Runs on the LAB library (the build selection / fault aids it needs are not in the product library): the front is
    switched to it for the duration."""
    with ss._lib.use_library(sslab):
        _on_lab_test_a_lost_tile_hand_off_becomes_a_status(ss, sslab)


def _on_lab_test_mel_build_switch_is_bit_identical(ss, sslib):
    """ss_debug_mel_tile selects the build of the 2048-point mel kernel on one batch: the whole-line tile, eight waves with
    direct stores give the same bits, twelve waves the same values to f32 rounding."""
    import torch

    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    x = torch.from_numpy(_signal(9, (2 * ncu + 5, 16000))).cuda()
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    outs = {}
    try:
        for mode, name in ((2, b"ss_mel_c1024<tile>"), (0, b"ss_mel_c1024"), (3, b"ss_mel_c1024<w12,mel6321>")):
            sslib.ss_debug_mel_tile(mode)
            outs[mode] = ss.mel_spectrogram(x, 16000, **kw)
            assert sslib.ss_last_kernel_name() == name, (mode, sslib.ss_last_kernel_name())
        assert sslib.ss_debug_mel_tile(7) == 3  # SS_ERR_ARG
    finally:
        sslib.ss_debug_mel_tile(1)
    assert torch.equal(outs[2], outs[0])
    scale = outs[2].abs().amax(dim=(1, 2), keepdim=True)  # the twelve-wave kernel: same arithmetic, FMA fusion may differ in the last bit
    assert ((outs[3] - outs[2]).abs() <= 1e-6 * scale).all()


@pytest.mark.gpu
def test_mel_build_switch_is_bit_identical(ss, sslib, sslab):
    """The build switch is a LAB aid, so the builds are compared there; the PRODUCT library's own twelve-wave kernel (compiled
    without the lab switches) must then give the lab build's bits for the same batch."""
    import torch

    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    x = torch.from_numpy(_signal(9, (2 * ncu + 5, 16000))).cuda()
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    prod = ss.mel_spectrogram(x, 16000, **kw)
    assert sslib.ss_last_kernel_name() == b"ss_mel_c1024<w12,mel6321>"
    with ss._lib.use_library(sslab):
        _on_lab_test_mel_build_switch_is_bit_identical(ss, sslab)
        try:
            sslab.ss_debug_mel_tile(3)
            lab = ss.mel_spectrogram(x, 16000, **kw)
            assert sslab.ss_last_kernel_name() == b"ss_mel_c1024<w12,mel6321>"
        finally:
            sslab.ss_debug_mel_tile(1)
    assert torch.equal(prod, lab)


@pytest.mark.gpu
def test_cpp_mirror_stage_outputs(tmp_path, oracle):
    """The C++ mirror of the crate's API (include/speechsauce_amd.hpp), built with plain g++ against the library:
    stack_frames -> power_spectrum(frames), stft2 and stft1 on seeded clips, compared with the oracle."""
    import os
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    x = _signal(61, (2, 16000))
    x.tofile(tmp_path / "x.f32")
    src = tmp_path / "t.cpp"
    src.write_text(
        '#include "speechsauce_amd.hpp"\n#include <cstdio>\n'
        "template <class T> static void put(FILE *f, const std::vector<T> &v) { std::fwrite(v.data(), sizeof(T), v.size(), f); }\n"
        "int main(int argc, char **argv) {\n"
        "  if (argc < 3) return 1;\n"
        "  std::vector<float> x(2 * 16000); FILE *f = std::fopen(argv[1], \"rb\");\n"
        "  if (!f || std::fread(x.data(), 4, x.size(), f) != x.size()) return 2; std::fclose(f);\n"
        "  speechsauce::SpeechConfig cfg = speechsauce::SpeechConfigBuilder(16000).build();\n"
        # the reference's own argument lists (processing.rs:65-76, :179): no config in sight
        "  auto fr = speechsauce::stack_frames(x.data(), 16000, 16000, 0.02f, 0.01f, nullptr, false);\n"
        "  auto ps = speechsauce::power_spectrum(fr, 512);\n"
        # ... the config forms give the same bits, a window comes through `filter`, zero_padding adds the tail frames
        "  auto fr_c = speechsauce::stack_frames(x.data(), 16000, cfg);\n"
        "  auto ps_c = speechsauce::power_spectrum(fr, cfg);\n"
        "  if (fr_c.data != fr.data || ps_c.data != ps.data) return 5;\n"
        "  auto half = [](std::size_t n) { speechsauce::Array2 w{1, n, std::vector<float>(n, 0.5f)}; return w; };\n"
        "  auto fr_w = speechsauce::stack_frames(x, 16000, 0.02f, 0.01f, +half, false);\n"
        "  for (std::size_t i = 0; i < fr.data.size(); ++i) if (fr_w.data[i] != 0.5f * fr.data[i]) return 6;\n"
        "  std::vector<float> x1(x.begin(), x.begin() + 16000);\n"
        "  auto fr_p = speechsauce::stack_frames(x1, 16000, 0.02f, 0.02f, nullptr, true);\n"
        "  if (fr_p.rows != 49 || fr_p.cols != 320) return 7;\n"
        "  auto fr_44 = speechsauce::stack_frames(x1, 44100, 0.025f, 0.010f, nullptr, false);  // 1103-sample frames: no FFT length involved\n"
        "  if (fr_44.cols != 1103 || fr_44.rows != 33 || fr_44(3, 5) != x1[3 * 441 + 5]) return 8;\n"
        "  speechsauce::SpeechConfig sc = speechsauce::SpeechConfigBuilder(16000).fft_points(2048).frame_length(0.032f).frame_stride(0.032f).num_filters(128).build();\n"
        "  auto s2 = speechsauce::stft2(x.data(), 2, 16000, sc);\n"
        "  auto s1 = speechsauce::stft1(x.data() + 16000, 16000, sc);\n"
        "  if (fr.rows != 98 || fr.cols != 320 || ps.rows != 98 || ps.cols != 257) return 3;\n"
        "  if (s2.d0 != 2 || s2.d1 != 32 || s2.d2 != 1025 || s1.d0 != 1 || s1.d1 != 32) return 4;\n"
        "  f = std::fopen(argv[2], \"wb\"); put(f, fr.data); put(f, ps.data); put(f, s2.data); put(f, s1.data); std::fclose(f);\n"
        "  return 0; }\n")
    libdir = os.path.join(root, "mfcc-rust_amd", "lib")
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), str(src), "-L", libdir, "-lspeechsauce_amd",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    subprocess.run([str(exe), str(tmp_path / "x.f32"), str(tmp_path / "o.bin")], check=True)
    raw = (tmp_path / "o.bin").read_bytes()
    o = 0

    def take(count, dtype):
        nonlocal o
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=o)
        o += a.nbytes
        return a

    fr = take(98 * 320, np.float32).reshape(98, 320)
    ps = take(98 * 257, np.float32).reshape(98, 257)
    s2 = take(2 * 32 * 1025, np.complex64).reshape(2, 32, 1025)
    s1 = take(32 * 1025, np.complex64).reshape(32, 1025)
    assert o == len(raw)
    p = oracle.make_params()
    np.testing.assert_array_equal(fr.astype(np.float64), oracle.stack_frames(p, x[0]))
    want = oracle.power_spectrum_frames(fr, 512)
    assert np.abs(ps - want).max() <= 1e-5 * want.max()
    ps_sig = oracle.power_spectrum(p, x[0])  # the frames came from the signal: the fused form must agree as well
    assert np.abs(ps - ps_sig).max() <= 1e-5 * ps_sig.max()
    sp = oracle.make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128)
    ws = oracle.stft(sp, x)
    assert np.abs(s2 - ws).max() <= 1e-5 * np.abs(ws).max()
    np.testing.assert_array_equal(s1, s2[1])
    rows, real_rows = oracle.stft_rows(sp, 16000)
    assert rows == 32 and not s2[:, real_rows:].any()

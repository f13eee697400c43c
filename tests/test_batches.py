"""ss_mfcc_batches_device / ss_mel_spectrogram_batches_device: several independent batches per call.

Where the configuration runs on the 512-point MFCC kernel's default build, up to 8 batches share ONE persistent launch (its work range
is the concatenation of the batches' frame quads); everything else is served batch by batch.  Either way the contract is the one
tested here: the results are those of separate ss_mfcc_batch_device calls BIT FOR BIT (per clip: feature.rs:99-148), and sampled
clips match the oracle.
"""
import ctypes as C

import numpy as np
import pytest

from common import BENCH_KERNELS

pytestmark = pytest.mark.gpu

RTOL = 1e-4
MULTI_KERNEL = b"ss_mfcc_c256m<10,exact,bank421,sym>"      # 512-point MFCC, default shape (cfg2)
MULTI_KERNEL_4096 = b"ss_mfcc_c2048m<exact,mel8321,w12>"    # 4096-point MFCC, default shape (cfg5)
MULTI_KERNEL_MEL = b"ss_mel_c1024m<w12,mel6321>"            # 2048-point mel spectrogram, reference bank shape (cfg3)


def _rel(got, want):
    return float(np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def _tables(xs, outs):
    n = len(xs)
    px = (C.c_void_p * n)(*[(x.data_ptr() if x is not None and x.numel() else 0) for x in xs])
    po = (C.c_void_p * n)(*[(o.data_ptr() if o is not None and o.numel() else 0) for o in outs])
    nb = (C.c_size_t * n)(*[(0 if x is None else x.shape[0]) for x in xs])
    return px, nb, po


def _batches(torch, counts, n_samples, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return [torch.randn((c, n_samples), generator=g, device="cuda", dtype=torch.float32).mul_(0.03 + 0.02 * i) for i, c in enumerate(counts)]


def _separate(torch, sslib, cfg, xs, shape_tail, n_samples):
    outs = []
    for x in xs:
        o = torch.full((x.shape[0],) + shape_tail, float("nan"), device="cuda")
        if x.shape[0]:
            assert sslib.ss_mfcc_batch_device(cfg.handle, x.data_ptr(), x.shape[0], n_samples, n_samples, o.data_ptr(), None) == 0
        outs.append(o)
    return outs


def test_four_cfg2_batches_in_one_launch(ss, sslib, oracle):
    """4 x (1024 clips x 1 s): what bench.py's secondary.cfg2_x4 times.  One launch of the batch-table build, bit-identical to four
    launches of the bench kernel, sampled clips against the oracle."""
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    cfg = SpeechConfig(make_params(sample_rate=16000))
    xs = _batches(torch, [1024] * 4, 16000, 301)
    want = _separate(torch, sslib, cfg, xs, (98, 13), 16000)
    assert sslib.ss_last_kernel_name() == BENCH_KERNELS["cfg2"]
    got = [torch.full((1024, 98, 13), float("nan"), device="cuda") for _ in xs]
    px, nb, po = _tables(xs, got)
    assert sslib.ss_mfcc_batches_device(cfg.handle, 4, px, nb, 16000, 16000, po, None) == 0, sslib.ss_last_error_string()
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL, sslib.ss_last_kernel_name()
    torch.cuda.synchronize()
    cfg.device_status()
    p = oracle.make_params(sample_rate=16000)
    for b in range(4):
        assert torch.equal(got[b], want[b]), b
        for clip in (0, 511 + b, 1023):
            assert _rel(got[b][clip].cpu().numpy(), oracle.mfcc(p, xs[b][clip].cpu().numpy())) <= RTOL, (b, clip)
    # the Python front: a list in, a list out, the same launch
    outs = ss.mfcc_batch(xs, 16000)
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL
    assert isinstance(outs, list) and all(torch.equal(o, w) for o, w in zip(outs, want))


def test_ragged_batches_cross_quad_and_workgroup_boundaries(ss, sslib, oracle):
    """Clip counts whose frame totals are not multiples of four (98 frames per clip: a 1-clip batch ends in a half-filled quad), an
    empty batch in the middle, more than eight non-empty batches (two launches), batches smaller than one workgroup's share."""
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    cfg = SpeechConfig(make_params(sample_rate=16000))
    counts = [1, 3, 1024, 7, 0, 250, 5, 999, 13, 2, 61]
    xs = _batches(torch, counts, 16000, 302)
    want = _separate(torch, sslib, cfg, xs, (98, 13), 16000)
    got = [torch.full((c, 98, 13), float("nan"), device="cuda") for c in counts]
    px, nb, po = _tables(xs, got)
    assert sslib.ss_mfcc_batches_device(cfg.handle, len(xs), px, nb, 16000, 16000, po, None) == 0, sslib.ss_last_error_string()
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL
    torch.cuda.synchronize()
    p = oracle.make_params(sample_rate=16000)
    for b, c in enumerate(counts):
        assert torch.equal(got[b], want[b]), (b, c)
        if c:
            assert _rel(got[b][c - 1].cpu().numpy(), oracle.mfcc(p, xs[b][c - 1].cpu().numpy())) <= RTOL, b
    # eight one-clip batches: fewer workgroups than CUs, every batch boundary inside a workgroup's range
    xs1 = _batches(torch, [1] * 8, 16000, 303)
    want1 = _separate(torch, sslib, cfg, xs1, (98, 13), 16000)
    got1 = [torch.full((1, 98, 13), float("nan"), device="cuda") for _ in xs1]
    px, nb, po = _tables(xs1, got1)
    assert sslib.ss_mfcc_batches_device(cfg.handle, 8, px, nb, 16000, 16000, po, None) == 0
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(got1, want1))
    # clips longer than the default second, a padded leading dimension (views of a wider block)
    wide = _batches(torch, [33, 18], 24000, 304)
    views = [w[:, :20000] for w in wide]
    T = cfg.num_frames(20000)
    wantv = []
    for v in views:
        o = torch.empty((v.shape[0], T, 13), device="cuda")
        assert sslib.ss_mfcc_batch_device(cfg.handle, v.data_ptr(), v.shape[0], 20000, 24000, o.data_ptr(), None) == 0
        wantv.append(o)
    gotv = [torch.full((v.shape[0], T, 13), float("nan"), device="cuda") for v in views]
    px, nb, po = _tables(views, gotv)
    assert sslib.ss_mfcc_batches_device(cfg.handle, 2, px, nb, 20000, 24000, po, None) == 0
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(gotv, wantv))


def test_configurations_without_a_batch_table_are_served_batch_by_batch(ss, sslib, oracle):
    """A 4096-point configuration outside the default shape, a 1024-point one, a windowed 512-point one, clips of fewer than four
    frames, another filter count: the same call, the same results, one launch per batch of the kernel a single call would use."""
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    cases = [
        (dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=20, num_filters=128,
              high_frequency=22050.0), 44100, [40, 3, 17], b"ss_mfcc_c2048<"),
        (dict(sample_rate=16000, fft_points=1024, frame_length=0.05, frame_stride=0.02), 16000, [12, 7], b"ss_mfcc_c512"),
        (dict(sample_rate=16000, mfcc_window="hann"), 16000, [9, 130], b"ss_mfcc_c256<10,exact,bank421,win>"),
        (dict(sample_rate=16000), 800, [5, 2, 11], b"ss_mfcc_c256<"),  # 3 frames per clip: a quad spans clips, no batch-table build
        (dict(sample_rate=16000, num_filters=26), 16000, [6, 6], b"ss_mfcc_c256<"),
    ]
    for pkw, n, counts, kernel in cases:
        cfg = SpeechConfig(make_params(**pkw))
        T, Cc = cfg.num_frames(n), cfg.params.num_cepstral
        xs = _batches(torch, counts, n, 310)
        want = _separate(torch, sslib, cfg, xs, (T, Cc), n)
        single = sslib.ss_last_kernel_name()
        assert single.startswith(kernel), single
        got = [torch.full((c, T, Cc), float("nan"), device="cuda") for c in counts]
        px, nb, po = _tables(xs, got)
        assert sslib.ss_mfcc_batches_device(cfg.handle, len(xs), px, nb, n, n, po, None) == 0, sslib.ss_last_error_string()
        assert sslib.ss_last_kernel_name() == single
        torch.cuda.synchronize()
        assert all(torch.equal(g, w) for g, w in zip(got, want)), pkw
        okw = {k: v for k, v in pkw.items()}
        assert _rel(got[0][0].cpu().numpy(), oracle.mfcc(oracle.make_params(**okw), xs[0][0].cpu().numpy())) <= RTOL, pkw


def test_cfg5_batches_in_one_launch(ss, sslib, oracle):
    """The 4096-point MFCC kernel's default shape (cfg5) takes a batch table too: 4 x 512 clips (bench.py's secondary.cfg5_x4) and a
    ragged set with one-clip batches, bit-identical to separate launches of the bench kernel, sampled clips against the oracle."""
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    pkw = dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256,
               high_frequency=22050.0)
    cfg = SpeechConfig(make_params(**pkw))
    p = oracle.make_params(**pkw)
    for counts, seed in (([512] * 4, 350), ([1, 40, 3, 512, 0, 17, 1, 2, 130, 9, 5], 351)):
        xs = _batches(torch, counts, 44100, seed)
        want = _separate(torch, sslib, cfg, xs, (39, 40), 44100)
        assert sslib.ss_last_kernel_name() == BENCH_KERNELS["cfg5"]
        got = [torch.full((c, 39, 40), float("nan"), device="cuda") for c in counts]
        px, nb, po = _tables(xs, got)
        assert sslib.ss_mfcc_batches_device(cfg.handle, len(xs), px, nb, 44100, 44100, po, None) == 0, sslib.ss_last_error_string()
        assert sslib.ss_last_kernel_name() == MULTI_KERNEL_4096, sslib.ss_last_kernel_name()
        torch.cuda.synchronize()
        for b, c in enumerate(counts):
            assert torch.equal(got[b], want[b]), (b, c)
            if c:
                assert _rel(got[b][c - 1].cpu().numpy(), oracle.mfcc(p, xs[b][c - 1].cpu().numpy())) <= RTOL, b
    outs = ss.mfcc_batch(xs, 44100, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096)
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL_4096
    assert all(torch.equal(o, w) for o, w in zip(outs, want))


def test_random_batch_sets_through_the_three_batch_table_builds(ss, sslib):
    """Seeded random sets of 2..8 batches -- mostly tiny ones, so that a workgroup's contiguous unit range crosses several batch
    boundaries, a boundary falls on a workgroup's last unit, and launches run with fewer workgroups than CUs -- through each of the
    three batch-table builds: every block bit-identical to a launch of its own."""
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    rng = np.random.default_rng(6)
    cfg2 = SpeechConfig(make_params(sample_rate=16000))
    cfg5 = SpeechConfig(make_params(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40,
                                    num_filters=256, high_frequency=22050.0))
    cfg3 = SpeechConfig(make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128, high_frequency=8000.0))
    R = cfg3.stft_rows(16000)[0]
    mel_ok = [129, 150, 192, 257, 300, 333, 384, 400, 512]  # block sizes that select the twelve-wave build on their own
    for trial in range(10):
        nb = int(rng.integers(2, 9))
        small = [int(c) for c in rng.integers(1, 30, nb)]
        if trial % 3 == 0:
            small[int(rng.integers(0, nb))] = int(rng.integers(200, 700))
        for cfg, n, counts, tail, kernel in ((cfg2, 16000, small, (98, 13), MULTI_KERNEL), (cfg5, 44100, [max(1, c // 2) for c in small], (39, 40), MULTI_KERNEL_4096)):
            xs = _batches(torch, counts, n, 400 + trial)
            want = _separate(torch, sslib, cfg, xs, tail, n)
            got = [torch.full((c,) + tail, float("nan"), device="cuda") for c in counts]
            px, nbt, po = _tables(xs, got)
            assert sslib.ss_mfcc_batches_device(cfg.handle, len(xs), px, nbt, n, n, po, None) == 0, sslib.ss_last_error_string()
            assert sslib.ss_last_kernel_name() == kernel, (counts, sslib.ss_last_kernel_name())
            torch.cuda.synchronize()
            for b, c in enumerate(counts):
                assert torch.equal(got[b], want[b]), (trial, counts, b)
        counts = [mel_ok[int(i)] for i in rng.integers(0, len(mel_ok), nb)]
        xs = _batches(torch, counts, 16000, 500 + trial)
        want = []
        for x in xs:
            o = torch.empty((x.shape[0], 128, R), device="cuda")
            assert sslib.ss_mel_spectrogram_device(cfg3.handle, x.data_ptr(), x.shape[0], 16000, 16000, o.data_ptr(), None) == 0
            want.append(o)
        got = [torch.full((c, 128, R), float("nan"), device="cuda") for c in counts]
        px, nbt, po = _tables(xs, got)
        assert sslib.ss_mel_spectrogram_batches_device(cfg3.handle, len(xs), px, nbt, 16000, 16000, po, None) == 0, sslib.ss_last_error_string()
        assert sslib.ss_last_kernel_name() == MULTI_KERNEL_MEL, (counts, sslib.ss_last_kernel_name())
        torch.cuda.synchronize()
        for b, c in enumerate(counts):
            assert torch.equal(got[b], want[b]), (trial, counts, b)
    for c in (cfg2, cfg3, cfg5):
        c.device_status()


def test_batches_argument_errors_launch_nothing(ss, sslib):
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    cfg = SpeechConfig(make_params(sample_rate=16000))
    xs = _batches(torch, [4, 4], 16000, 320)
    got = [torch.full((4, 98, 13), float("nan"), device="cuda") for _ in xs]
    px, nb, po = _tables(xs, got)
    assert sslib.ss_mfcc_batches_device(cfg.handle, 0, None, None, 16000, 16000, None, None) == 0  # nothing to do
    assert sslib.ss_mfcc_batches_device(None, 2, px, nb, 16000, 16000, po, None) == 3
    assert sslib.ss_mfcc_batches_device(cfg.handle, 2, None, nb, 16000, 16000, po, None) == 3
    assert sslib.ss_mfcc_batches_device(cfg.handle, 2, px, nb, 16000, 15999, po, None) == 3  # ld < n_samples
    # fewer samples than one frame: the reference underflows and panics (processing.rs:101)
    assert sslib.ss_mfcc_batches_device(cfg.handle, 2, px, nb, 100, 16000, po, None) == 1
    assert sslib.ss_last_error_string()
    bad = (C.c_void_p * 2)(xs[0].data_ptr(), 0)  # a null input block in a non-empty batch
    assert sslib.ss_mfcc_batches_device(cfg.handle, 2, bad, nb, 16000, 16000, po, None) == 3
    assert b"batch 1" in sslib.ss_last_error_string()
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(g).all()) for g in got)  # none of the failing calls wrote anything
    zero = (C.c_size_t * 2)(0, 0)  # only empty batches: nothing to do, null blocks are fine
    assert sslib.ss_mfcc_batches_device(cfg.handle, 2, (C.c_void_p * 2)(0, 0), zero, 16000, 16000, (C.c_void_p * 2)(0, 0), None) == 0
    with pytest.raises(ValueError):
        ss.mfcc_batch([xs[0], xs[1][:, :8000]], 16000)
    assert ss.mfcc_batch([], 16000) == []


def test_mel_spectrogram_batches(ss, sslib, oracle):
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    pkw = dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128, high_frequency=8000.0)
    cfg = SpeechConfig(make_params(**pkw))
    R = cfg.stft_rows(16000)[0]
    counts = [1024, 0, 5, 300]
    xs = _batches(torch, counts, 16000, 330)
    want = []
    for x in xs:
        o = torch.empty((x.shape[0], 128, R), device="cuda")
        if x.shape[0]:
            assert sslib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), x.shape[0], 16000, 16000, o.data_ptr(), None) == 0
        want.append(o)
    got = [torch.full((c, 128, R), float("nan"), device="cuda") for c in counts]
    px, nb, po = _tables(xs, got)
    assert sslib.ss_mel_spectrogram_batches_device(cfg.handle, 4, px, nb, 16000, 16000, po, None) == 0, sslib.ss_last_error_string()
    torch.cuda.synchronize()
    cfg.device_status()
    assert all(torch.equal(g, w) for g, w in zip(got, want))
    # (the 5-channel block alone runs on the eight-wave build, which rounds a few FMAs differently: a group with such a block is served
    # block by block, so that the results never depend on how the blocks were grouped)
    assert sslib.ss_last_kernel_name().startswith(b"ss_mel_c1024<"), sslib.ss_last_kernel_name()
    p = oracle.make_params(**pkw)
    assert _rel(got[3][299].cpu().numpy(), oracle.mel_spectrogram(p, xs[3][299].cpu().numpy()[None, :])[0]) <= RTOL
    assert sslib.ss_mel_spectrogram_batches_device(cfg.handle, 4, px, nb, 16000, 100, po, None) == 3
    # blocks that each select the twelve-wave build share ONE launch of its batch-table build (4 x 1024: bench.py's secondary.cfg3_x4)
    for counts2, seed in (([1024] * 4, 331), ([1024, 300, 0, 512, 150, 129, 1000, 700, 400, 350, 333], 332)):
        xs2 = _batches(torch, counts2, 16000, seed)
        want2 = []
        for x in xs2:
            o = torch.empty((x.shape[0], 128, R), device="cuda")
            if x.shape[0]:
                assert sslib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), x.shape[0], 16000, 16000, o.data_ptr(), None) == 0
                assert sslib.ss_last_kernel_name() == BENCH_KERNELS["cfg3"], (x.shape[0], sslib.ss_last_kernel_name())
            want2.append(o)
        got2 = [torch.full((c, 128, R), float("nan"), device="cuda") for c in counts2]
        px, nb, po = _tables(xs2, got2)
        assert sslib.ss_mel_spectrogram_batches_device(cfg.handle, len(xs2), px, nb, 16000, 16000, po, None) == 0, sslib.ss_last_error_string()
        assert sslib.ss_last_kernel_name() == MULTI_KERNEL_MEL, sslib.ss_last_kernel_name()
        torch.cuda.synchronize()
        cfg.device_status()
        for b, c in enumerate(counts2):
            assert torch.equal(got2[b], want2[b]), (b, c)
        b, c = len(counts2) - 1, counts2[-1]
        assert _rel(got2[b][c - 1].cpu().numpy(), oracle.mel_spectrogram(p, xs2[b][c - 1].cpu().numpy()[None, :])[0]) <= RTOL
    # the Python front: a list in, a list out
    kw = dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    outs = ss.mel_spectrogram([x for x in xs2 if x.shape[0]], 16000, **kw)
    assert sslib.ss_last_kernel_name() == MULTI_KERNEL_MEL
    assert all(torch.equal(o, w) for o, w in zip(outs, [w for w, c in zip(want2, counts2) if c]))


def test_timed_region_times_and_clocks_the_same_launches(ss, sslib):
    """ss_mfcc_timed_region (bench.py's secondary.cfg2): events and per-wave stamps over the same launches; the outputs are the
    ordinary ones."""
    import torch
    from speechsauce_amd import SpeechConfig, make_params

    cfg = SpeechConfig(make_params(sample_rate=16000))
    xs = _batches(torch, [1024] * 5, 16000, 340)
    want = _separate(torch, sslib, cfg, xs, (98, 13), 16000)
    outs = [torch.full((1024, 98, 13), float("nan"), device="cuda") for _ in range(5)]
    px, _, po = _tables(xs, outs)
    ms, ghz, wall = C.c_float(0), C.c_float(0), C.c_float(0)
    st = torch.cuda.Stream()
    assert sslib.ss_mfcc_timed_region(cfg.handle, px, 5, 1024, 16000, 16000, po, 5, C.c_void_p(st.cuda_stream), 200, 64,
                                      C.byref(ms), C.byref(ghz), C.byref(wall)) == 0, sslib.ss_last_error_string()
    assert sslib.ss_last_kernel_name() == BENCH_KERNELS["cfg2"]
    assert 0.015 < ms.value < 0.08 and 1.0 < ghz.value < 2.6, (ms.value, ghz.value)
    assert 200 * ms.value <= wall.value < 200 * ms.value + 2.0, (ms.value, wall.value)  # the host saw the same region, plus its own latency
    assert all(torch.equal(o, w) for o, w in zip(outs, want))
    # timing only
    assert sslib.ss_mfcc_timed_region(cfg.handle, px, 5, 1024, 16000, 16000, po, 5, None, 20, 0, C.byref(ms), C.byref(ghz), None) == 0
    assert ms.value > 0 and ghz.value == 0
    # the 4096-point MFCC kernel's and the 2048-point mel kernel's twelve-wave builds stamp their waves too (secondary.cfg5 / cfg3)
    c5 = SpeechConfig(make_params(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40,
                                  num_filters=256, high_frequency=22050.0))
    x5 = _batches(torch, [512, 512], 44100, 341)
    w5 = _separate(torch, sslib, c5, x5, (39, 40), 44100)
    g5 = [torch.full((512, 39, 40), float("nan"), device="cuda") for _ in range(2)]
    px5, _, po5 = _tables(x5, g5)
    assert sslib.ss_mfcc_timed_region(c5.handle, px5, 2, 512, 44100, 44100, po5, 2, None, 60, 60, C.byref(ms), C.byref(ghz), C.byref(wall)) == 0, sslib.ss_last_error_string()
    assert sslib.ss_last_kernel_name() == BENCH_KERNELS["cfg5"] and 0.03 < ms.value < 0.15 and 1.0 < ghz.value < 2.6, (ms.value, ghz.value)
    assert all(torch.equal(o, w) for o, w in zip(g5, w5))
    c3 = SpeechConfig(make_params(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128, high_frequency=8000.0))
    R = c3.stft_rows(16000)[0]
    g3 = [torch.full((1024, 128, R), float("nan"), device="cuda") for _ in range(2)]
    po3 = (C.c_void_p * 2)(*[o.data_ptr() for o in g3])
    assert sslib.ss_mel_spectrogram_timed_region(c3.handle, px, 5, 1024, 16000, 16000, po3, 2, None, 60, 60, C.byref(ms), C.byref(ghz), C.byref(wall)) == 0, sslib.ss_last_error_string()
    assert sslib.ss_last_kernel_name() == BENCH_KERNELS["cfg3"] and 0.02 < ms.value < 0.12 and 1.0 < ghz.value < 2.6, (ms.value, ghz.value)
    w3 = torch.empty((1024, 128, R), device="cuda")
    assert sslib.ss_mel_spectrogram_device(c3.handle, xs[59 % 5].data_ptr(), 1024, 16000, 16000, w3.data_ptr(), None) == 0
    assert torch.equal(g3[59 % 2], w3)  # launch 59 read batch 59 % 5 and wrote block 59 % 2
    assert sslib.ss_mel_spectrogram_timed_region(c3.handle, px, 0, 1024, 16000, 16000, po3, 2, None, 6, 0, C.byref(ms), C.byref(ghz), None) == 3
    # a kernel without stamps: timed, then SS_ERR_UNSUPPORTED
    cfg5 = SpeechConfig(make_params(sample_rate=16000, fft_points=1024))
    o5 = torch.empty((8, cfg5.num_frames(16000), 13), device="cuda")
    p5, q5 = (C.c_void_p * 1)(xs[0].data_ptr()), (C.c_void_p * 1)(o5.data_ptr())
    assert sslib.ss_mfcc_timed_region(cfg5.handle, p5, 1, 8, 16000, 16000, q5, 1, None, 4, 2, C.byref(ms), C.byref(ghz), None) == 5
    assert sslib.ss_mfcc_timed_region(cfg.handle, px, 0, 1024, 16000, 16000, po, 5, None, 20, 0, C.byref(ms), C.byref(ghz), None) == 3

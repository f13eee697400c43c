#!/usr/bin/env python3
"""Times the stage outputs of one workload (power spectrum / mfe / mfcc, or stft / mel) to see where a kernel spends time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from speechsauce_amd import SpeechConfig, _lib, make_params  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
desc, pkw, n, clips, kind = WORKLOADS[name]
cfg = SpeechConfig(make_params(**pkw))
lib = _lib.lib()
x = torch.randn((clips, n), device="cuda") * 0.1
F = cfg.params.fft_points // 2 + 1
M, Cc = cfg.params.num_filters, cfg.params.num_cepstral


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


if kind == "mfcc":
    T = cfg.num_frames(n)
    P = torch.empty((clips, T, F), device="cuda")
    feat = torch.empty((clips, T, M), device="cuda")
    en = torch.empty((clips, T), device="cuda")
    out = torch.empty((clips, T, Cc), device="cuda")
    print(name, "power  %.1f us" % timeit(lambda: _lib.check(lib.ss_power_spectrum_batch_device(cfg.handle, x.data_ptr(), clips, n, n, P.data_ptr(), None))))
    print(name, "mfe    %.1f us" % timeit(lambda: _lib.check(lib.ss_mfe_batch_device(cfg.handle, x.data_ptr(), clips, n, n, feat.data_ptr(), en.data_ptr(), None))))
    print(name, "mfcc   %.1f us" % timeit(lambda: _lib.check(lib.ss_mfcc_batch_device(cfg.handle, x.data_ptr(), clips, n, n, out.data_ptr(), None))))
else:
    R, _ = cfg.stft_rows(n)
    S = torch.empty((clips, R, F, 2), device="cuda")
    out = torch.empty((clips, M, R), device="cuda")
    print(name, "stft   %.1f us" % timeit(lambda: _lib.check(lib.ss_stft_device(cfg.handle, x.data_ptr(), clips, n, n, S.data_ptr(), None))))
    print(name, "mel    %.1f us" % timeit(lambda: _lib.check(lib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), clips, n, n, out.data_ptr(), None))))

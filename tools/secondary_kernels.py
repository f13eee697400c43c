"""Runs every dedicated kernel other than the three BASELINE ones on its measurement configuration, 200 launches
each with rotating inputs, so that one `rocprofv3 --kernel-trace --stats` pass records their average durations."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss
from speechsauce_amd import _lib

LIB = dict(framing="center", pad_mode="reflect", mfcc_window="hann", spectrum_exponent=2, mel_scale="slaney", mel_norm="slaney", dct_norm="ortho")
CASES = [
    ("mfcc", 8000, 16000, 1024, dict(fft_length=256), {}),
    ("mfcc", 16000, 16000, 1024, dict(frame_length=0.025, num_filters=80), {}),
    ("mfe", 16000, 16000, 1024, dict(frame_length=512 / 16000, num_filters=80), {k: v for k, v in LIB.items() if k != "dct_norm"}),
    ("mfcc", 22050, 22050, 512, dict(frame_length=1024 / 22050, frame_stride=256 / 22050, num_cepstral=20, num_filters=64, fft_length=1024), {}),
    ("mfcc", 44100, 44100, 512, dict(frame_length=2048 / 44100, frame_stride=512 / 44100, num_cepstral=20, num_filters=128, fft_length=2048), {}),
    ("mfcc", 44100, 44100, 512, dict(frame_length=2048 / 44100, frame_stride=512 / 44100, num_cepstral=20, num_filters=128, fft_length=2048), LIB),
    ("mel", 16000, 16000, 1024, dict(frame_length=0.016, frame_stride=0.016, num_filters=40, fft_length=512), {}),
    ("mel", 16000, 16000, 1024, dict(frame_length=0.032, frame_stride=0.032, num_filters=80, fft_length=1024), {}),
    ("mel", 44100, 44100, 512, dict(frame_length=1024 / 44100, frame_stride=1024 / 44100, num_filters=256, fft_length=4096), {}),
]
lib = _lib.lib()
for kind, sr, n, clips, kw, sw in CASES:
    xs = [torch.randn((clips, n), device="cuda") * 0.1 for _ in range(max(2, 300 * 2 ** 20 // (4 * clips * n) + 1))]
    fn = {"mfcc": ss.mfcc_batch, "mfe": ss.mfe_batch, "mel": ss.mel_spectrogram}[kind]
    for i in range(220):
        fn(xs[i % len(xs)], sr, **kw, **sw)
    torch.cuda.synchronize()
    print(kind, sr, kw.get("fft_length", 512), lib.ss_last_kernel_name().decode())

"""stft2 output (functions.rs:86-123) of the stft builds of the mel kernels: time per launch and HBM rate.
usage: stft_rate.py [fft_points sample_rate hop clips]   (default: the cfg3 batch, 2048 points)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss
from speechsauce_amd import SpeechConfig, make_params, _lib

nfft, sr, hop, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (2048, 16000, 512, 1024)))
cfg = SpeechConfig(make_params(sample_rate=sr, fft_points=nfft, frame_length=hop / sr, frame_stride=hop / sr, num_filters=128, high_frequency=sr / 2))
lib = _lib.lib()
L = sr
F = nfft // 2 + 1
R, _ = cfg.stft_rows(L)
xs = [torch.randn((B, L), device="cuda") * 0.1 for _ in range(5)]
out = torch.empty((B, R, F, 2), device="cuda")
def run(i):
    _lib.check(lib.ss_stft_device(cfg.handle, xs[i % 5].data_ptr(), B, L, L, out.data_ptr(), None))
for i in range(20): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 300
for i in range(n): run(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byt = 4 * B * L + out.numel() * 4
print(lib.ss_last_kernel_name().decode(), f"{us:.1f} us per launch, {byt/1e6:.1f} MB algorithmic -> {byt/us/1e6:.2f} TB/s ({byt/us/1e6/8:.3f} of 8 TB/s)")

"""stft on the cfg3 shape against the batch size: where the output crosses the 256 MiB Infinity Cache."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "mfcc-rust_amd"))
import torch

import speechsauce_amd as ss

CLIPS = [int(c) for c in sys.argv[1:]] or [256, 512, 768, 896, 960, 1024, 1280, 1536, 2048, 4096]
for clips in CLIPS:
    x = torch.randn(clips, 16000, device="cuda") * 0.1
    fn = lambda: ss.stft(x, 16000, frame_length=0.032, fft_length=2048)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    out_mb = clips * 32 * 8200 / 1e6
    print("%5d clips: %7.1f us  %6.1f ns/clip  output %6.1f MB  %.2f TB/s (in + out)" % (clips, us, us * 1e3 / clips, out_mb, (out_mb + clips * 0.064) / us))
    del x

"""Rates of the stage-output entry points (device forms) on the cfg2 shape: stack_frames, power_spectrum of a frames matrix,
power_spectrum of the signal (fused), stft (cfg3 shape)."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "mfcc-rust_amd"))
import torch

import speechsauce_amd as ss


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


lib = ss._lib.lib()
x = torch.randn(1024, 16000, device="cuda") * 0.1
x1 = x.reshape(-1)  # stack_frames / power_spectrum take one signal (processing.rs:65, :179): 1024 s of audio as one clip
us = timed(lambda: ss.stack_frames(x1, 16000, frame_length=0.02, frame_stride=0.01))
fr = ss.stack_frames(x1, 16000, frame_length=0.02, frame_stride=0.01)
print("stack_frames [16 384 000] -> %s: %.1f us, %.2f TB/s written, kernel %s" % (tuple(fr.shape), us, fr.numel() * 4 / us / 1e6, lib.ss_last_kernel_name().decode()))
frames = fr.reshape(-1, fr.shape[-1])
us = timed(lambda: ss.power_spectrum(frames, 512))
print("power_spectrum(frames [%d x %d], 512): %.1f us, kernel %s" % (frames.shape[0], frames.shape[1], us, lib.ss_last_kernel_name().decode()))
us = timed(lambda: ss.power_spectrum_of_signal(x1, 16000))
print("power_spectrum_of_signal [16 384 000]: %.1f us, kernel %s" % (us, lib.ss_last_kernel_name().decode()))
us = timed(lambda: ss.stft(x, 16000, frame_length=0.032, fft_length=2048))
print("stft [1024 x 16000], n_fft 2048 hop 512: %.1f us, kernel %s" % (us, lib.ss_last_kernel_name().decode()))

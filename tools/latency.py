"""Single-utterance latency of the host-pointer API (the reference's mfcc(signal)): per call from a numpy array through the
Python front, and for a clip already resident on the device.  SS_HOST_SMALL_KB=0 disables the mapped-staging path (A/B)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'mfcc-rust_amd'))
import numpy as np
import torch
import speechsauce_amd as ss

for secs in (1, 3):
    x = (np.random.default_rng(0).standard_normal(16000 * secs) * 0.1).astype(np.float32)
    for _ in range(50):
        ss.mfcc(x, 16000)
    t0 = time.perf_counter()
    for _ in range(1000):
        ss.mfcc(x, 16000)
    print("host, one %d s clip: %.1f us per mfcc() call" % (secs, (time.perf_counter() - t0) / 1000 * 1e6))
    for _ in range(50):
        ss.mel_spectrogram(x, 16000, frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    t0 = time.perf_counter()
    for _ in range(1000):
        ss.mel_spectrogram(x, 16000, frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
    print("host, one %d s clip: %.1f us per mel_spectrogram() call (n_fft 2048, 128 mels)" % (secs, (time.perf_counter() - t0) / 1000 * 1e6))
    xd = torch.from_numpy(x[None]).cuda()
    for _ in range(50):
        ss.mfcc_batch(xd, 16000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        o = ss.mfcc_batch(xd, 16000)
    torch.cuda.synchronize()
    print("device-resident %d s clip: %.1f us per mfcc_batch() call" % (secs, (time.perf_counter() - t0) / 1000 * 1e6))

#!/usr/bin/env python3
"""Per-phase cycle sums of the 4096-point MFCC kernel (lab build with -DSS_PROF5=1, SS_DEBUG_ROWS=<file>): runs one cfg5
batch and prints, per phase, the share of the waves' main-loop lifetime and the cycles per frame."""
import os, sys
out = "/tmp/prof5.bin"
os.environ["SS_DEBUG_ROWS"] = out
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
import numpy as np, torch
import speechsauce_amd as ss
kw = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096, high_frequency=22050.0)
x = (torch.randn((512, 44100), device="cuda") * 0.1)
for _ in range(3):
    ss.mfcc_batch(x, 44100, **kw)  # only the FIRST launch of the process is stamped (static flag): make it this one
torch.cuda.synchronize()
w = np.fromfile(out, dtype=np.uint64).reshape(-1, 16)
w = w[w[:, 0] > 0]
names = ["iters", "claim", "loads", "pass1", "exchange", "twiddle", "pass2", "radix2", "untangle", "mel+ln", "dct+store", "loop"]
it = w[:, 0].astype(float)
print("waves", len(w), "frames/wave mean", it.mean(), "loop cycles/frame", (w[:, 11] / it).mean())
tot = w[:, 11].astype(float).sum()
for k in range(1, 11):
    print(f"{names[k]:10s} share {w[:, k].astype(float).sum() / tot:6.3f}   cycles/frame {(w[:, k] / it).mean():8.0f}")

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
# by wave index within the workgroup (12 waves; 0-3 dispatched first = the oldest wave of each SIMD): frames done, cycles per
# frame, main-loop lifetime -- the three speed classes of profiles/r05/unit_timeline_cfg3*.txt, and what the tail looks like
w12 = w[: (len(w) // 12) * 12].reshape(-1, 12, 16).astype(float)
print("by wave index: frames        ", [round(w12[:, k, 0].mean(), 2) for k in range(12)])
print("by wave index: cycles/frame  ", [int(round((w12[:, k, 11] / w12[:, k, 0]).mean())) for k in range(12)])
print("by wave index: loop k cycles ", [round(w12[:, k, 11].mean() / 1e3, 1) for k in range(12)])
print("slowest wave of a CU is index", np.bincount(w12[:, :, 11].argmax(axis=1), minlength=12).tolist(),
      "| CU loop k cycles p10 p50 p90 max", [round(float(x) / 1e3, 1) for x in np.percentile(w12[:, :, 11].max(axis=1), [10, 50, 90, 100])],
      "| sum of wave loops / 12:", round(w12[:, :, 11].sum(axis=1).mean() / 12e3, 1))

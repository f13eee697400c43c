#!/usr/bin/env python3
"""Per-phase cycle sums of the 512-point MFCC kernel (lab build with -DSS_PROF2=1, loaded through SS_LIB_PATH): runs cfg2
batches with a 16-words-per-wave stamp buffer and prints, per phase, the share of the waves' main-loop lifetime and the cycles
per quad."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
import numpy as np, torch
import speechsauce_amd as ss
from speechsauce_amd import _lib
lib = _lib.lib()
xs = [torch.randn((1024, 16000), device="cuda") * 0.1 for _ in range(5)]
ncu = torch.cuda.get_device_properties(0).multi_processor_count
st = torch.zeros((ncu * 16, 16), dtype=torch.int64, device="cuda")
for x in xs:
    ss.mfcc_batch(x, 16000)
torch.cuda.synchronize()
lib.ss_debug_stamp_buffer(st.data_ptr())
ss.mfcc_batch(xs[0], 16000)
torch.cuda.synchronize()
lib.ss_debug_stamp_buffer(None)
w = st.cpu().numpy()
w = w[w[:, 0] > 0]
names = ["iters", "claim", "samples", "pass1+xw", "xr+twid", "pass2+bperm", "untangle", "mel", "ln+dct", "store", "-", "loop"]
it = w[:, 0].astype(float)
print("waves", len(w), "quads/wave mean", it.mean(), "loop cycles/quad", (w[:, 11] / it).mean())
tot = w[:, 11].astype(float).sum()
for k in range(1, 10):
    print(f"{names[k]:12s} share {w[:, k].astype(float).sum() / tot:6.3f}   cycles/quad {(w[:, k] / it).mean():8.0f}")
# by wave index within the workgroup (16 stamp rows per workgroup, 12 waves used; waves 0-3 are dispatched first = the oldest wave
# of each SIMD): quads done, cycles per quad, main-loop lifetime -- the three speed classes of profiles/r05/unit_timeline_cfg3*.txt
full = st.cpu().numpy().reshape(ncu, 16, 16).astype(float)
full = full[full[:, :12, 0].min(axis=1) > 0][:, :12, :]
print("by wave index: quads         ", [round(float(full[:, k, 0].mean()), 2) for k in range(12)])
print("by wave index: cycles/quad   ", [int(round(float((full[:, k, 11] / full[:, k, 0]).mean()))) for k in range(12)])
print("by wave index: loop k cycles ", [round(float(full[:, k, 11].mean()) / 1e3, 1) for k in range(12)])
print("slowest wave of a CU is index", np.bincount(full[:, :, 11].argmax(axis=1), minlength=12).tolist(),
      "| CU loop k cycles p10 p50 p90 max", [round(float(x) / 1e3, 1) for x in np.percentile(full[:, :, 11].max(axis=1), [10, 50, 90, 100])],
      "| sum of wave loops / 12:", round(float(full[:, :, 11].sum(axis=1).mean()) / 12e3, 1))

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

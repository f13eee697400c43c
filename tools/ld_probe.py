"""Does the clips' leading dimension (the distance between the CUs' input ranges) matter?  cfg2 shape on views of padded
batches: same work, different address strides between the workgroups' ranges (HBM channel aliasing would show here)."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "mfcc-rust_amd"))
import torch

import speechsauce_amd as ss

B, L = 1024, 16000
for ld in (16000, 16016, 16064, 16128, 16384, 16448, 17000, 20000):
    bufs = [torch.randn(B, ld, device="cuda") * 0.1 for _ in range(5)]
    views = [b[:, :L] for b in bufs]
    for v in views:
        ss.mfcc_batch(v, 16000)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(1000):
            ss.mfcc_batch(views[i % 5], 16000)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1))
    print("ld", ld, "us per launch", [round(t, 2) for t in best], ss._lib.lib().ss_last_kernel_name().decode(), flush=True)
    del bufs, views

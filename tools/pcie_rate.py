"""Host-buffer entry point (ss_mfcc_batch: hipMalloc + H2D + kernel + D2H per call) vs the device-resident rate, cfg2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss

x = (np.random.default_rng(1).standard_normal((1024, 16000)) * 0.1).astype(np.float32)
xp = torch.from_numpy(x).pin_memory().numpy()
for name, arr in (("pageable", x), ("pinned", xp)):
    ss.mfcc_batch(arr, 16000)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = ss.mfcc_batch(arr, 16000)
    dt = (time.perf_counter() - t0) / n
    print(f"host {name}: {dt*1e3:.2f} ms per 1024-clip call = {out.shape[0]*out.shape[1]/dt:.3e} frames/s, {x.nbytes/dt/1e9:.1f} GB/s in")
xd = torch.from_numpy(x).cuda()
ss.mfcc_batch(xd, 16000); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    o = ss.mfcc_batch(xd, 16000)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print(f"device-resident (python call incl.): {dt*1e6:.1f} us per call = {o.shape[0]*o.shape[1]/dt:.3e} frames/s")

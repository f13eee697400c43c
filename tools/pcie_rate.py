"""Host-buffer entry point (ss_mfcc_batch: chunked H2D / kernel / D2H on two private streams, cached device buffers) against
the device-resident rate, cfg2.  Three callers: the Python front on a pageable numpy array (allocates its result per call),
the same on pinned memory, and the C ABI called directly with pinned input AND a reused pinned output (what a Rust / C++
service with its own buffers sees)."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss
from speechsauce_amd import SpeechConfig, _lib, make_params

x = (np.random.default_rng(1).standard_normal((1024, 16000)) * 0.1).astype(np.float32)
xp = torch.from_numpy(x).pin_memory().numpy()
ref = None
for name, arr in (("pageable numpy, Python front", x), ("pinned, Python front", xp)):
    ss.mfcc_batch(arr, 16000)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = ss.mfcc_batch(arr, 16000)
    dt = (time.perf_counter() - t0) / n
    ref = out
    print(f"host {name}: {dt*1e3:.2f} ms per 1024-clip call = {out.shape[0]*out.shape[1]/dt:.3e} frames/s, {x.nbytes/dt/1e9:.1f} GB/s in")
cfg = SpeechConfig(make_params(sample_rate=16000))
lib = _lib.lib()
outp = torch.empty((1024, 98, 13), dtype=torch.float32).pin_memory()
for name, src in (("pageable in, pinned out, C ABI", x), ("pinned in + out, C ABI", xp)):
    _lib.check(lib.ss_mfcc_batch(cfg.handle, src.ctypes.data, 1024, 16000, 16000, outp.data_ptr()))
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        _lib.check(lib.ss_mfcc_batch(cfg.handle, src.ctypes.data, 1024, 16000, 16000, outp.data_ptr()))
    dt = (time.perf_counter() - t0) / n
    assert np.array_equal(outp.numpy(), ref)
    print(f"host {name}: {dt*1e3:.2f} ms per 1024-clip call = {1024*98/dt:.3e} frames/s, {x.nbytes/dt/1e9:.1f} GB/s in")
xd = torch.from_numpy(x).cuda()
o = ss.mfcc_batch(xd, 16000); torch.cuda.synchronize()
assert np.array_equal(o.cpu().numpy(), ref), "host and device paths must give the same bits"
t0 = time.perf_counter()
for _ in range(200):
    o = ss.mfcc_batch(xd, 16000)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print(f"device-resident (python call incl.): {dt*1e6:.1f} us per call = {o.shape[0]*o.shape[1]/dt:.3e} frames/s")

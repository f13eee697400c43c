"""Experiment: do dword-aligned (not 8-byte aligned) sample-pair loads work, and what do they cost?  Runs the frame-path kernels
on odd hops / odd leading dimensions / odd base offsets with SS_ALLOW_UNALIGNED=1 and compares with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import numpy as np, torch
import speechsauce_amd as ss
import oracle_c as oracle
lib = ss._lib.lib()
def rel(g, w): return float(np.abs(g - w).max() / np.abs(w).max())
for sr, nfft, flen, step, M, C in ((16000, 512, 320, 161, 40, 13), (16000, 512, 400, 161, 80, 13), (22050, 1024, 882, 221, 64, 20),
                                   (44100, 2048, 1764, 441, 128, 20), (44100, 4096, 4096, 1023, 256, 40)):
    n = sr + 1
    x = (np.random.default_rng(3).standard_normal((5, n + 1)) * 0.1).astype(np.float32)
    xd = torch.from_numpy(x).cuda()[:, 1:]            # odd base offset, odd leading dimension
    kw = dict(frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
    p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=step / sr, num_cepstral=C, num_filters=M)
    got = ss.mfcc_batch(xd, sr, **kw).cpu().numpy()
    print(nfft, lib.ss_last_kernel_name().decode(), "rel err", max(rel(got[b], oracle.mfcc(p, x[b, 1:])) for b in range(5)))
# cost: cfg2-sized batch with an odd hop vs an even one
for step in (160, 161):
    xs = [torch.randn((1024, 16001), device="cuda")[:, 1:] * 0.1 for _ in range(5)] if step == 161 else [torch.randn((1024, 16000), device="cuda") * 0.1 for _ in range(5)]
    kw = dict(frame_stride=step / 16000)
    for i in range(20): ss.mfcc_batch(xs[i % 5], 16000, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): o = ss.mfcc_batch(xs[i % 5], 16000, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print("hop", step, lib.ss_last_kernel_name().decode(), f"{dt*1e6:.1f} us per call (python incl.), {o.shape[1]} frames/clip")

"""Timing of the post-processing kernels on the cfg2 feature block [1024, 98, 13] (device-resident)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss

x = torch.randn((1024, 16000), device="cuda") * 0.1
f = ss.mfcc_batch(x, 16000)
def t(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:32s} {e0.elapsed_time(e1) * 1e3 / n:8.1f} us")
t("mfcc_batch", lambda: ss.mfcc_batch(x, 16000))
t("cmvn(var)", lambda: ss.cmvn(f, True))
t("cmvnw(301)", lambda: ss.cmvnw(f, 301, False))
t("cmvnw(301, var)", lambda: ss.cmvnw(f, 301, True))
t("cmvnw(31, var)", lambda: ss.cmvnw(f, 31, True))
t("derivative_extraction(2)", lambda: ss.derivative_extraction(f, 2))
t("extract_derivative_feature", lambda: ss.extract_derivative_feature(f))

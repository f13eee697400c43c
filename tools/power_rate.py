"""power_spectrum output (processing.rs:179-181) of the 512-point kernel on the cfg2 batch: time per launch and HBM rate."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss
from speechsauce_amd import SpeechConfig, make_params, _lib

cfg = SpeechConfig(make_params(sample_rate=16000))
lib = _lib.lib()
B, L = 1024, 16000
T = cfg.num_frames(L)
xs = [torch.randn((B, L), device="cuda") * 0.1 for _ in range(5)]
out = torch.empty((B, T, 257), device="cuda")
def run(i):
    _lib.check(lib.ss_power_spectrum_batch_device(cfg.handle, xs[i % 5].data_ptr(), B, L, L, out.data_ptr(), None))
for i in range(20): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 300
for i in range(n): run(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byt = 4 * B * L + out.numel() * 4
print(lib.ss_last_kernel_name().decode(), f"{us:.1f} us per launch, {byt/1e6:.1f} MB algorithmic -> {byt/us/1e6:.2f} TB/s ({byt/us/1e6/8:.3f} of 8 TB/s)")

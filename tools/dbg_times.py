#!/usr/bin/env python3
"""Runs one cfg2 batch with SS_DEBUG_TIMES set (per-wave stamps of one cold launch) and prints tools/wave_times.py's summary."""
import os, subprocess, sys
# SS_DEBUG_TIMES exists in the LAB build of the library only (`make -C mfcc-rust_amd/csrc lab`)
os.environ.setdefault("SS_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mfcc-rust_amd", "lib", "libspeechsauce_amd_lab.so"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/wt.txt"
os.environ["SS_DEBUG_TIMES"] = out
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
import torch
import speechsauce_amd as ss
x = torch.randn((1024, 16000), device="cuda") * 0.1
ss.mfcc_batch(x, 16000)
torch.cuda.synchronize()
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wave_times.py"), out])

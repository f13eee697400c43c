#!/usr/bin/env python3
"""Runs one cfg2 batch with SS_DEBUG_TIMES set (per-wave stamps of one cold launch) and prints tools/wave_times.py's summary."""
import os, subprocess, sys
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

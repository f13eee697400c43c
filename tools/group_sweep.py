#!/usr/bin/env python3
"""How many batches per launch does it take?  bench.py's secondary legs with 1 / 2 / 4 / 8 independent batches per
ss_mfcc_batches_device / ss_mel_spectrogram_batches_device call, cfg2 / cfg3 / cfg5, interleaved on ONE box (two rounds); per-batch
microseconds and fraction of the 8 TB/s roofline.  The input ring holds 300 MiB of distinct batches per batch in flight (measure_simple scales it: on a fixed
300 MiB ring the Infinity Cache serves part of a multi-batch launch's input, tools/ring_check.py).
(profiles/r06/group_sweep.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import speechsauce_amd as ss  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(device)
for rnd in range(2):
    for wl in ("cfg2", "cfg3", "cfg5"):
        row = []
        for g in (1, 2, 4, 8):
            r = bench.measure_simple(torch, ss, wl, device, steps=960, warmup=96, prewarm_ms=150.0, group=g, probe_board=False, ring_mib=300)  # x the batches in flight (measure_simple)
            row.append(f"x{g}: {r['avg_launch_us']:6.2f} us = {r['frac']:.3f} ({r['kernel'].split('<')[0]})")
        print(f"round {rnd + 1} {wl}: " + " | ".join(row), flush=True)

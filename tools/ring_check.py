#!/usr/bin/env python3
"""Does the size of the rotated input ring matter when several batches are in flight at once?  cfg2 / cfg3 / cfg5 with one batch per
launch, four batches per launch and (cfg2) four streams, input rings of 300 / 1200 / 2400 MiB, interleaved on ONE box.  With a ring of
five 65 MB batches a launch of four batches re-reads three of the batches its predecessor read ~260 MB ago: at the edge of the
256 MiB Infinity Cache.  (profiles/r06/ring_check.txt)"""
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
        for mode, kw in (("x1", dict(group=1)), ("x4", dict(group=4)), ("4 streams", dict(streams=4))):
            if mode == "4 streams" and wl != "cfg2":
                continue
            row = []
            for ring in (300, 1200, 2400):
                r = bench.measure_simple(torch, ss, wl, device, steps=960, warmup=96, prewarm_ms=150.0, probe_board=False, ring_mib=max(1, ring // max(kw.get("group", 1), kw.get("streams", 1))), **kw)  # (measure_simple scales the ring by the batches in flight: undo it, `ring` is the total)
                per = r["ms_per_step"] * 1e3 if mode == "4 streams" else r["avg_launch_us"]
                row.append(f"ring {ring:4d} MiB: {per:6.2f} us")
            print(f"round {rnd + 1} {wl} {mode:9s}: " + " | ".join(row), flush=True)

#!/usr/bin/env python3
"""How the secondary lines of bench.py depend on their step count and pre-roll (r05: 200-step regions read 10 % slower than
1000-step ones).  usage: python tools/secondary_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import speechsauce_amd as ss
dev = torch.device("cuda", 0)
for wl in ("cfg3", "cfg5"):
    for steps, warm, pre in ((200, 20, 300.0), (1000, 100, 300.0), (200, 20, 1500.0), (1000, 100, 1500.0), (3000, 100, 300.0)):
        r = bench.measure_simple(torch, ss, wl, dev, steps=steps, warmup=warm, prewarm_ms=pre)
        print(wl, steps, warm, pre, round(r["avg_launch_us"], 2), "us  clk", r.get("clock_ghz_measured"), "cyc", round(r.get("cycles_per_launch", 0)),
              "board", (r.get("board") or {}).get("sclk_mhz_mean"), (r.get("board") or {}).get("power_w_mean"), flush=True)

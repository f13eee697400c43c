#!/usr/bin/env python3
"""Do the per-wave stamps of ss_mfcc_timed_region cost launch time?  cfg2 (1024 x 1 s clips, 5-batch input ring), 1000 launches per
region, regions with 0 / 256 / 1000 stamped launches interleaved on ONE box; prints the event-timed microseconds per launch, the
host wall time per launch and the stamps' clock for each.  (profiles/r06/stamp_cost.txt)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
import torch  # noqa: E402
from speechsauce_amd import SpeechConfig, _lib, make_params  # noqa: E402

lib = _lib.lib()
cfg = SpeechConfig(make_params(sample_rate=16000))
g = torch.Generator(device="cuda")
g.manual_seed(1)
xs = [torch.randn((1024, 16000), generator=g, device="cuda").mul_(0.1) for _ in range(5)]
outs = [torch.empty((1024, 98, 13), device="cuda") for _ in range(2)]
px = (C.c_void_p * 5)(*[x.data_ptr() for x in xs])
po = (C.c_void_p * 2)(*[o.data_ptr() for o in outs])
ms, ghz, wall = C.c_float(), C.c_float(), C.c_float()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for r in range(rounds + 1):
    for stamped in (0, 256, 1000):
        _lib.check(lib.ss_mfcc_timed_region(cfg.handle, px, 5, 1024, 16000, 16000, po, 2, None, 1000, stamped, C.byref(ms), C.byref(ghz), C.byref(wall)))
        if r:  # round 0 warms up
            print(f"round {r} stamped {stamped:4d}: {ms.value * 1e3:7.3f} us / launch (events)  {wall.value:7.3f} us / launch (host wall)  clock {ghz.value:.3f} GHz"
                  + (f"  -> {ms.value * 1e6 * ghz.value:8.0f} cycles" if ghz.value else ""))

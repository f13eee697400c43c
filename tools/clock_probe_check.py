#!/usr/bin/env python3
"""The one-wave clock probe (ss_shader_clock_probe, bench.probe_clock) against the 512-point kernel's own stamps
(ss_mfcc_shader_clock) on cfg2, alternating, on one box; then the probe on cfg3 / cfg5 beside the hwmon shader clock."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from speechsauce_amd import SpeechConfig, _lib, make_params
lib = _lib.lib(); dev = torch.device("cuda", 0)
for wl in ("cfg2", "cfg3", "cfg5"):
    desc, pkw, n, clips, kind = bench.WORKLOADS[wl]
    cfg = SpeechConfig(make_params(**pkw))
    rows = cfg.num_frames(n) if kind == "mfcc" else cfg.stft_rows(n)[0]
    out = torch.empty((clips, rows, cfg.params.num_cepstral) if kind == "mfcc" else (clips, cfg.params.num_filters, rows), device=dev)
    xs = [bench.synth_batch(torch, clips, n, 1 + i, dev) for i in range(5)]
    fn = lib.ss_mfcc_batch_device if kind == "mfcc" else lib.ss_mel_spectrogram_device
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def step(i): assert fn(cfg.handle, xs[i % 5].data_ptr(), clips, n, n, out.data_ptr(), sp) == 0
    for i in range(5000): step(i)
    torch.cuda.synchronize()
    for rep in range(4):
        probe = bench.probe_clock(torch, lib, step, {"cfg2": 29.5e-6, "cfg3": 45e-6, "cfg5": 61e-6}[wl], dev)
        own = None
        if wl == "cfg2":
            for i in range(300): step(i)
            g = C.c_float(0.0)
            if lib.ss_mfcc_shader_clock(cfg.handle, xs[0].data_ptr(), clips, n, n, out.data_ptr(), sp, 30, C.byref(g)) == 0: own = g.value
        pb = bench.BoardProbe(torch, dev)
        for k in range(40):
            for i in range(100): step(i)
            torch.cuda.synchronize()
        board = pb.stop() or {}
        print(wl, "probe %.3f GHz" % (probe or 0), "| kernel's own stamps", ("%.3f GHz" % own) if own else "-", "| hwmon sclk", board.get("sclk_mhz_mean"), "MHz at", board.get("power_w_mean"), "W", flush=True)

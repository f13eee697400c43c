#!/usr/bin/env python3
"""Does the one-wave clock probe run BESIDE back-to-back launches of the persistent kernels?  (r05 diagnostic)"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from speechsauce_amd import SpeechConfig, _lib, make_params
lib = _lib.lib()
dev = torch.device("cuda", 0)
for wl in ("cfg2", "cfg3", "cfg5"):
    desc, pkw, n, clips, kind = bench.WORKLOADS[wl]
    cfg = SpeechConfig(make_params(**pkw))
    rows = cfg.num_frames(n) if kind == "mfcc" else cfg.stft_rows(n)[0]
    out = torch.empty((clips, rows, cfg.params.num_cepstral) if kind == "mfcc" else (clips, cfg.params.num_filters, rows), device=dev)
    xs = [bench.synth_batch(torch, clips, n, 1 + i, dev) for i in range(5)]
    fn = lib.ss_mfcc_batch_device if kind == "mfcc" else lib.ss_mel_spectrogram_device
    main = torch.cuda.current_stream(); sp = C.c_void_p(main.cuda_stream)
    side = torch.cuda.Stream(device=dev); ssp = C.c_void_p(side.cuda_stream)
    def step(i): assert fn(cfg.handle, xs[i % 5].data_ptr(), clips, n, n, out.data_ptr(), sp) == 0
    for i in range(3000): step(i)
    torch.cuda.synchronize()
    for mode in ("mid", "first", "first+warm"):
        for probe_us in (500, 2000):
            g = C.c_float(0.0); res = {}
            nl = int(6 * probe_us / 30)
            t0 = time.perf_counter()
            if mode == "mid":
                for i in range(nl // 4): step(i)
                def run(): res["rc"] = lib.ss_shader_clock_probe(ssp, probe_us, C.byref(g)); res["t"] = time.perf_counter() - t0
                th = threading.Thread(target=run); th.start()
                for i in range(nl): step(i)
            else:
                if mode == "first+warm":
                    for i in range(200): step(i)
                def run(): res["rc"] = lib.ss_shader_clock_probe(ssp, probe_us, C.byref(g)); res["t"] = time.perf_counter() - t0
                th = threading.Thread(target=run); th.start()
                for i in range(nl): step(i)
            behind = torch.cuda.Event(); behind.record(main)
            t_enq = time.perf_counter() - t0
            th.join()
            beside = not behind.query()
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(wl, mode, probe_us, "clk %.3f" % g.value, "beside", beside, "enqueue %.2f ms probe returned %.2f ms all done %.2f ms" % (t_enq * 1e3, res["t"] * 1e3, t_all * 1e3), flush=True)

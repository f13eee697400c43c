#!/usr/bin/env python3
"""Unit timeline of the twelve-wave 2048-point mel kernel (cfg3): a lab build with -DSS_PROF3=1 (tools/ablate.sh, SRC=ss_mel2048)
stamps, per wave and unit, the 100 MHz clock at the unit's top, at the arrival of its samples and at its end, into words behind
the output block, which this script allocates.   SS_LIB_PATH=$PWD/ab/lib_<name>.so python tools/prof3.py [zeros]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from speechsauce_amd import SpeechConfig, _lib, make_params
desc, pkw, n, clips, kind = bench.WORKLOADS["cfg3"]
cfg = SpeechConfig(make_params(**pkw)); lib = _lib.lib()
rows = cfg.stft_rows(n)[0]; M = cfg.params.num_filters
ncu = torch.cuda.get_device_properties(0).multi_processor_count
nw = ncu * 12
out = torch.zeros((clips * M * rows + 2 * 32 * nw,), dtype=torch.float32, device="cuda")
zeros = len(sys.argv) > 1 and sys.argv[1] == "zeros"
xs = [torch.zeros((clips, n), device="cuda") if zeros else bench.synth_batch(torch, clips, n, 1 + i, "cuda") for i in range(5)]
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(200):
    assert lib.ss_mel_spectrogram_device(cfg.handle, xs[i % 5].data_ptr(), clips, n, n, out.data_ptr(), sp) == 0
torch.cuda.synchronize()
w = out[clips * M * rows:].cpu().numpy().view(np.uint64).reshape(nw, 32)
cnt = (w[:, 0] & 0xffffffff).astype(int); xcc = (w[:, 0] >> 32).astype(int)
t0 = min(int(w[i, 1]) for i in range(nw) if cnt[i])
loads, comp, ends, starts = [], [], [], []
for i in range(nw):
    for k in range(min(cnt[i], 10)):
        a, b, c = (int(w[i, 1 + 3 * k + j]) for j in range(3))
        loads.append((b - a) / 100.0); comp.append((c - b) / 100.0)
        if k == 0: starts.append((a - t0) / 100.0)
    if cnt[i]: ends.append((int(w[i, 1 + 3 * (min(cnt[i], 10) - 1) + 2]) - t0) / 100.0)
loads, comp, ends, starts = map(np.array, (loads, comp, ends, starts))
print(lib.ss_last_kernel_name().decode(), "zeros" if zeros else "ring", "| waves", nw, "units per wave: mean %.2f min %d max %d" % (cnt.mean(), cnt.min(), cnt.max()),
      "hist", np.bincount(cnt).tolist())
print("sample wait per unit  us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (loads.mean(), *np.percentile(loads, [10, 50, 90]), loads.max()))
print("compute per unit      us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (comp.mean(), *np.percentile(comp, [10, 50, 90]), comp.max()))
print("first unit starts     us: p50 %.2f max %.2f | wave ends us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % (np.median(starts), starts.max(), *np.percentile(ends, [10, 50, 90]), ends.max()))
cu_end = ends.reshape(-1, 12).max(axis=1) if len(ends) == nw else None
if cu_end is not None:
    print("CU ends               us: p10 %.2f p50 %.2f p90 %.2f max %.2f | per XCC mean" % tuple(np.percentile(cu_end, [10, 50, 90]).tolist() + [cu_end.max()]),
          [round(float(cu_end[xcc.reshape(-1, 12)[:, 0] == x].mean()), 2) for x in range(8)])
# first / middle / last unit of a wave
for k in range(0, 7):
    sel = [(int(w[i, 2 + 3 * k]) - int(w[i, 1 + 3 * k]), int(w[i, 3 + 3 * k]) - int(w[i, 2 + 3 * k])) for i in range(nw) if cnt[i] > k]
    if sel:
        a = np.array(sel) / 100.0
        print("unit #%d of a wave (%4d waves): wait %.2f  compute %.2f" % (k, len(sel), a[:, 0].mean(), a[:, 1].mean()))
# by wave index within the workgroup (waves 0-3 are dispatched first: the oldest wave of each SIMD)
per = w.reshape(-1, 12, 32)
c12 = cnt.reshape(-1, 12)
print("by wave index: units        ", [round(float(c12[:, k].mean()), 2) for k in range(12)])
dur = np.zeros((per.shape[0], 12)); wt = np.zeros((per.shape[0], 12)); en = np.zeros((per.shape[0], 12))
for b in range(per.shape[0]):
    for k in range(12):
        n_ = min(int(c12[b, k]), 10)
        if n_:
            st_ = per[b, k, 1:1 + 3 * n_].astype(np.int64).reshape(n_, 3)
            dur[b, k] = ((st_[:, 2] - st_[:, 1]).mean()) / 100.0
            wt[b, k] = ((st_[:, 1] - st_[:, 0]).mean()) / 100.0
            en[b, k] = (int(st_[-1, 2]) - t0) / 100.0
print("by wave index: compute us   ", [round(float(dur[:, k].mean()), 2) for k in range(12)])
print("by wave index: wait us      ", [round(float(wt[:, k].mean()), 2) for k in range(12)])
print("by wave index: end us       ", [round(float(en[:, k].mean()), 2) for k in range(12)])
print("last wave of a CU is index  ", np.bincount(en.argmax(axis=1), minlength=12).tolist())

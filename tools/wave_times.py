#!/usr/bin/env python3
"""Summarise a SS_DEBUG_TIMES dump: per-wave start / prologue-end / end stamps (100 MHz ticks)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
t0 = d[:, 1].min()
st, pro, en, nq, xcc = (d[:, 1] - t0) / 100.0, (d[:, 2] - t0) / 100.0, (d[:, 3] - t0) / 100.0, d[:, 4], d[:, 5]
print(f"waves {len(d)}  span {en.max():.1f} us")
for name, v in (("start", st), ("prologue_end", pro), ("end", en), ("lifetime", en - st), ("main_loop", en - pro)):
    print(f"{name:13s} min {v.min():7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f} us")
for q in np.unique(nq):
    m = nq == q
    print(f"quads={q}: {m.sum()} waves, main_loop p50 {np.median((en-pro)[m]):.2f} us, per quad {np.median((en-pro)[m])/q*1000:.0f} ns")
for x in np.unique(xcc):
    m = xcc == x
    print(f"xcc {x}: {m.sum()} waves, start p50 {np.median(st[m]):.2f} end p50 {np.median(en[m]):.2f} max {en[m].max():.2f}")

if d.shape[1] >= 14:
    names = ["loop+input wait", "radix-16 #1", "LDS exchange", "twiddle+radix-16 #2", "untangle+|X|+energy", "mel+ln+DCT (MFMA)", "stage+store", "-"]
    seg = d[:, 6:14].astype(np.float64)
    tot = seg.sum(axis=1)
    print("segment shares (shader-clock ticks, mean over waves; stamps fence overlaps, read SHARES not lengths):")
    for i, nme in enumerate(names[:7]):
        print(f"  {nme:24s} {seg[:, i].mean():12.0f}  {100 * seg[:, i].sum() / tot.sum():5.1f} %")
    print(f"  total per wave {tot.mean():.0f} ticks over {nq.mean():.2f} chunks -> {tot.mean() / nq.mean():.0f} ticks per chunk")

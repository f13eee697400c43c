#!/usr/bin/env python3
"""Summarise a SS_DEBUG_TIMES dump of the 512-point kernel: per-wave stamps in 100 MHz ticks
(columns: wave, start, prologue end, end, quads done, xcc, first samples arrived, last quad claimed)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
t0 = d[:, 1].min()
us = lambda c: (d[:, c] - t0) / 100.0
st, pro, en, nq, xcc = us(1), us(2), us(3), d[:, 4], d[:, 5]
print(f"waves {len(d)}  span {en.max():.1f} us")
cols = [("start", st), ("prologue_end", pro), ("end", en), ("lifetime", en - st), ("main_loop", en - pro)]
if d.shape[1] >= 8:
    first, last = us(6), us(7)  # column 7: table words arrived (prologue)
    cols += [("first_samples", first)]
    raw = d[:, 7]
    cyc = (raw >= (1 << 40)) & (raw < (1 << 41))  # waves other than the table waves report the shader cycles they lived (flag bit 40)
    if cyc.any():
        cycles = (raw[cyc] - (1 << 40)).astype(np.float64)
        life_us = (d[cyc, 3] - d[cyc, 1]) / 100.0
        print(f"shader clock over the waves' lifetime: mean {np.mean(cycles / life_us):.0f} MHz  (min {np.min(cycles / life_us):.0f}, max {np.max(cycles / life_us):.0f})")
    # only the workgroup's table waves stamp this: ticks since the wave's start, flagged 2 in bits 40..41
    tabm = (raw >> 40) == 2
    tab = (st + (raw & ((1 << 40) - 1)) / 100.0)[tabm]
    if len(tab):
        cols += [("table_in_lds", np.pad(tab, (0, len(first) - len(tab)), mode="edge"))]
for name, v in cols:
    print(f"{name:15s} min {v.min():7.2f}  p10 {np.percentile(v,10):7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f} us")
for q in np.unique(nq):
    m = nq == q
    print(f"quads={q}: {m.sum()} waves, main_loop p50 {np.median((en-pro)[m]):.2f} us, per quad {np.median((en-pro)[m])/q*1000:.0f} ns")
cu = d[:, 0] // 12
cu_end = np.array([en[cu == c].max() for c in np.unique(cu)])
cu_first_end = np.array([en[cu == c].min() for c in np.unique(cu)])
print(f"per-CU end: min {cu_end.min():.2f} p50 {np.median(cu_end):.2f} max {cu_end.max():.2f} us; first wave done -> last wave done within a CU: p50 {np.median(cu_end - cu_first_end):.2f} max {(cu_end - cu_first_end).max():.2f} us")
for x in np.unique(xcc):
    m = xcc == x
    print(f"xcc {x}: {m.sum()} waves, start p50 {np.median(st[m]):.2f} end p50 {np.median(en[m]):.2f} max {en[m].max():.2f}")

#!/usr/bin/env python3
"""Are the same workgroups of the cfg2 launch late every time?  Stamps SS_DEBUG_TIMES_REPS launches (cold samples each) and
prints, per launch, the spread of the workgroups' end times, and across launches the correlation of each workgroup's lateness
(end - median end) -- a stable pattern could be levelled by giving those workgroups fewer quads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/skew.txt"
os.environ["SS_DEBUG_TIMES"] = out
# SS_DEBUG_TIMES exists in the LAB build of the library only (`make -C mfcc-rust_amd/csrc lab`)
os.environ.setdefault("SS_LIB_PATH", os.path.join(ROOT, "mfcc-rust_amd", "lib", "libspeechsauce_amd_lab.so"))
os.environ["SS_DEBUG_TIMES_REPS"] = os.environ.get("SS_DEBUG_TIMES_REPS", "12")
sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
import numpy as np
import torch
import speechsauce_amd as ss
x = torch.randn((1024, 16000), device="cuda") * 0.1
ss.mfcc_batch(x, 16000)
torch.cuda.synchronize()
reps = int(os.environ["SS_DEBUG_TIMES_REPS"])
ends, starts, firsts = [], [], []
for k in range(2, reps):
    d = np.loadtxt(out if k == reps - 1 else f"{out}.{k}", dtype=np.int64)
    t0 = d[:, 1].min()
    wg = d[:, 0] // 12
    n = wg.max() + 1
    e = np.array([(d[wg == c, 3].max() - t0) / 100.0 for c in range(n)])
    s = np.array([(d[wg == c, 1].min() - t0) / 100.0 for c in range(n)])
    f = np.array([(d[wg == c, 6].min() - t0) / 100.0 for c in range(n)])
    ends.append(e); starts.append(s); firsts.append(f)
    print(f"launch {k}: wg end min {e.min():.2f} p50 {np.median(e):.2f} p90 {np.percentile(e, 90):.2f} max {e.max():.2f} us; start p50 {np.median(s):.2f} max {s.max():.2f}; first samples p50 {np.median(f):.2f} max {f.max():.2f}")
E = np.array(ends); late = E - np.median(E, axis=1, keepdims=True)
c = np.corrcoef(late)
print("correlation of workgroup lateness between launches: mean off-diagonal %.2f (min %.2f)" % ((c.sum() - len(c)) / (len(c) * (len(c) - 1)), c.min()))
m = late.mean(axis=0)
print("mean lateness per workgroup: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us; std of the mean %.2f, mean within-wg std %.2f" % (m.min(), np.percentile(m, 10), np.median(m), np.percentile(m, 90), m.max(), m.std(), late.std(axis=0).mean()))
xcd = np.arange(len(m)) % 8
for x8 in range(8):
    print(f"  workgroups with blockIdx % 8 == {x8}: mean lateness {m[xcd == x8].mean():+.2f} us")
order = np.argsort(-m)[:10]
print("ten latest workgroups (blockIdx: mean lateness):", ", ".join(f"{i}: {m[i]:+.2f}" for i in order))
lev = E.max(axis=1) - np.median(E, axis=1)
print("launch end - median workgroup end: mean %.2f us" % lev.mean())
np.save("/tmp/skew_late.npy", late)

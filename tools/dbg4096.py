"""Diagnostic: compare the stages of ss_mfcc_c2048 for frame 0 (SS_DEBUG_ROWS) with a numpy emulation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mfcc-rust_amd"))
os.environ["SS_DEBUG_ROWS"] = "/tmp/rows.bin"
import torch
import speechsauce_amd as ss

rng = np.random.default_rng(5)
x = (rng.standard_normal((2, 44100)) * 0.1).astype(np.float32)
kw = dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096)
out = ss.mfcc_batch(torch.from_numpy(x).cuda(), 44100, **kw).cpu().numpy()
rows = np.fromfile("/tmp/rows.bin", dtype=np.float32)
P = rows[:1028]
st = rows[1284:].reshape(4, 64, 32, 2)  # all zero unless kDbgStages is set in ss_mfcc4096.hip
st = st[..., 0] + 1j * st[..., 1]
xx = x[0, :4096].astype(np.float64)
z = xx[0::2] + 1j * xx[1::2]
lane = np.arange(64)
W = lambda n, den: np.exp(-2j * np.pi * n / den)
v = np.stack([z[lane + 64 * e] for e in range(32)], axis=1)
v = np.fft.fft(v, axis=1)
def cmp(name, got, want):
    err = np.abs(got - want)
    print(name, "max err", err.max(), "of", np.abs(want).max())
    if err.max() > 1e-3 * np.abs(want).max():
        bad = np.argwhere(err > 1e-3 * np.abs(want).max())
        print("  n bad", len(bad), "first (lane, reg):", bad[:12].tolist())
        print("  bad lanes:", sorted(set(bad[:, 0].tolist()))[:64])
        print("  bad regs:", sorted(set(bad[:, 1].tolist())))
cmp("pass1", st[0], v)
u = np.zeros((64, 32), complex)
for L in range(64):
    k1, d = L & 31, L >> 5
    for b in range(32):
        u[L, b] = v[d + 2 * b, k1]
cmp("exchange", st[1], u)
for L in range(64):
    k1 = L & 31
    u[L] *= W(np.arange(32) * k1, 1024)
u = np.fft.fft(u, axis=1)
cmp("pass2", st[2], u)
Z = np.zeros((64, 32), complex)
for L in range(64):
    k1, h = L & 31, L >> 5
    for e in range(32):
        c = (e & 15) + 16 * h
        Z[L, e] = u[k1, c] + (-1) ** (e >> 4) * W(k1 + 32 * c, 2048) * u[k1 + 32, c]  # bin k1 + 32 c + 1024 (e >> 4)
cmp("radix2", st[3], Z)
ref = 2 * np.abs(np.fft.rfft(xx))
print("P bad:", int((np.abs(P[:1025] - ref[:1025]) > 1e-4 * ref.max()).sum()))

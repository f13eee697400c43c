#!/usr/bin/env python3
"""Worst observed parity errors per BASELINE configuration and signal class, HIP path (through the C ABI) against the
f64-accumulating oracle, with the oracle's own f32 reference-shaped port measured the same way beside it.

    python tools/parity_report.py profiles/r02/parity.json        (on the GPU box)

Metrics per (config, signal), all over one clip:
  max_norm      max|got - want| / max|want| over the whole block (the round-1 metric)
  col0_norm     the same over column 0 alone (ln E when dc_elimination is on)          -- MFCC only
  rest_norm     the same over columns 1.. alone (the cepstra proper)                   -- MFCC only
  elem_rel      max over elements with |want| > 1e-3 max|want| of |got - want| / |want| (columns 1.. for MFCC)
`port_*` are the same numbers for oracle/ss_oracle.c's single-thread f32 port: what f32 arithmetic in the reference's
own operation order costs against the f64 oracle.  tests/test_gpu_parity_strict.py asserts on these metrics.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mfcc-rust_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from common import CONFIGS, N_SAMPLES  # noqa: E402


def signals(n, sr):
    """Golden signals plus quiet clips and a clip with silent frames (digital silence between bursts)."""
    t = np.arange(n)
    noise = (np.random.default_rng(0).standard_normal(n) * 0.1).astype(np.float32)
    gaps = noise.copy()
    gaps[n // 5: 2 * n // 5] = 0.0
    gaps[3 * n // 5: 7 * n // 10] = 0.0
    return {
        "noise": noise,
        "sine1k": (0.5 * np.sin(2 * np.pi * 1000.0 * t / sr)).astype(np.float32),
        "dc": np.full(n, 0.25, np.float32),
        "impulse": np.where(t % 160 == 0, 1.0, 0.0).astype(np.float32),
        "quiet_1e-3": (noise * 1e-3).astype(np.float32),
        "quiet_1e-5": (noise * 1e-5).astype(np.float32),
        "silent_frames": gaps,
    }


def metrics(got, want, mfcc):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    out = {"max_norm": float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))}
    body_g, body_w = (got[:, 1:], want[:, 1:]) if mfcc else (got, want)
    if mfcc:
        out["col0_norm"] = float(np.abs(got[:, 0] - want[:, 0]).max() / max(np.abs(want[:, 0]).max(), 1e-300))
        out["rest_norm"] = float(np.abs(body_g - body_w).max() / max(np.abs(body_w).max(), 1e-300))
    big = np.abs(body_w) > 1e-3 * np.abs(body_w).max()
    out["elem_rel"] = float((np.abs(body_g - body_w)[big] / np.abs(body_w)[big]).max()) if big.any() else 0.0
    return out


def run(ss, oracle):
    rep = {}
    for name, kw in CONFIGS.items():
        n, sr = N_SAMPLES[name], kw["sample_rate"]
        p = oracle.make_params(**kw)
        mfcc = name != "cfg3"
        py_kw = dict(frame_length=kw.get("frame_length", 0.02), frame_stride=kw.get("frame_stride", 0.01),
                     num_cepstral=kw.get("num_cepstral", 13), num_filters=kw.get("num_filters", 40),
                     fft_length=kw.get("fft_points", 512), high_frequency=kw.get("high_frequency"))
        rep[name] = {}
        for sname, x in signals(n, sr).items():
            if mfcc:
                got, want, port = ss.mfcc(x, sr, **py_kw), oracle.mfcc(p, x), oracle.port_mfcc(p, x)
            else:
                got, want, port = ss.mel_spectrogram(x, sr, **py_kw), oracle.mel_spectrogram(p, x), oracle.port_mel_spectrogram(p, x)
            m = metrics(got, want, mfcc)
            m.update({"port_" + k: v for k, v in metrics(port, want, mfcc).items()})
            rep[name][sname] = m
    return rep


if __name__ == "__main__":
    import oracle_c
    import speechsauce_amd as ss

    rep = run(ss, oracle_c)
    worst = {c: {k: max(v[k] for v in sig.values()) for k in next(iter(sig.values()))} for c, sig in rep.items()}
    doc = {"kernel_library": "mfcc-rust_amd/lib/libspeechsauce_amd.so", "tolerance": "1e-4 (BASELINE.json north_star)", "worst": worst, "cases": rep}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02", "parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(worst, indent=1))

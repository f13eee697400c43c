"""Shared test helpers: BASELINE.json configs, seeded signals, golden fixture access."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "golden_v1.npz")

CONFIGS = {
    "cfg1": dict(sample_rate=16000),
    "cfg3": dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128,
                 high_frequency=8000.0),
    "cfg5": dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                 num_cepstral=40, num_filters=256, high_frequency=22050.0),
}
N_SAMPLES = {"cfg1": 16000, "cfg3": 16000, "cfg5": 44100}
# the kernel builds bench.py's BASELINE workloads run on (asserted by name wherever a test means "what the bench times")
BENCH_KERNELS = {
    "cfg2": b"ss_mfcc_c256<10,exact,bank421,sym>",
    "cfg3": b"ss_mel_c1024<w12,mel6321>",
    "cfg5": b"ss_mfcc_c2048<exact,mel8321,w12>",
}
RTOL = 1e-4  # BASELINE.json north_star: outputs within 1e-4 (relative to the per-clip max)


def golden_signals(n, sr):
    """Same construction as tests/golden/make_golden.py."""
    t = np.arange(n)
    return {
        "noise": (np.random.default_rng(0).standard_normal(n) * 0.1).astype(np.float32),
        "sine1k": (0.5 * np.sin(2 * np.pi * 1000.0 * t / sr)).astype(np.float32),
        "dc": np.full(n, 0.25, np.float32),
        "impulse": np.where(t % 160 == 0, 1.0, 0.0).astype(np.float32),
    }


def rel(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def load_golden():
    return np.load(GOLDEN)

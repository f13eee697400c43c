#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the numpy restatement
(oracle/oracle_np.py).  Run in the build container:  python tests/golden/make_golden.py

The reference crate is Rust and cannot be built or imported here, and it ships no golden vectors
(its tests assert shapes only), so these fixtures pin OUR restatement of its semantics: the C
oracle, the C-ABI host tables and the HIP path are all checked against them.  Fixtures are data
only: seeded inputs are regenerated from the seed, expected outputs are stored as float64/int32.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import oracle_np as on  # noqa: E402

CONFIGS = {
    # BASELINE.json configs 1/2/4, 3, 5
    "cfg1": dict(sample_rate=16000),
    "cfg3": dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128,
                 high_frequency=8000.0),
    "cfg5": dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                 num_cepstral=40, num_filters=256, high_frequency=22050.0),
}
N_SAMPLES = {"cfg1": 16000, "cfg3": 16000, "cfg5": 44100}


def signals(n, sr):
    t = np.arange(n)
    return {
        "noise": (np.random.default_rng(0).standard_normal(n) * 0.1).astype(np.float32),
        "sine1k": (0.5 * np.sin(2 * np.pi * 1000.0 * t / sr)).astype(np.float32),
        "dc": np.full(n, 0.25, np.float32),
        "impulse": np.where(t % 160 == 0, 1.0, 0.0).astype(np.float32),
    }


def main():
    out = {}
    for name, kw in CONFIGS.items():
        p = on.Params(**kw)
        fb, idx = on.filterbank(p)
        nz = np.nonzero(fb)
        out[f"{name}/fb_idx"] = idx.astype(np.int32)
        out[f"{name}/fb_nz_rows"] = nz[0].astype(np.int32)
        out[f"{name}/fb_nz_cols"] = nz[1].astype(np.int32)
        out[f"{name}/fb_nz_vals"] = fb[nz].astype(np.float32)
        out[f"{name}/vorbis_window"] = on.vorbis_window(p.fft_points)
        n = N_SAMPLES[name]
        for sname, x in signals(n, p.sample_rate).items():
            if name == "cfg3":
                out[f"{name}/{sname}/mel"] = on.mel_spectrogram(p, x)          # [M, R]
                S = on.stft(p, x)[0]
                out[f"{name}/{sname}/stft_row5"] = np.stack([S[5].real, S[5].imag])
            else:
                feat, en = on.mfe(p, x)
                out[f"{name}/{sname}/mfcc"] = on.mfcc(p, x)
                out[f"{name}/{sname}/energy"] = en
                out[f"{name}/{sname}/feat_rows"] = feat[[0, feat.shape[0] // 2, -1]]
                out[f"{name}/{sname}/P_row1"] = on.power_spectrum(p, x)[1]
    # frame-count table incl. the reference's own test shapes (lib.rs:50-68, 93-134)
    p1 = on.Params()
    lens = [480, 481, 639, 640, 641, 8191, 16000, 16001, 44100, 1_000_000]
    out["frames/lengths"] = np.array(lens, np.int64)
    out["frames/cfg1"] = np.array([on.num_frames(p1, n) for n in lens], np.int64)
    out["frames/padded_stride0.02_1e6"] = np.array([on.num_frames_padded(on.Params(frame_stride=0.02), 1_000_000)], np.int64)
    # switches on the noise clip
    x = signals(16000, 16000)["noise"]
    for tag, sw in {"pow2": dict(spectrum_exponent=2), "ortho": dict(dct_norm="ortho"), "hann": dict(mfcc_window="hann"),
                    "preemph": dict(preemph_coef=0.97), "nodc": dict(dc_elimination=False),
                    "literal": dict(framing="literal")}.items():
        out[f"switch/{tag}"] = on.mfcc(on.Params(**sw), x)
    out["preemphasis/shift1_cof0.98"] = on.preemphasis(x[:1000], 1, 0.98)
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()

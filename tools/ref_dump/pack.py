#!/usr/bin/env python3
"""Packs the directory written by `cargo run --release -- <dir>` (tools/ref_dump) into tests/golden/reference_v1.npz.

    python3 tools/ref_dump/pack.py <dir> tests/golden/reference_v1.npz

Also holds the input generator shared with src/main.rs, so that the test can rebuild the inputs and check that the
vectors in the file are the ones the Rust program says it used."""
import os
import sys

import numpy as np


def lcg_signal(seed: int, n: int, amp: float) -> np.ndarray:
    """x[i] = ((state >> 8) / 2^24 - 0.5) * 2 * amp, state <- state * 1664525 + 1013904223 (mod 2^32): main.rs lcg_signal."""
    out = np.empty(n, np.float32)
    s = np.uint64(seed)
    a, c, m = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    for i in range(n):
        s = (s * a + c) & m
        out[i] = (np.float32(int(s) >> 8) / np.float32(16777216.0) - np.float32(0.5)) * np.float32(2.0) * np.float32(amp)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    arrays = {}
    for line in open(os.path.join(src, "manifest.txt")):
        name, dims = line.split()
        shape = tuple(int(d) for d in dims.split(","))
        arrays[name] = np.fromfile(os.path.join(src, name + ".f32"), dtype="<f4").reshape(shape)
    np.savez_compressed(dst, **arrays)
    print(f"{dst}: {len(arrays)} arrays")


if __name__ == "__main__":
    main()

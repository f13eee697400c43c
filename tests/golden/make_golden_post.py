#!/usr/bin/env python3
"""Fixtures for the post-processing functions (cmvn, cmvnw, derivative_extraction, extract_derivative_feature), generated
from the np.pad-based numpy restatement (oracle/oracle_np.py).  Run: python tests/golden/make_golden_post.py

Same policy as make_golden.py: the reference ships no golden vectors for these functions, so the fixtures pin OUR
restatement (numpy's own pad semantics, which the reference quotes in util.rs:108-124) for the C oracle and the HIP path.
Inputs are regenerated from seeds; expected outputs are stored as float64.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import oracle_np as on  # noqa: E402

CASES = {"mfcc_like": (7, (98, 13)), "short": (8, (5, 4)), "wide": (9, (39, 40))}


def matrix(seed, shape):
    return (np.random.default_rng(seed).standard_normal(shape) * 2.0 + 0.5).astype(np.float32)


def main():
    out = {}
    for name, (seed, shape) in CASES.items():
        v = matrix(seed, shape)
        for var in (0, 1):
            out[f"{name}/cmvn{var}"] = on.cmvn(v, bool(var))
            for win in (3, 31, 301):
                out[f"{name}/cmvnw{var}_{win}"] = on.cmvnw(v, win, bool(var))
        for dw in (1, 2, 9):
            out[f"{name}/deriv{dw}"] = on.derivative_extraction(v, dw)
        out[f"{name}/cube"] = on.extract_derivative_feature(v)
    path = os.path.join(HERE, "golden_post_v1.npz")
    np.savez_compressed(path, **out)
    print(path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

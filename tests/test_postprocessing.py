"""Post-processing on the feature matrix: cmvn, cmvnw, derivative_extraction, extract_derivative_feature.

CPU (-m "not gpu"): pins the C oracle against (a) the property the reference's own test asserts (lib.rs:70-91: cmvn with
variance normalisation -> column mean 0 and std 1), (b) the np.pad semantics the reference quotes (util.rs:108-124) through
the independent numpy restatement, (c) hand-computed known answers.  The reference holds no golden vectors for these
functions either ("parity unpinned").  GPU (-m gpu): the HIP kernels against the oracle through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

RTOL = 1e-4


def _rel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def _mat(seed, shape):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


# ---------------------------------------------------------------------------------------------------- oracle pins (CPU)

def test_reference_cmvn_property(oracle):
    """lib.rs:70-91: Uniform(0,1) [50 x 100], variance_normalization = true -> mean 0, std 1 per column."""
    v = np.random.default_rng(3).uniform(0, 1, (50, 100)).astype(np.float32)
    out = oracle.cmvn(v, True)
    assert out.shape == v.shape
    assert np.abs(out.mean(axis=0)).max() < 1e-8
    assert np.abs(out.std(axis=0) - 1).max() < 1e-8


def test_oracle_matches_numpy_pad_semantics(oracle):
    import oracle_np

    for seed, shape in enumerate([(98, 13), (5, 4), (50, 100), (1, 3), (39, 40), (7, 1)]):
        v = _mat(seed, shape)
        for var in (False, True):
            assert np.abs(oracle.cmvn(v, var) - oracle_np.cmvn(v, var)).max() < 1e-12
            for win in (1, 3, 31, 301):  # 301 (the reference default) wraps a 98-row matrix more than once
                if var and shape[0] == 1:
                    continue  # std of a constant window: 0/(0+eps), skip the degenerate case
                assert np.abs(oracle.cmvnw(v, win, var) - oracle_np.cmvnw(v, win, var)).max() < 1e-9
        for dw in (1, 2, 9):
            assert np.abs(oracle.derivative_extraction(v, dw) - oracle_np.derivative_extraction(v, dw)).max() < 1e-12
        assert np.abs(oracle.extract_derivative_feature(v) - oracle_np.extract_derivative_feature(v)).max() < 1e-12


def test_known_answers(oracle):
    # np.pad([1,2,3,4,5], (2,3), 'symmetric') = [2,1,1,2,3,4,5,5,4,3] (util.rs:108-112): window 5 centred on row 0 sees
    # rows [2,1,1,2,3] -> mean 1.8
    v = np.arange(1, 6, dtype=np.float32).reshape(5, 1)
    out = oracle.cmvnw(v, 5, False)
    assert abs(out[0, 0] - (1 - 1.8)) < 1e-12 and abs(out[2, 0]) < 1e-12 and abs(out[4, 0] - (5 - 4.2)) < 1e-12
    # derivative, delta_windows = 1, literally R*f[c+1] - f[c-1] over the edge-padded row, / 2
    f = np.array([[1.0, 2.0, 4.0, 8.0]], dtype=np.float32)
    assert np.allclose(oracle.derivative_extraction(f, 1), [[(2 - 1) / 2, (4 - 1) / 2, (8 - 2) / 2, (8 - 4) / 2]])
    # delta_windows = 2: sum_R (R f[c+R] - f[c-R]) / (2 + 8)
    assert np.allclose(oracle.derivative_extraction(f, 2)[0, 0], ((2 - 1) + (2 * 4 - 1)) / 10)
    cube = oracle.extract_derivative_feature(f)
    assert cube.shape == (1, 4, 3) and np.allclose(cube[..., 0], f)
    assert np.allclose(cube[..., 2], oracle.derivative_extraction(oracle.derivative_extraction(f, 2), 2))
    with pytest.raises(oracle.OracleError):  # assert!(win_size % 2 == 1), processing.rs:327
        oracle.cmvnw(v, 4, False)


def _golden_cases():
    from golden.make_golden_post import CASES, matrix

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_post_v1.npz"))
    return CASES, matrix, g


def _check_against_golden(fns, tol):
    cases, matrix, g = _golden_cases()
    for name, (seed, shape) in cases.items():
        v = matrix(seed, shape)
        for var in (0, 1):
            assert _rel(fns.cmvn(v, bool(var)), g[f"{name}/cmvn{var}"]) <= tol
            for win in (3, 31, 301):
                assert _rel(fns.cmvnw(v, win, bool(var)), g[f"{name}/cmvnw{var}_{win}"]) <= tol, (name, var, win)
        for dw in (1, 2, 9):
            assert _rel(fns.derivative_extraction(v, dw), g[f"{name}/deriv{dw}"]) <= tol
        assert _rel(fns.extract_derivative_feature(v), g[f"{name}/cube"]) <= tol


def test_oracle_against_committed_fixtures(oracle):
    """tests/golden/golden_post_v1.npz (np.pad-based restatement) pins the C oracle."""
    _check_against_golden(oracle, 1e-12)


# ------------------------------------------------------------------------------------------------------- HIP path (GPU)

@pytest.mark.gpu
def test_gpu_against_committed_fixtures(ss):
    _check_against_golden(ss, RTOL)


@pytest.mark.gpu
def test_cmvn_gpu(ss, oracle):
    for seed, shape in enumerate([(98, 13), (6248, 13), (50, 100), (1, 3), (39, 40)]):
        v = _mat(10 + seed, shape) * 3 + 1
        for var in (False, True):
            if var and shape[0] == 1:
                continue
            assert _rel(ss.cmvn(v, var), oracle.cmvn(v, var)) <= RTOL, (shape, var)
    # the reference's own assertion (lib.rs:70-91) on the device result
    u = np.random.default_rng(3).uniform(0, 1, (50, 100)).astype(np.float32)
    out = ss.cmvn(u, True).astype(np.float64)
    assert np.abs(out.mean(axis=0)).max() < 1e-6 and np.abs(out.std(axis=0) - 1).max() < 1e-6


@pytest.mark.gpu
def test_cmvnw_gpu(ss, oracle):
    from speechsauce_amd import SpeechSauceError

    for seed, shape in enumerate([(98, 13), (5, 4), (400, 13)]):
        v = _mat(20 + seed, shape) * 2 - 0.5
        for win in (1, 3, 31, 301):
            for var in (False, True):
                if var and win == 1:
                    continue  # window of one row: 0 / (0 + eps)
                assert _rel(ss.cmvnw(v, win, var), oracle.cmvnw(v, win, var)) <= RTOL, (shape, win, var)
    with pytest.raises(SpeechSauceError) as e:
        ss.cmvnw(_mat(0, (10, 3)), 4)
    assert e.value.status == 2


@pytest.mark.gpu
def test_derivatives_gpu(ss, oracle):
    for seed, shape in enumerate([(98, 13), (3, 1), (17, 40), (2, 2)]):
        f = _mat(30 + seed, shape)
        for dw in (1, 2, 9):
            assert _rel(ss.derivative_extraction(f, dw), oracle.derivative_extraction(f, dw)) <= RTOL, (shape, dw)
        cube = ss.extract_derivative_feature(f)
        assert cube.shape == shape + (3,)
        assert _rel(cube, oracle.extract_derivative_feature(f)) <= RTOL
    with pytest.raises(Exception):
        ss.derivative_extraction(_mat(0, (4, 4)), 0)
    with pytest.raises(TypeError):
        ss.cmvn(np.zeros((4, 4), np.float64))


@pytest.mark.gpu
def test_post_processing_on_device_block(ss, oracle):
    """The [batch, frames, ceps] block the MFCC kernel wrote, normalised and differentiated in place on the device."""
    import torch

    x = (np.random.default_rng(41).standard_normal((64, 16000)) * 0.1).astype(np.float32)
    feats = ss.mfcc_batch(torch.from_numpy(x).cuda(), 16000)
    assert feats.shape == (64, 98, 13)
    host = feats.cpu().numpy()
    a = ss.cmvn(feats, True).cpu().numpy()
    b = ss.cmvnw(feats, 31, True).cpu().numpy()
    c = ss.extract_derivative_feature(feats).cpu().numpy()
    d = ss.derivative_extraction(feats, 2).cpu().numpy()
    assert c.shape == (64, 98, 13, 3)
    for i in (0, 31, 63):
        assert _rel(a[i], oracle.cmvn(host[i], True)) <= RTOL
        assert _rel(b[i], oracle.cmvnw(host[i], 31, True)) <= RTOL
        assert _rel(c[i], oracle.extract_derivative_feature(host[i])) <= RTOL
        assert _rel(d[i], oracle.derivative_extraction(host[i], 2)) <= RTOL

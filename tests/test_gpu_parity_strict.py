"""Tightened parity metrics (round-2 review): the round-1 assert normalised every error by the largest value of the block,
which for MFCC is column 0 = ln(frame energy) (|.| up to 16 on quiet or silent frames) -- far larger than the cepstra.

Here, per BASELINE configuration and for the golden signals plus quiet clips (scale 1e-3, 1e-5) and a clip with silent
frames (tools/parity_report.py builds them; the same code writes profiles/r02/parity.json):
  * column 0 and columns 1.. are normalised SEPARATELY, each <= 1e-4 of its own maximum;
  * element-wise relative error over the elements with |want| > 1e-3 max|want| (columns 1.. for MFCC).  The cepstra are
    sums of ~40..256 ln() terms that cancel to a few per cent of their size, so f32 rounding alone costs a few 1e-4 here: the
    oracle's f32 port of the reference's own operation order misses the f64 oracle by 6.8e-4 on cfg1.  The HIP path has to
    stay within 1e-4 or within 1.5x of that port's error (worst case over the signal classes), whichever is larger --
    i.e. it may not be worse than the reference's own arithmetic -- and every case under an absolute cap of 2e-3.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

TOL = 1e-4


@pytest.fixture(scope="module")
def report(ss, oracle):
    import parity_report

    return parity_report.run(ss, oracle)


@pytest.mark.parametrize("cfg", ["cfg1", "cfg3", "cfg5"])
def test_columns_normalised_separately(report, cfg):
    for sig, m in report[cfg].items():
        assert m["max_norm"] <= TOL, (cfg, sig, m)
        if "col0_norm" in m:
            assert m["col0_norm"] <= TOL, (cfg, sig, m)
            assert m["rest_norm"] <= TOL, (cfg, sig, m)


@pytest.mark.parametrize("cfg", ["cfg1", "cfg3", "cfg5"])
def test_element_wise_relative_error(report, cfg):
    # which element is worst is a matter of rounding luck (the bound is rest_norm / 1e-3), so the comparison with the f32
    # port is made on the worst case over the signal classes, not signal by signal
    worst = max(m["elem_rel"] for m in report[cfg].values())
    port_worst = max(m["port_elem_rel"] for m in report[cfg].values())
    assert worst <= max(TOL, 1.5 * port_worst), (cfg, worst, port_worst)
    for sig, m in report[cfg].items():
        assert m["elem_rel"] <= 2e-3, (cfg, sig, m)


def test_quiet_and_silent_clips_are_covered(report):
    for cfg in report:
        assert {"quiet_1e-3", "quiet_1e-5", "silent_frames"} <= set(report[cfg])


def test_cfg2_batch_quiet_clips_and_silent_frames(ss, oracle):
    """The batch kernel on a batch that mixes levels: loud, 1e-3, 1e-5 and clips with digital silence, against the oracle with
    column 0 and the cepstra normalised separately."""
    import torch

    rng = np.random.default_rng(11)
    x = (rng.standard_normal((64, 16000)) * 0.1).astype(np.float32)
    x[16:32] *= 1e-3
    x[32:48] *= 1e-5
    x[48:, 4000:9000] = 0.0
    got = ss.mfcc_batch(torch.from_numpy(x).cuda(), 16000).cpu().numpy().astype(np.float64)
    p = oracle.make_params(sample_rate=16000)
    for b in range(0, 64, 3):
        want = oracle.mfcc(p, x[b])
        assert np.abs(got[b][:, 0] - want[:, 0]).max() <= TOL * np.abs(want[:, 0]).max(), b
        assert np.abs(got[b][:, 1:] - want[:, 1:]).max() <= TOL * np.abs(want[:, 1:]).max(), b
    # frames of digital silence: column 0 is exactly ln(f32::EPSILON) (functions.rs:66-71)
    silent = got[50][30:50, 0]
    np.testing.assert_allclose(silent, np.log(np.float64(np.float32(1.1920929e-7))), rtol=0, atol=2e-5)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mfcc-rust_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_c

    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def sslib():
    """The built C-ABI library (fails loudly if it is missing: there is no CPU fallback).  A clean checkout has no
    binaries (they are git-ignored), so the library is built here the way `__graft_entry__.build()` does when it is absent;
    hipcc cross-compiles for gfx950 without a GPU."""
    import subprocess

    from speechsauce_amd import _lib

    # always: make is incremental, and a library older than its sources must never be what the tests exercise
    subprocess.run(["make", "-C", os.path.join(ROOT, "mfcc-rust_amd", "csrc"), "-j4", "-s"], check=True)

    return _lib.lib()


@pytest.fixture(scope="session")
def sslab(sslib):
    """The LAB build of the library (same sources, -DSS_LAB=1; `make lab`): the only one that exports the process-wide test
    aids of include/speechsauce_amd_debug.h.  Tests that poison LDS, force a kernel build or inject the tile fault load it
    explicitly; everything else runs on the product library (`sslib`).  Loaded after the product library, never instead of it."""
    import subprocess

    from speechsauce_amd import _lib

    # always (incremental): the product and the lab library must come from the same sources -- a stale lab build with the same
    # ABI number would silently run old kernels in the kernel-variant, tile and fault tests
    subprocess.run(["make", "-C", os.path.join(ROOT, "mfcc-rust_amd", "csrc"), "-j4", "-s", "lab"], check=True)
    return _lib.lab()


@pytest.fixture(scope="session")
def ss():
    """The Python front on a real device (GPU tests only)."""
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import speechsauce_amd

    return speechsauce_amd

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


def _make(subprocess, targets, lib_path):
    """`make [targets]` in csrc, incremental.  A host that cannot build (no hipcc: a box that was handed prebuilt libraries) keeps the
    library it has -- with a warning if `make -q` says the sources are newer -- instead of failing every test that loads it; a
    missing library on such a host fails loudly in _lib.load()."""
    import shutil
    import warnings

    csrc = os.path.join(ROOT, "mfcc-rust_amd", "csrc")
    cmd = ["make", "-C", csrc, "-j4", "-s"] + targets
    if subprocess.run(cmd[:3] + ["-q"] + targets, capture_output=True).returncode == 0:
        return  # up to date
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if os.path.exists(lib_path) and not (os.path.exists(hipcc) or shutil.which("hipcc")):
        warnings.warn(f"{lib_path} is older than its sources and this host has no hipcc: testing the library as it is")
        return
    subprocess.run(cmd, check=True)


@pytest.fixture(scope="session")
def sslib():
    """The built C-ABI library (fails loudly if it is missing: there is no CPU fallback).  A clean checkout has no
    binaries (they are git-ignored), so the library is built here the way `__graft_entry__.build()` does when it is absent;
    hipcc cross-compiles for gfx950 without a GPU."""
    import subprocess

    from speechsauce_amd import _lib

    # always: make is incremental, and a library older than its sources must never be what the tests exercise
    _make(subprocess, [], _lib.LIB_PATH)

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
    _make(subprocess, ["lab"], _lib.LAB_LIB_PATH)
    return _lib.lab()


@pytest.fixture(scope="session")
def ss():
    """The Python front on a real device (GPU tests only)."""
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import speechsauce_amd

    return speechsauce_amd

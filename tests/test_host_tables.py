"""Host-side table builders under AddressSanitizer + UBSan (CPU only): tools/hosttest/fuzz_tables.cpp builds the LDS blocks of
every kernel for pseudo-random valid configurations and checks that each filter's weights reach exactly one (slot, lane) bit
for bit, that no lane reads past its P row and that the cosine rows are the host DCT table in the kernel's layout."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_table_builders_sanitized(tmp_path):
    exe = str(tmp_path / "fuzz_tables")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "mfcc-rust_amd", "csrc"), os.path.join(ROOT, "tools", "hosttest", "fuzz_tables.cpp"),
           os.path.join(ROOT, "mfcc-rust_amd", "csrc", "ss_host.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, timeout=300)
    for seed in ("12345", "777"):
        r = subprocess.run([exe, "1000", seed], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
        assert "all checks passed" in r.stdout

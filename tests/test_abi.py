"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/speechsauce_amd.h declares,
its host-side functions (sizes, tables, validation) agree with the oracle and the fixtures, and every
compute entry point FAILS LOUDLY without a HIP device (there is no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from common import CONFIGS, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "speechsauce_amd.h")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


DEBUG_HEADER = os.path.join(ROOT, "include", "speechsauce_amd_debug.h")


def _declared_symbols(headers=(HEADER,)):
    names = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported(sslib):
    from speechsauce_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(sslib, n), f"{n} declared in the header but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype in the Python front"
    assert sslib.ss_abi_version() == 7


def test_the_lab_library_carries_the_test_aids(sslab):
    """include/speechsauce_amd_debug.h is the LAB library's header: every symbol it declares is exported there (and, by the
    test below, by the lab library only)."""
    from speechsauce_amd import _lib

    names = [n for n in _declared_symbols((DEBUG_HEADER,)) if n.startswith("ss_debug_")]
    assert len(names) >= 5
    for n in names:
        assert hasattr(sslab, n) and n in _lib.LAB_PROTOTYPES, n
    assert sslab.ss_abi_version() == 7


def test_the_product_library_exports_only_the_documented_abi():
    """`nm -D` of the shipped library: every defined dynamic symbol is an ss_* entry point declared in
    include/speechsauce_amd.h -- no C++ internals, no lab switches, NO ss_debug_* test aid (those change kernel selection or
    inject faults process-wide: lab library only) -- and no environment knob name is compiled in."""
    import subprocess

    from speechsauce_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    declared = set(_declared_symbols())
    assert exported, "no dynamic symbols?"
    stray = [n for n in exported if n not in declared]
    assert not stray, f"exported but undocumented: {stray}"
    assert not [n for n in exported if n.startswith("ss_debug")], "a process-wide test aid is exported by the product library"
    blob = open(_lib.LIB_PATH, "rb").read()
    for knob in (b"SS_FORCE_GENERIC", b"SS_RES", b"SS_WAVES", b"SS_MEL_WAVES", b"SS_STFT_WAVES", b"SS_MEL_TILE", b"SS_HOST_CHUNK_MB", b"SS_HOST_SMALL_KB",
                 b"SS_DEBUG_TIMES", b"SS_DEBUG_ROWS", b"SS_POOL", b"SS_MEL_ROWS4"):
        assert knob + b"\0" not in blob, f"the product build still reads {knob.decode()}"


def test_params_struct_matches_header(sslib):
    from speechsauce_amd._lib import SsParams

    p = SsParams()
    assert sslib.ss_params_default(C.byref(p), 16000) == 0
    assert p.struct_size == C.sizeof(SsParams) == 80
    # SpeechConfigBuilder::new defaults, config.rs:35-47
    assert (p.sample_rate, p.fft_points, p.num_cepstral, p.num_filters, p.dc_elimination) == (16000, 512, 13, 40, 1)
    assert p.frame_length == pytest.approx(0.02) and p.frame_stride == pytest.approx(0.01)
    assert p.low_frequency == 0.0 and p.high_frequency == 8000.0
    # reference-mode switches
    assert (p.framing, p.spectrum_exponent, p.dct_norm, p.mfcc_window, p.preemph_shift) == (0, 1, 0, 0, 1)
    assert (p.mel_scale, p.mel_norm, p.pad_mode) == (0, 0, 0)
    assert p.dct2_gain == 2.0 and p.preemph_coef == 0.0
    p.struct_size = 4
    assert sslib.ss_params_validate(C.byref(p)) == 3  # SS_ERR_ARG: ABI guard


def test_host_sizes_match_oracle(sslib, oracle):
    from speechsauce_amd import make_params

    for name, kw in CONFIGS.items():
        p, po = make_params(**kw), oracle.make_params(**kw)
        fl, st = C.c_size_t(), C.c_size_t()
        assert sslib.ss_frame_sizes(C.byref(p), C.byref(fl), C.byref(st)) == 0
        assert (fl.value, st.value) == oracle.frame_sizes(po)
        for n in (fl.value + st.value, 8191, 16000, 44100, 1_000_000):
            t = C.c_size_t()
            assert sslib.ss_num_frames(C.byref(p), n, C.byref(t)) == 0
            assert t.value == oracle.num_frames(po, n)
    p = make_params()
    t = C.c_size_t()
    for n in (0, 319, 320, 479):
        assert sslib.ss_num_frames(C.byref(p), n, C.byref(t)) == 1  # SS_ERR_SHORT_SIGNAL
    p3, po3 = make_params(**CONFIGS["cfg3"]), oracle.make_params(**CONFIGS["cfg3"])
    hop, npad, wn = C.c_size_t(), C.c_size_t(), C.c_float()
    assert sslib.ss_stft_sizes(C.byref(p3), C.byref(hop), C.byref(npad), C.byref(wn)) == 0
    assert (hop.value, npad.value, wn.value) == oracle.stft_sizes(po3)
    r, rr = C.c_size_t(), C.c_size_t()
    for n in (1, 511, 512, 513, 5000, 16000):
        assert sslib.ss_stft_rows(C.byref(p3), n, C.byref(r), C.byref(rr)) == 0
        assert (r.value, rr.value) == oracle.stft_rows(po3, n)
    # the default config cannot run the STFT path (functions.rs:136 usize underflow) -> error, not a panic
    assert sslib.ss_stft_sizes(C.byref(p), C.byref(hop), C.byref(npad), C.byref(wn)) == 2
    assert b"fft_points >= 2" in sslib.ss_last_error_string()


@pytest.mark.parametrize("name", ["cfg1", "cfg3", "cfg5"])
def test_host_filterbank_matches_fixture_and_oracle(sslib, oracle, name):
    from speechsauce_amd import make_params

    g = load_golden()
    p = make_params(**CONFIGS[name])
    M, F = p.num_filters, p.fft_points // 2 + 1
    fb = np.empty((M, F), np.float32)
    idx = np.empty(M + 2, np.int32)
    assert sslib.ss_filterbank(C.byref(p), fb.ctypes.data, idx.ctypes.data) == 0
    np.testing.assert_array_equal(idx, g[f"{name}/fb_idx"])
    ofb, oidx = oracle.filterbank(oracle.make_params(**CONFIGS[name]))
    np.testing.assert_array_equal(fb, ofb)
    w = np.empty(p.fft_points, np.float32)
    assert sslib.ss_vorbis_window(p.fft_points, w.ctypes.data) == 0
    np.testing.assert_array_equal(w, g[f"{name}/vorbis_window"])


def test_validation_mirrors_reference_panics(sslib):
    from speechsauce_amd import make_params

    bad = [
        (dict(high_frequency=8000.5), 2),      # feature.rs:47-50 assert
        (dict(low_frequency=-1.0), 2),         # feature.rs:51 assert
        (dict(num_cepstral=41), 2),            # feature.rs:133 slice panic
        (dict(frame_length=0.04), 2),          # ndfft_r2c size assert, processing.rs:146-164
        (dict(fft_points=2731, frame_length=0.02), 5),  # valid in the reference, unsupported here: SS_ERR_UNSUPPORTED
        (dict(fft_points=16384), 5),           # (non powers of two run a chirp-z transform up to 2730 points)
        (dict(spectrum_exponent=3), 2),
        (dict(frame_stride=0.0), 2),
    ]
    for kw, code in bad:
        p = make_params(**kw)
        assert sslib.ss_params_validate(C.byref(p)) == code, kw
        assert sslib.ss_last_error_string() != b""
    assert sslib.ss_params_validate(C.byref(make_params())) == 0
    assert sslib.ss_params_validate(C.byref(make_params(fft_points=400))) == 0  # 25 ms at 16 kHz: chirp-z
    for s in range(7):
        assert sslib.ss_status_string(s) not in (b"", b"unknown status")


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device failure mode")
def test_fails_loudly_without_a_device(sslib):
    """There is no CPU compute path: config creation and every hot-path call must error out."""
    import speechsauce_amd as ss
    from speechsauce_amd import SpeechSauceError, make_params

    cfg = C.c_void_p()
    p = make_params()
    assert sslib.ss_config_create(C.byref(p), C.byref(cfg)) == 4  # SS_ERR_HIP
    assert not cfg.value
    assert b"no CPU fallback" in sslib.ss_last_error_string()
    x = np.zeros(16000, np.float32)
    with pytest.raises(SpeechSauceError) as e:
        ss.mfcc(x, 16000)
    assert e.value.status == 4
    with pytest.raises(SpeechSauceError):
        ss.preemphasis(x)
    n = C.c_int(-1)
    sslib.ss_device_count(C.byref(n))
    assert n.value == 0


def test_python_front_argument_rules():
    """py-speechsauce/src/lib.rs:170,182,199-201: float32 only, mfcc 1-D, mel_spectrogram 1-D or 2-D."""
    import speechsauce_amd as ss

    with pytest.raises(TypeError):
        ss.mfcc(np.zeros(16000, np.float64), 16000)
    with pytest.raises(ValueError):
        ss.mfcc(np.zeros((2, 16000), np.float32), 16000)
    with pytest.raises(ValueError):
        ss.mel_spectrogram(np.zeros((2, 2, 1600), np.float32), 16000)
    assert set(ss.__all__) >= {"mfcc", "mel_spectrogram", "preemphasis"}
    import inspect

    sig = inspect.signature(ss.mfcc)
    assert list(sig.parameters)[:10] == ["signal", "sampling_frequency", "frame_length", "frame_stride", "num_cepstral",
                                         "num_filters", "fft_length", "low_frequency", "high_frequency", "dc_elimination"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["frame_length"], d["frame_stride"], d["num_cepstral"], d["num_filters"], d["fft_length"],
            d["low_frequency"], d["high_frequency"], d["dc_elimination"]) == (0.020, 0.01, 13, 40, 512, 0, None, True)


def test_cpp_mirror_header_compiles_and_fails_loudly(tmp_path, sslib):
    """include/speechsauce_amd.hpp (C++ mirror of the Rust API) builds against the library with plain g++;
    without a device it throws speechsauce::Error(SS_ERR_HIP) instead of computing anything on the CPU."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src = tmp_path / "t.cpp"
    src.write_text(
        '#include "speechsauce_amd.hpp"\n'
        "int main() {\n"
        "  try { speechsauce::SpeechConfig cfg; std::vector<float> x(16000, 0.f);\n"
        "        auto m = speechsauce::mfcc(x, cfg); return m.rows == 98 && m.cols == 13 ? 0 : 3; }\n"
        "  catch (const speechsauce::Error &e) { return e.status == SS_ERR_HIP ? 10 : 4; }\n"
        "}\n")
    libdir = os.path.join(ROOT, "mfcc-rust_amd", "lib")
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lspeechsauce_amd",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    rc = subprocess.run([str(exe)]).returncode
    assert rc == (0 if _has_gpu() else 10)


def test_the_front_can_be_pointed_at_the_lab_build_and_back(sslib, sslab):
    """speechsauce_amd._lib.use_library: inside the block the front runs on the given build (the lab library, for tests that
    need its aids), configs memoised by the front are dropped on the way in and out (a config belongs to the library that
    created it), and the product library is back afterwards."""
    import speechsauce_amd as ss
    from speechsauce_amd import _lib

    assert _lib.lib() is sslib and sslib is not sslab
    with _lib.use_library(sslab) as h:
        assert h is sslab and _lib.lib() is sslab
        assert ss._get_speech_config.cache_info().currsize == 0
    assert _lib.lib() is sslib
    assert _lib.lab() is sslab  # loaded once

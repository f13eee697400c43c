"""ctypes driver for oracle/libss_oracle.so (TEST INFRASTRUCTURE ONLY -- see ss_oracle.h).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libss_oracle.so")

ORC_OK, ORC_ERR_SHORT_SIGNAL, ORC_ERR_BAD_CONFIG, ORC_ERR_ARG = 0, 1, 2, 3
FRAMING = {"contract": 0, "literal": 1, "center": 2, "padded": 3}
MEL_SCALE = {"reference": 0, "slaney": 1, "htk": 2}
MEL_NORM = {"none": 0, "slaney": 1}
PAD_MODE = {"reflect": 0, "constant": 1}
DCT_NORM = {"reference": 0, "ortho": 1}
WINDOW = {"rect": 0, "hann": 1, "vorbis": 2}


class OrcParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("sample_rate", C.c_uint32),
        ("fft_points", C.c_uint32),
        ("frame_length", C.c_float),
        ("frame_stride", C.c_float),
        ("num_cepstral", C.c_uint32),
        ("num_filters", C.c_uint32),
        ("low_frequency", C.c_float),
        ("high_frequency", C.c_float),
        ("dc_elimination", C.c_int32),
        ("framing", C.c_int32),
        ("spectrum_exponent", C.c_int32),
        ("dct_norm", C.c_int32),
        ("dct2_gain", C.c_float),
        ("mfcc_window", C.c_int32),
        ("preemph_coef", C.c_float),
        ("preemph_shift", C.c_int32),
        ("mel_scale", C.c_int32),
        ("mel_norm", C.c_int32),
        ("pad_mode", C.c_int32),
    ]


class OracleError(RuntimeError):
    def __init__(self, code: int):
        super().__init__(f"oracle error {code}")
        self.code = code


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ss_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libss_oracle.so"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def make_params(sample_rate=16000, fft_points=512, frame_length=0.02, frame_stride=0.01, num_cepstral=13,
                num_filters=40, low_frequency=0.0, high_frequency=None, dc_elimination=True,
                framing="contract", spectrum_exponent=1, dct_norm="reference", dct2_gain=2.0,
                mfcc_window="rect", preemph_coef=0.0, preemph_shift=1, mel_scale="reference", mel_norm="none",
                pad_mode="reflect") -> OrcParams:
    p = OrcParams()
    lib().orc_params_default(C.byref(p), C.c_uint32(sample_rate))
    p.fft_points = fft_points
    p.frame_length = frame_length
    p.frame_stride = frame_stride
    p.num_cepstral = num_cepstral
    p.num_filters = num_filters
    p.low_frequency = low_frequency
    p.high_frequency = sample_rate / 2.0 if high_frequency is None else high_frequency
    p.dc_elimination = int(bool(dc_elimination))
    p.framing = FRAMING[framing]
    p.spectrum_exponent = spectrum_exponent
    p.dct_norm = DCT_NORM[dct_norm]
    p.dct2_gain = dct2_gain
    p.mfcc_window = WINDOW[mfcc_window]
    p.preemph_coef = preemph_coef
    p.preemph_shift = preemph_shift
    p.mel_scale = MEL_SCALE[mel_scale]
    p.mel_norm = MEL_NORM[mel_norm]
    p.pad_mode = PAD_MODE[pad_mode]
    return p


def _chk(rc):
    if rc != ORC_OK:
        raise OracleError(rc)


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def frame_sizes(p):
    a, b = C.c_size_t(), C.c_size_t()
    _chk(lib().orc_frame_sizes(C.byref(p), C.byref(a), C.byref(b)))
    return a.value, b.value


def num_frames(p, n):
    t = C.c_size_t()
    _chk(lib().orc_num_frames(C.byref(p), C.c_size_t(n), C.byref(t)))
    return t.value


def num_frames_padded(p, n):
    t = C.c_size_t()
    _chk(lib().orc_num_frames_padded(C.byref(p), C.c_size_t(n), C.byref(t)))
    return t.value


def stft_sizes(p):
    h, npad, wn = C.c_size_t(), C.c_size_t(), C.c_float()
    _chk(lib().orc_stft_sizes(C.byref(p), C.byref(h), C.byref(npad), C.byref(wn)))
    return h.value, npad.value, wn.value


def stft_rows(p, n):
    r, rr = C.c_size_t(), C.c_size_t()
    _chk(lib().orc_stft_rows(C.byref(p), C.c_size_t(n), C.byref(r), C.byref(rr)))
    return r.value, rr.value


def vorbis_window(n):
    w = np.empty(n, dtype=np.float32)
    lib().orc_vorbis_window(C.c_size_t(n), _ptr(w, C.c_float))
    return w


def hann_window(n):
    w = np.empty(n, dtype=np.float32)
    lib().orc_hann_window(C.c_size_t(n), _ptr(w, C.c_float))
    return w


def filterbank(p):
    M, F = p.num_filters, p.fft_points // 2 + 1
    fb = np.empty((M, F), dtype=np.float32)
    idx = np.empty(M + 2, dtype=np.int32)
    _chk(lib().orc_filterbank(C.byref(p), _ptr(fb, C.c_float), _ptr(idx, C.c_int32)))
    return fb, idx


def power_spectrum(p, x):
    x = _f32(x)
    T, F = num_frames(p, x.size), p.fft_points // 2 + 1
    out = np.empty((T, F), dtype=np.float64)
    _chk(lib().orc_power_spectrum(C.byref(p), _ptr(x, C.c_float), C.c_size_t(x.size), _ptr(out, C.c_double)))
    return out


def stack_frames(p, x):
    x = _f32(x)
    T = num_frames(p, x.size)
    flen, _ = frame_sizes(p)
    out = np.empty((T, flen), dtype=np.float64)
    _chk(lib().orc_stack_frames(C.byref(p), _ptr(x, C.c_float), C.c_size_t(x.size), _ptr(out, C.c_double)))
    return out


def power_spectrum_frames(frames, fft_points):
    f = np.ascontiguousarray(frames, dtype=np.float32)
    out = np.empty((f.shape[0], fft_points // 2 + 1), dtype=np.float64)
    _chk(lib().orc_power_spectrum_frames(_ptr(f, C.c_float), C.c_size_t(f.shape[0]), C.c_size_t(f.shape[1]), C.c_size_t(fft_points),
                                         _ptr(out, C.c_double)))
    return out


def mfe(p, x):
    x = _f32(x)
    T = num_frames(p, x.size)
    feat = np.empty((T, p.num_filters), dtype=np.float64)
    en = np.empty(T, dtype=np.float64)
    _chk(lib().orc_mfe(C.byref(p), _ptr(x, C.c_float), C.c_size_t(x.size), _ptr(feat, C.c_double), _ptr(en, C.c_double)))
    return feat, en


def mfcc(p, x):
    x = _f32(x)
    T = num_frames(p, x.size)
    out = np.empty((T, p.num_cepstral), dtype=np.float64)
    _chk(lib().orc_mfcc(C.byref(p), _ptr(x, C.c_float), C.c_size_t(x.size), _ptr(out, C.c_double)))
    return out


def stft(p, x):
    x = np.atleast_2d(_f32(x))
    ch, n = x.shape
    R, _ = stft_rows(p, n)
    F = p.fft_points // 2 + 1
    out = np.empty((ch, R, F, 2), dtype=np.float64)
    _chk(lib().orc_stft(C.byref(p), _ptr(x, C.c_float), C.c_size_t(ch), C.c_size_t(n), _ptr(out, C.c_double)))
    return out[..., 0] + 1j * out[..., 1]


def mel_spectrogram(p, x):
    one_d = np.asarray(x).ndim == 1
    x = np.atleast_2d(_f32(x))
    ch, n = x.shape
    R, _ = stft_rows(p, n)
    out = np.empty((ch, p.num_filters, R), dtype=np.float64)
    _chk(lib().orc_mel_spectrogram(C.byref(p), _ptr(x, C.c_float), C.c_size_t(ch), C.c_size_t(n), _ptr(out, C.c_double)))
    return out[0] if one_d else out


def preemphasis(x, shift=1, cof=0.98):
    x = _f32(x)
    y = np.empty(x.size, dtype=np.float64)
    _chk(lib().orc_preemphasis(_ptr(x, C.c_float), C.c_size_t(x.size), C.c_long(shift), C.c_float(cof), _ptr(y, C.c_double)))
    return y


# ---- post-processing on the feature matrix (processing.rs:222-371, feature.rs:253-269) ----

def cmvn(vec, variance_normalization=False):
    v = np.ascontiguousarray(vec, dtype=np.float32)
    out = np.empty(v.shape, dtype=np.float64)
    _chk(lib().orc_cmvn(_ptr(v, C.c_float), C.c_size_t(v.shape[0]), C.c_size_t(v.shape[1]), C.c_int(int(variance_normalization)),
                        _ptr(out, C.c_double)))
    return out


def cmvnw(vec, win_size=301, variance_normalization=False):
    v = np.ascontiguousarray(vec, dtype=np.float32)
    out = np.empty(v.shape, dtype=np.float64)
    _chk(lib().orc_cmvnw(_ptr(v, C.c_float), C.c_size_t(v.shape[0]), C.c_size_t(v.shape[1]), C.c_size_t(win_size),
                         C.c_int(int(variance_normalization)), _ptr(out, C.c_double)))
    return out


def derivative_extraction(feat, delta_windows):
    f = np.ascontiguousarray(feat, dtype=np.float64)
    out = np.empty(f.shape, dtype=np.float64)
    _chk(lib().orc_derivative_extraction(_ptr(f, C.c_double), C.c_size_t(f.shape[0]), C.c_size_t(f.shape[1]),
                                         C.c_size_t(delta_windows), _ptr(out, C.c_double)))
    return out


def extract_derivative_feature(feat):
    f = np.ascontiguousarray(feat, dtype=np.float32)
    out = np.empty(f.shape + (3,), dtype=np.float64)
    _chk(lib().orc_extract_derivative_feature(_ptr(f, C.c_float), C.c_size_t(f.shape[0]), C.c_size_t(f.shape[1]),
                                              _ptr(out, C.c_double)))
    return out


# ---- reference-shaped f32 port (timed CPU baseline) ----

_native = None
_native_flags = None


def _host_cpu_id():
    """What -march=native depends on: the CPU model and its feature flags (first processor of /proc/cpuinfo)."""
    import hashlib

    try:
        first = open("/proc/cpuinfo").read().split("\n\n")[0]
        keep = [ln for ln in first.splitlines() if ln.split(":")[0].strip() in ("vendor_id", "cpu family", "model", "model name", "stepping", "flags")]
        return hashlib.sha256("\n".join(keep).encode()).hexdigest()[:16]
    except Exception:
        return "unknown"


def native_port():
    """The port built for THIS host (`make native`: -O3 -march=native, no fast-math, no contraction) -- what bench.py's
    cpu_baseline times.  Built on the machine that runs it, on first use: `make native` is incremental (nothing is rebuilt while the
    source is older than the library), and a library built on ANOTHER CPU model (oracle/_native/host.txt differs) is rebuilt; the
    Makefile writes the new library under a temporary name and moves it into place, so a concurrent process never maps a
    half-written file.  (None, reason) when the build fails (the caller then times the portable -O2 build and says so).  Only
    port_* are taken from this library; the checker stays libss_oracle.so."""
    global _native, _native_flags
    if _native is None:
        nd = os.path.join(_HERE, "_native")
        so, host_file, host = os.path.join(nd, "libss_oracle_native.so"), os.path.join(nd, "host.txt"), _host_cpu_id()
        try:
            stale = os.path.exists(so) and (not os.path.exists(host_file) or open(host_file).read().strip() != host)
            subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if stale else []) + ["native"], check=True, capture_output=True, timeout=300)
            tmp = host_file + f".tmp.{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(host + "\n")
            os.replace(tmp, host_file)
            _native = C.CDLL(so)
            _native_flags = open(os.path.join(nd, "flags.txt")).read().strip()
        except Exception as e:  # no compiler / read-only tree: fall back, loudly, in the caller's report
            _native, _native_flags = False, f"native build failed: {e!r}"
    return (_native or None), _native_flags


def port_mfcc(p, x, from_lib=None):
    x = _f32(x)
    T = num_frames(p, x.size)
    out = np.empty((T, p.num_cepstral), dtype=np.float32)
    _chk((from_lib or lib()).port_mfcc_f32(C.byref(p), _ptr(x, C.c_float), C.c_size_t(x.size), _ptr(out, C.c_float)))
    return out


def port_mfe(p, x):
    x = _f32(x)
    T = num_frames(p, x.size)
    feat = np.empty((T, p.num_filters), dtype=np.float32)
    en = np.empty(T, dtype=np.float32)
    _chk(lib().port_mfe_f32(C.byref(p), _ptr(x, C.c_float), C.c_size_t(x.size), _ptr(feat, C.c_float), _ptr(en, C.c_float)))
    return feat, en


def port_mel_spectrogram(p, x, from_lib=None):
    one_d = np.asarray(x).ndim == 1
    x = np.atleast_2d(_f32(x))
    ch, n = x.shape
    R, _ = stft_rows(p, n)
    out = np.empty((ch, p.num_filters, R), dtype=np.float32)
    _chk((from_lib or lib()).port_mel_spectrogram_f32(C.byref(p), _ptr(x, C.c_float), C.c_size_t(ch), C.c_size_t(n), _ptr(out, C.c_float)))
    return out[0] if one_d else out

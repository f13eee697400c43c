"""numpy twin of the C oracle (TEST INFRASTRUCTURE ONLY -- see oracle/ss_oracle.h).

An independent restatement of the reference's hot path in numpy (f64 via ``np.fft``), used to
cross-check ``oracle/ss_oracle.c`` and to generate the committed fixtures under ``tests/golden/``
(``tests/golden/make_golden.py``).  Nothing in the product imports this module.

"parity unpinned": the reference (Rust) cannot be built here and has no golden vectors; the
FFT sign/normalisation and the DCT-II gain (``DCT2_GAIN = 2``) follow the published conventions of
the un-vendored crates (rustfft / ndrustfft ^0.4).  See ss_oracle.h for the full statement.

File:line citations are relative to the reference checkout (``speechsauce/src/...``).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math
from dataclasses import dataclass

import numpy as np

DCT2_GAIN = 2.0
EPS_F32 = float(np.finfo(np.float32).eps)  # functions.rs:70

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]
f32 = np.float32


@dataclass
class Params:
    """Mirrors SpeechConfig::new's nine arguments (config.rs:140-150) plus the SURVEY section-0 switches."""

    sample_rate: int = 16000
    fft_points: int = 512
    frame_length: float = 0.02
    frame_stride: float = 0.01
    num_cepstral: int = 13
    num_filters: int = 40
    low_frequency: float = 0.0
    high_frequency: float | None = None
    dc_elimination: bool = True
    framing: str = "contract"  # or "literal" (processing.rs:110-120 as written)
    spectrum_exponent: int = 1
    dct_norm: str = "reference"  # or "ortho"
    dct2_gain: float = DCT2_GAIN
    mfcc_window: str = "rect"  # "hann" | "vorbis"
    preemph_coef: float = 0.0
    preemph_shift: int = 1
    # librosa-compatible variants; the defaults are reference mode
    mel_scale: str = "reference"  # "slaney" | "htk": librosa.filters.mel, triangles in Hz at the rfft bin frequencies
    mel_norm: str = "none"  # "slaney": area normalisation 2 / (f[m+2] - f[m])
    pad_mode: str = "reflect"  # framing == "center": np.pad mode of the flen/2 samples on either side ("constant" = zeros)

    def high(self) -> float:
        return float(self.sample_rate) / 2.0 if self.high_frequency is None else float(self.high_frequency)


def _round_half_away(v: np.float32) -> int:
    return int(math.floor(float(v) + 0.5)) if v >= 0 else -int(math.floor(-float(v) + 0.5))


def frame_sizes(p: Params) -> tuple[int, int]:
    """processing.rs:77-78"""
    flen = _round_half_away(f32(p.sample_rate) * f32(p.frame_length))
    step = _round_half_away(f32(p.sample_rate) * f32(p.frame_stride))
    return flen, step


def num_frames(p: Params, n: int) -> int:
    """processing.rs:101 (zero_padding=false); f32 division then floor."""
    flen, step = frame_sizes(p)
    if p.framing == "center":  # librosa center=True
        if n == 0 or (p.pad_mode == "reflect" and n <= flen // 2):
            raise ValueError("signal too short for centred frames")
        return 1 + n // step
    if n < flen:
        raise ValueError("signal shorter than one frame")
    t = int(math.floor(float(f32(n - flen) / f32(step))))
    if t == 0:
        raise ValueError("zero frames")
    return t


def num_frames_padded(p: Params, n: int) -> int:
    """processing.rs:91-92 (zero_padding=true)"""
    flen, step = frame_sizes(p)
    return int(math.ceil(float(f32(n - flen) / f32(step))))


def vorbis_window(n: int) -> np.ndarray:
    """config.rs:151-160"""
    i = np.arange(n, dtype=np.float64)
    s = np.sin(0.5 * np.pi * (i + 0.5) / (n // 2))
    return np.sin(0.5 * np.pi * s * s).astype(np.float32)


def hann_window(n: int) -> np.ndarray:
    """functions.rs:349-357 (commented out in the reference; an option here)"""
    i = np.arange(n, dtype=np.float64)
    return (0.5 * (1.0 - np.cos(2.0 * np.pi * i / n))).astype(np.float32)


def filterbank(p: Params) -> tuple[np.ndarray, np.ndarray]:
    """feature.rs:36-90 + functions.rs:19-21,36-60, f32 with glibc logf/expf."""
    M, F = p.num_filters, p.fft_points // 2 + 1
    if p.mel_scale != "reference":
        return _filterbank_librosa(p), np.zeros(M + 2, dtype=np.int64)
    sr = f32(p.sample_rate)

    def mel(f):
        return f32(1127.0) * f32(_libm.logf(f32(1.0) + f32(f) / f32(700.0)))

    def hz(m):
        return f32(700.0) * (f32(_libm.expf(f32(m) / f32(1127.0))) - f32(1.0))

    lo, hi = mel(f32(p.low_frequency)), mel(f32(p.high()))
    step = (hi - lo) / f32(M + 1)
    idx = np.zeros(M + 2, dtype=np.int64)
    for i in range(M + 2):
        v = f32(F + 1) * hz(lo + step * f32(i)) / sr
        idx[i] = int(v) if v > 0 else 0
    fb = np.zeros((M, F), dtype=np.float32)
    for i in range(M):
        l, m, r = (f32(v) for v in idx[i : i + 3])
        for x in range(int(idx[i]), int(idx[i + 2]) + 1):
            xf, v = f32(x), f32(0.0)
            if l <= xf < r:
                with np.errstate(invalid="ignore", divide="ignore"):
                    if xf <= m:
                        v = (xf - l) / (m - l)
                    if m <= xf:
                        v = (r - xf) / (r - m)
            fb[i, x] = v
    return fb, idx


def _filterbank_librosa(p: Params) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk, norm) written the way librosa writes it (vectorised
    ramps / np.subtract.outer), float64 arithmetic, float32 result."""
    M, n_fft = p.num_filters, p.fft_points
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        if p.mel_scale == "htk":
            return 2595.0 * np.log10(1.0 + f / 700.0)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        if p.mel_scale == "htk":
            return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.fft.rfftfreq(n_fft, 1.0 / p.sample_rate)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(p.low_frequency), hz_to_mel(p.high()), M + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0, np.minimum(lower, upper))
    if p.mel_norm == "slaney":
        weights *= (2.0 / (mel_f[2 : M + 2] - mel_f[:M]))[:, None]
    return weights.astype(np.float32)


def preemphasis(x: np.ndarray, shift: int = 1, cof: float = 0.98) -> np.ndarray:
    """processing.rs:31-53: np.roll semantics."""
    x = np.asarray(x, dtype=np.float64)
    return x - float(f32(cof)) * np.roll(x, shift)


def _signal(p: Params, x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32).astype(np.float64)
    if p.preemph_coef != 0.0:
        x = x - float(f32(p.preemph_coef)) * np.roll(x, p.preemph_shift)
    return x


def power_spectrum(p: Params, x: np.ndarray) -> np.ndarray:
    """processing.rs:65-181 (contract framing D1, or the literal chunk copy)."""
    flen, step = frame_sizes(p)
    T = num_frames(p, len(x))
    N = p.fft_points
    xs = _signal(p, x)
    frames = np.zeros((T, N))
    if p.framing == "literal":
        if T <= 2:
            frames[:, : flen & ~1] = xs[: flen & ~1]
    elif p.framing == "center":  # librosa center=True: np.pad then plain framing
        xp = np.pad(xs, flen // 2, mode="reflect" if p.pad_mode == "reflect" else "constant")
        for t in range(T):
            frames[t, :flen] = xp[t * step : t * step + flen]
    else:
        for t in range(T):
            frames[t, :flen] = xs[t * step : t * step + flen]
    if p.mfcc_window == "hann":
        frames[:, :flen] *= hann_window(flen).astype(np.float64)
    elif p.mfcc_window == "vorbis":
        frames[:, :flen] *= vorbis_window(flen).astype(np.float64)
    mag = np.abs(np.fft.rfft(frames, axis=1))
    inv_n = float(f32(1.0) / f32(N))
    return inv_n * (mag**2 if p.spectrum_exponent == 2 else mag)


def mfe(p: Params, x: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """feature.rs:200-233"""
    P = power_spectrum(p, x)
    fb, _ = filterbank(p)
    e = P.sum(axis=1)
    e[e == 0.0] = EPS_F32
    feat = P @ fb.astype(np.float64).T
    feat[feat == 0.0] = EPS_F32
    return feat, e


def mfcc(p: Params, x: np.ndarray) -> np.ndarray:
    """feature.rs:99-148"""
    feat, e = mfe(p, x)
    T, M = feat.shape
    C = p.num_cepstral
    k = np.arange(M)[:, None]
    m = np.arange(M)[None, :]
    basis = np.cos(np.pi * k * (2 * m + 1) / (2.0 * M))  # [k, m]
    y = float(f32(p.dct2_gain)) * (np.log(feat) @ basis.T)
    if p.dct_norm == "ortho":
        y[:, 0] *= 1.0 / math.sqrt(4.0 * M)
        y[:, 1:] *= 1.0 / math.sqrt(2.0 * M)
    else:
        n = f32(T * M)
        y[0, 0] *= float(f32(1.0) / np.sqrt(f32(4.0) * n))
        y[:, 1:] *= float(f32(1.0) / np.sqrt(f32(2.0) * n))
    out = y[:, :C].copy()
    if p.dc_elimination:
        out[:, 0] = np.log(e)
    return out


def stft_sizes(p: Params) -> tuple[int, int, float]:
    """config.rs:154,178; functions.rs:96"""
    W = p.fft_points
    H = int(f32(p.frame_length) * f32(p.sample_rate))
    if H == 0 or W < 2 * H:
        raise ValueError("STFT path needs fft_points >= 2*frame_size")
    wnorm = f32(1.0) / (f32(W * W) / f32(2 * H))
    return H, W // H - 1, float(wnorm)


def stft(p: Params, x: np.ndarray) -> np.ndarray:
    """functions.rs:86-170 with zero initial state per channel (D3). x: [ch, n] -> [ch, R, F] complex."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float32)).astype(np.float64)
    ch, n = x.shape
    H, n_pad, wnorm = stft_sizes(p)
    W = p.fft_points
    R = int(math.ceil(float(f32(n) / f32(H))))
    win = vorbis_window(W).astype(np.float64)
    out = np.zeros((ch, R, W // 2 + 1), dtype=np.complex128)
    padded = np.concatenate([np.zeros((ch, W)), x, np.zeros((ch, W + H))], axis=1)
    for r in range(max(R - n_pad, 0)):
        start = (r + n_pad + 1) * H - W + W  # offset by the W leading zeros
        out[:, r, :] = np.fft.rfft(padded[:, start : start + W] * win, axis=1) * wnorm
    return out


def mel_spectrogram(p: Params, x: np.ndarray) -> np.ndarray:
    """feature.rs:151-174 (1-D input == one channel, D2). Returns [ch, M, R] (or [M, R] for 1-D)."""
    one_d = np.asarray(x).ndim == 1
    S = stft(p, x)
    P = np.abs(S) ** 2
    fb, _ = filterbank(p)
    out = np.einsum("ntf,mf->nmt", P, fb.astype(np.float64))
    return out[0] if one_d else out


# ---- post-processing on the feature matrix, written with np.pad (the semantics the reference quotes, util.rs:108-124) ----

def cmvn(vec: np.ndarray, variance_normalization: bool = False) -> np.ndarray:
    """processing.rs:265-300."""
    v = np.asarray(vec, dtype=np.float64)
    ms = v - v.mean(axis=0, keepdims=True)
    if variance_normalization:
        return ms / (ms.std(axis=0, keepdims=True) + 2.0 ** -30)
    return ms


def cmvnw(vec: np.ndarray, win_size: int = 301, variance_normalization: bool = False) -> np.ndarray:
    """processing.rs:315-371."""
    if win_size % 2 != 1:
        raise ValueError("Windows size must be odd!")
    v = np.asarray(vec, dtype=np.float64)
    pad = (win_size - 1) // 2
    vp = np.pad(v, ((pad, pad), (0, 0)), "symmetric")
    ms = np.stack([v[i] - vp[i:i + win_size].mean(axis=0) for i in range(v.shape[0])])
    if not variance_normalization:
        return ms
    mp = np.pad(ms, ((pad, pad), (0, 0)), "symmetric")
    return np.stack([ms[i] / (mp[i:i + win_size].std(axis=0) + 2.0 ** -30) for i in range(v.shape[0])])


def derivative_extraction(feat: np.ndarray, delta_windows: int) -> np.ndarray:
    """processing.rs:222-254 (literal: Range * f[c + Range] - f[c - Range], along the feature axis)."""
    f = np.asarray(feat, dtype=np.float64)
    cols = f.shape[1]
    fp = np.pad(f, ((0, 0), (delta_windows, delta_windows)), "edge")
    acc, scale, off = np.zeros_like(f), 0.0, delta_windows
    for i in range(delta_windows):
        r = i + 1
        acc += fp[:, off + r:off + r + cols] * r - fp[:, off - r:off - r + cols]
        scale += 2.0 * r ** 2
    return acc / scale


def extract_derivative_feature(feat: np.ndarray) -> np.ndarray:
    """feature.rs:253-269."""
    f = np.asarray(feat, dtype=np.float64)
    d1 = derivative_extraction(f, 2)
    return np.stack([f, d1, derivative_extraction(d1, 2)], axis=2)

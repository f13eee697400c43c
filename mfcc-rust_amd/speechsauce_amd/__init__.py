"""speechsauce_amd -- Python front of the MI355X-native MFCC / mel-spectrogram hot path.

Drop-in for the hot-path functions of the reference's ``speechsauce`` package
(py-speechsauce/speechsauce/__init__.py:37-132): the same names, keyword arguments, defaults,
dtype rule (float32 only) and output shapes, served by hand-written HIP kernels through the C ABI
in ``include/speechsauce_amd.h``.  ``import speechsauce_amd as speechsauce`` is the intended use.

Inputs may be numpy arrays (host path: H2D, kernels, D2H inside the library) or torch tensors on
a ROCm device (zero-copy device path on torch's current stream; the result is a torch tensor).
There is no CPU compute path: without the built library and a HIP device these functions raise.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Optional

import numpy as np

from . import _lib
from ._lib import SpeechSauceError, SsParams, make_params  # noqa: F401

__all__ = ["mfcc", "mel_spectrogram", "preemphasis", "cmvn", "cmvnw", "derivative_extraction", "extract_derivative_feature",
           "mfe", "mfcc_batch", "mfe_batch", "lmfe", "lmfe_batch", "power_to_db", "stft", "stack_frames", "power_spectrum",
           "power_spectrum_of_signal", "SpeechConfig", "SpeechSauceError"]


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class SpeechConfig:
    """Owns an ``ss_config`` handle: the counterpart of ``PySpeechSauce(SpeechConfig)``
    (py-speechsauce/src/lib.rs:7-10; speechsauce/src/config.rs:99-185)."""

    def __init__(self, params: SsParams):
        self.params = params
        self._h = C.c_void_p()
        self._owner = _lib.lib()  # the build that created the handle destroys it (tests run the front on the lab build too)
        _lib.check(self._owner.ss_config_create(C.byref(params), C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._owner.ss_config_destroy(h)
            except Exception:
                pass

    # ---- derived sizes -------------------------------------------------------------------
    def num_frames(self, n_samples: int) -> int:
        t = C.c_size_t()
        _lib.check(_lib.lib().ss_num_frames(C.byref(self.params), n_samples, C.byref(t)))
        return t.value

    def stft_rows(self, n_samples: int) -> tuple[int, int]:
        r, rr = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib().ss_stft_rows(C.byref(self.params), n_samples, C.byref(r), C.byref(rr)))
        return r.value, rr.value

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def device_status(self) -> None:
        """Raises SpeechSauceError (SS_ERR_DEVICE) if a kernel of an asynchronous launch on this config has reported a
        device-side protocol error since the last call.  Meaningful after the stream has been synchronised; the numpy
        (host-pointer) calls check it themselves."""
        _lib.check(_lib.lib().ss_config_device_status(self._h))


def _speech_config(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                   low_frequency, dc_elimination, high_frequency=None, **switches) -> SpeechConfig:
    """Same positional order as ``_internal._speech_config`` (py-speechsauce/src/lib.rs:226-254)."""
    return SpeechConfig(make_params(
        sample_rate=sampling_frequency, fft_points=fft_length, frame_length=frame_length,
        frame_stride=frame_stride, num_cepstral=num_cepstral, num_filters=num_filters,
        low_frequency=low_frequency, high_frequency=high_frequency, dc_elimination=dc_elimination, **switches))


@lru_cache(maxsize=32)
def _get_speech_config(sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13, num_filters=40,
                       fft_length=512, low_frequency=0, high_frequency: Optional[float] = None,
                       dc_elimination=True, switches: tuple = (), device: int = -1) -> SpeechConfig:
    """Memoised config factory (py-speechsauce/speechsauce/__init__.py:8-34).

    ``device`` is part of the key: a config owns tables in the memory of the HIP device that was current when it was
    created, so each device gets its own (-1 = whichever device is current, the host-array path)."""
    if device >= 0:
        import torch

        with torch.cuda.device(device):
            return _speech_config(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                                  low_frequency, dc_elimination, high_frequency, **dict(switches))
    return _speech_config(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                          low_frequency, dc_elimination, high_frequency, **dict(switches))


_lib._on_switch.append(_get_speech_config.cache_clear)  # a config belongs to the library that created it


def _device_key(signal) -> int:
    """Index of the ROCm device a tensor lives on; -1 for host arrays (the library's current device)."""
    if _is_torch(signal) and signal.is_cuda:
        return signal.device.index if signal.device.index is not None else -1
    if not _is_torch(signal):
        try:  # host arrays run on the current device: key the config by it, so that a later ss_set_device gets its own
            import torch

            if torch.cuda.is_initialized():
                return torch.cuda.current_device()
        except Exception:
            pass
    return -1


def _require_f32(signal, ndims: tuple[int, ...], what: str):
    """The binding takes PyReadonlyArray<f32> only (py-speechsauce/src/lib.rs:170,182): no silent casts."""
    if _is_torch(signal):
        import torch

        if signal.dtype != torch.float32:
            raise TypeError(f"{what}: signal must be float32, got {signal.dtype}")
        if signal.dim() not in ndims:
            raise ValueError(f"{what}: Input signal must be {' or '.join(str(d) + 'd' for d in ndims)}")
        if not signal.is_cuda:
            signal = signal.detach().numpy()
        return signal
    arr = np.asarray(signal)
    if arr.dtype != np.float32:
        raise TypeError(f"{what}: signal must be float32, got {arr.dtype}")
    if arr.ndim not in ndims:
        raise ValueError(f"{what}: Input signal must be {' or '.join(str(d) + 'd' for d in ndims)}")
    return arr


def _stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- internal entry points (the `_internal` pyfns, py-speechsauce/src/lib.rs:167-204) -------------

def _internal_mfcc_batch(signal, config: SpeechConfig):
    """signal [B, L] -> [B, T, num_cepstral]"""
    lib = _lib.lib()
    B, L = signal.shape
    T = config.num_frames(L)
    Cc = config.params.num_cepstral
    if _is_torch(signal):
        import torch

        x = signal if signal.stride(1) == 1 else signal.contiguous()
        out = torch.empty((B, T, Cc), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_mfcc_batch_device(config.handle, x.data_ptr(), B, L, x.stride(0) if B > 1 else L,
                                                out.data_ptr(), _stream_ptr()))
        return out
    x = np.ascontiguousarray(signal)
    out = np.empty((B, T, Cc), dtype=np.float32)
    _lib.check(lib.ss_mfcc_batch(config.handle, x.ctypes.data, B, L, L, out.ctypes.data))
    return out


def _internal_mfe_batch(signal, config: SpeechConfig):
    lib = _lib.lib()
    B, L = signal.shape
    T = config.num_frames(L)
    M = config.params.num_filters
    if _is_torch(signal):
        import torch

        x = signal if signal.stride(1) == 1 else signal.contiguous()
        feat = torch.empty((B, T, M), dtype=torch.float32, device=x.device)
        en = torch.empty((B, T), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_mfe_batch_device(config.handle, x.data_ptr(), B, L, x.stride(0) if B > 1 else L,
                                               feat.data_ptr(), en.data_ptr(), _stream_ptr()))
        return feat, en
    x = np.ascontiguousarray(signal)
    feat = np.empty((B, T, M), dtype=np.float32)
    en = np.empty((B, T), dtype=np.float32)
    _lib.check(lib.ss_mfe_batch(config.handle, x.ctypes.data, B, L, L, feat.ctypes.data, en.ctypes.data))
    return feat, en


def _internal_mel_spectrogram(signal, config: SpeechConfig):
    """1-D -> [n_mels, rows]; 2-D [C, L] -> [C, n_mels, rows] (py-speechsauce/src/lib.rs:179-204)."""
    lib = _lib.lib()
    one_d = signal.ndim == 1
    sig2 = signal[None, :] if one_d else signal
    ch, L = sig2.shape
    R, _ = config.stft_rows(L)
    M = config.params.num_filters
    if _is_torch(sig2):
        import torch

        x = sig2 if sig2.stride(1) == 1 else sig2.contiguous()
        out = torch.empty((ch, M, R), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_mel_spectrogram_device(config.handle, x.data_ptr(), ch, L, x.stride(0) if ch > 1 else L,
                                                     out.data_ptr(), _stream_ptr()))
    else:
        x = np.ascontiguousarray(sig2)
        out = np.empty((ch, M, R), dtype=np.float32)
        _lib.check(lib.ss_mel_spectrogram(config.handle, x.ctypes.data, ch, L, out.ctypes.data))
    return out[0] if one_d else out


def _internal_mel_spectrogram_batches(signals, config: SpeechConfig):
    """Several [C_i, L] blocks (same L) -> list of [C_i, n_mels, rows]."""
    if not all(_is_torch(x) and x.is_cuda for x in signals):
        return [_internal_mel_spectrogram(x, config) for x in signals]
    import torch

    lib = _lib.lib()
    dev = signals[0].device
    L = signals[0].shape[1]
    if any(x.device != dev or x.shape[1] != L for x in signals):
        raise ValueError("mel_spectrogram: the blocks of one call must live on one device and hold clips of one length")
    R, _ = config.stft_rows(L)
    M = config.params.num_filters
    xs = [x if (x.stride(1) == 1 and (x.shape[0] <= 1 or x.stride(0) == L)) else x.contiguous() for x in signals]
    outs = [torch.empty((x.shape[0], M, R), dtype=torch.float32, device=dev) for x in xs]
    n = len(xs)
    px = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    po = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    nb = (C.c_size_t * n)(*[x.shape[0] for x in xs])
    with torch.cuda.device(dev):
        _lib.check(lib.ss_mel_spectrogram_batches_device(config.handle, n, px, nb, L, L, po, _stream_ptr()))
    return outs


# ---- public API (py-speechsauce/speechsauce/__init__.py:37-132) ------------------------------------

def _cfg(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length, low_frequency,
         high_frequency, dc_elimination, switches, signal=None) -> SpeechConfig:
    return _get_speech_config(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                              low_frequency, high_frequency, dc_elimination, tuple(sorted(switches.items())),
                              _device_key(signal) if signal is not None else -1)


def mfcc(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13, num_filters=40,
         fft_length=512, low_frequency=0, high_frequency=None, dc_elimination=True, **switches):
    """MFCC features of a 1-D float32 signal -> (num_frames, num_cepstral).

    Mirrors ``speechsauce.mfcc`` (py-speechsauce/speechsauce/__init__.py:37-83 -> feature.rs:99-148).
    ``switches`` are the SURVEY section-0 options (framing, spectrum_exponent, dct_norm, dct2_gain,
    mfcc_window, preemph_coef, preemph_shift); none given == reference mode.
    """
    sig = _require_f32(signal, (1,), "mfcc")
    config = _cfg(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                  low_frequency, high_frequency, dc_elimination, switches, sig)
    return _internal_mfcc_batch(sig[None, :], config)[0]


def _internal_mfcc_batches(signals, config: SpeechConfig):
    """Several [B_i, L] batches (same L) -> list of [B_i, T, num_cepstral]: ONE ss_mfcc_batches_device call for device tensors
    (one launch for up to 8 batches where the kernel takes a batch table), one host call per batch for numpy arrays."""
    if not all(_is_torch(x) and x.is_cuda for x in signals):
        return [_internal_mfcc_batch(x, config) for x in signals]
    import torch

    lib = _lib.lib()
    dev = signals[0].device
    L = signals[0].shape[1]
    if any(x.device != dev or x.shape[1] != L for x in signals):
        raise ValueError("mfcc_batch: the batches of one call must live on one device and hold clips of one length")
    T = config.num_frames(L)
    Cc = config.params.num_cepstral
    xs = [x if (x.stride(1) == 1 and (x.shape[0] <= 1 or x.stride(0) == L)) else x.contiguous() for x in signals]
    outs = [torch.empty((x.shape[0], T, Cc), dtype=torch.float32, device=dev) for x in xs]
    n = len(xs)
    px = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    po = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    nb = (C.c_size_t * n)(*[x.shape[0] for x in xs])
    with torch.cuda.device(dev):
        _lib.check(lib.ss_mfcc_batches_device(config.handle, n, px, nb, L, L, po, _stream_ptr()))
    return outs


def mfcc_batch(signals, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13, num_filters=40,
               fft_length=512, low_frequency=0, high_frequency=None, dc_elimination=True, **switches):
    """Batch form: [B, L] float32 -> [B, num_frames, num_cepstral] in one launch.  A list / tuple of such batches (same clip
    length; e.g. the blocks a data loader hands over) -> the list of their feature blocks from ONE call: device tensors share one
    kernel launch where the configuration's kernel takes a batch table (ss_mfcc_batches_device)."""
    if isinstance(signals, (list, tuple)):
        sigs = [_require_f32(x, (2,), "mfcc_batch") for x in signals]
        if not sigs:
            return []
        config = _cfg(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                      low_frequency, high_frequency, dc_elimination, switches, sigs[0])
        return _internal_mfcc_batches(sigs, config)
    sig = _require_f32(signals, (2,), "mfcc_batch")
    config = _cfg(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                  low_frequency, high_frequency, dc_elimination, switches, sig)
    return _internal_mfcc_batch(sig, config)


def mfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
        low_frequency=0, high_frequency=None, **switches):
    """Mel filterbank energies and frame energies (feature.rs:200-233): ((T, num_filters), (T,))."""
    sig = _require_f32(signal, (1,), "mfe")
    config = _cfg(sampling_frequency, frame_length, frame_stride, min(13, num_filters), num_filters, fft_length,
                  low_frequency, high_frequency, True, switches, sig)
    feat, en = _internal_mfe_batch(sig[None, :], config)
    return feat[0], en[0]


def mfe_batch(signals, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
              low_frequency=0, high_frequency=None, **switches):
    sig = _require_f32(signals, (2,), "mfe_batch")
    config = _cfg(sampling_frequency, frame_length, frame_stride, min(13, num_filters), num_filters, fft_length,
                  low_frequency, high_frequency, True, switches, sig)
    return _internal_mfe_batch(sig, config)


def _internal_lmfe_batch(signal, config: SpeechConfig):
    lib = _lib.lib()
    B, L = signal.shape
    T = config.num_frames(L)
    M = config.params.num_filters
    if _is_torch(signal):
        import torch

        x = signal if signal.stride(1) == 1 else signal.contiguous()
        feat = torch.empty((B, T, M), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_lmfe_batch_device(config.handle, x.data_ptr(), B, L, x.stride(0) if B > 1 else L,
                                                feat.data_ptr(), None, _stream_ptr()))
        return feat
    x = np.ascontiguousarray(signal)
    feat = np.empty((B, T, M), dtype=np.float32)
    _lib.check(lib.ss_lmfe_batch(config.handle, x.ctypes.data, B, L, L, feat.ctypes.data))
    return feat


def lmfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
         low_frequency=0, high_frequency=None, **switches):
    """Log mel-filterbank energies (feature.rs:242-245; README.md:14): ln of mfe's features, (T, num_filters)."""
    sig = _require_f32(signal, (1,), "lmfe")
    config = _cfg(sampling_frequency, frame_length, frame_stride, min(13, num_filters), num_filters, fft_length,
                  low_frequency, high_frequency, True, switches, sig)
    return _internal_lmfe_batch(sig[None, :], config)[0]


def lmfe_batch(signals, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
               low_frequency=0, high_frequency=None, **switches):
    sig = _require_f32(signals, (2,), "lmfe_batch")
    config = _cfg(sampling_frequency, frame_length, frame_stride, min(13, num_filters), num_filters, fft_length,
                  low_frequency, high_frequency, True, switches, sig)
    return _internal_lmfe_batch(sig, config)


def power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0):
    """librosa.power_to_db: 10 log10(max(amin, S)) - 10 log10(max(amin, |ref|)), floored at max - top_db (None: no floor).
    The reference lists librosa's mel-spectrogram conventions as its remaining work (README.md:44-46).  numpy in -> numpy
    out; a ROCm tensor stays on the device (current stream)."""
    lib = _lib.lib()
    td = -1.0 if top_db is None else float(top_db)
    if top_db is not None and top_db < 0:
        raise ValueError("top_db must be non-negative")
    if _is_torch(S) and S.is_cuda:
        import torch

        if S.dtype != torch.float32:
            raise TypeError("power_to_db: expected float32")
        x = S.contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_power_to_db_device(x.data_ptr(), x.numel(), float(ref), float(amin), td, out.data_ptr(), _stream_ptr()))
        return out
    arr = np.ascontiguousarray(S.numpy() if _is_torch(S) else S)
    if arr.dtype != np.float32:
        raise TypeError("power_to_db: expected float32")
    out = np.empty_like(arr)
    _lib.check(lib.ss_power_to_db(arr.ctypes.data, arr.size, float(ref), float(amin), td, out.ctypes.data))
    return out


def mel_spectrogram(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13,
                    num_filters=40, fft_length=512, low_frequency=0, high_frequency=None, dc_elimination=True,
                    **switches):
    """Mel spectrogram of a 1-D or 2-D float32 signal -> (..., n_mels, time).

    Mirrors ``speechsauce.mel_spectrogram`` (py-speechsauce/speechsauce/__init__.py:85-132 ->
    feature.rs:151-174).  The STFT hop is ``frame_length * sampling_frequency`` samples and the
    window is ``fft_length`` samples (config.rs:154, functions.rs:96-101); the reference panics
    unless ``fft_length >= 2 * hop`` -- that raises SpeechSauceError here.
    """
    if isinstance(signal, (list, tuple)):
        # several [C_i, L] blocks (same L) -> the list of their [C_i, n_mels, time] spectrograms from ONE call (device tensors:
        # ss_mel_spectrogram_batches_device -- one launch where the configuration's kernel takes a batch table)
        sigs = [_require_f32(x, (2,), "mel_spectrogram") for x in signal]
        if not sigs:
            return []
        config = _cfg(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                      low_frequency, high_frequency, dc_elimination, switches, sigs[0])
        return _internal_mel_spectrogram_batches(sigs, config)
    sig = _require_f32(signal, (1, 2), "mel_spectrogram")
    config = _cfg(sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length,
                  low_frequency, high_frequency, dc_elimination, switches, sig)
    return _internal_mel_spectrogram(sig, config)


# ---- stage outputs the reference exposes as pub fns: processing::{stack_frames, power_spectrum}, functions::{stft1, stft2} ----

def stft(signal, sampling_frequency, frame_length=0.020, fft_length=512, **switches):
    """``speechsauce::functions::stft1`` (1-D signal, functions.rs:199-233) / ``stft2`` (2-D [C, L], functions.rs:86-123):
    complex64 spectrum rows ``(..., rows, fft_length // 2 + 1)`` of the Vorbis-windowed chunks, scaled by wnorm, from zero
    state per clip.  The hop is ``frame_length * sampling_frequency`` samples, the window ``fft_length`` samples
    (config.rs:154); ``fft_length >= 2 * hop`` as in ``mel_spectrogram``.  numpy in -> numpy out; a ROCm tensor stays on
    the device (torch.complex64 view of the interleaved block)."""
    sig = _require_f32(signal, (1, 2), "stft")
    config = _cfg(sampling_frequency, frame_length, 0.01, 13, 40, fft_length, 0, None, True, switches, sig)
    lib = _lib.lib()
    one_d = sig.ndim == 1
    sig2 = sig[None, :] if one_d else sig
    ch, L = sig2.shape
    R, _ = config.stft_rows(L)
    F = config.params.fft_points // 2 + 1
    if _is_torch(sig2):
        import torch

        x = sig2 if sig2.stride(1) == 1 else sig2.contiguous()
        out = torch.empty((ch, R, F, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_stft_device(config.handle, x.data_ptr(), ch, L, x.stride(0) if ch > 1 else L, out.data_ptr(), _stream_ptr()))
        z = torch.view_as_complex(out)
    else:
        x = np.ascontiguousarray(sig2)
        out = np.empty((ch, R, F, 2), dtype=np.float32)
        _lib.check(lib.ss_stft(config.handle, x.ctypes.data, ch, L, out.ctypes.data))
        z = out.view(np.complex64)[..., 0]
    return z[0] if one_d else z


def stack_frames(signal, sampling_frequency, frame_length=0.020, frame_stride=0.020, filter=None, zero_padding=False, **switches):
    """``speechsauce::processing::stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding)``
    (processing.rs:65-129): (num_frames, frame_len) frames of a 1-D signal, any sampling rate and frame length (the function
    has no FFT dependency: 44.1 kHz x 25 ms = round(1102.5) = 1103-sample frames are fine).  ``filter``: as in the reference, a callable that
    gets frame_len and returns the window as a (1, frame_len) array (row 0 is used; a 1-D array of frame_len works too), or an
    array; ``zero_padding=True`` is the reference's flag (ceil instead of floor frames, the tail reading appended zeros).
    ``switches`` (``framing="literal" | "center"``, ``mfcc_window=``, ``pad_mode=``) go through a SpeechConfig instead and then
    need frame_len <= 8192."""
    sig = _require_f32(signal, (1,), "stack_frames")
    if isinstance(filter, (bool, np.bool_)):
        # round 3's signature had zero_padding in this position: an old positional call must fail clearly, not index a bool
        raise TypeError("stack_frames: the fifth argument is `filter` (a callable or an array), as in the reference "
                        "(processing.rs:65-76); pass zero_padding=... by keyword or as the sixth argument")
    lib = _lib.lib()
    L = sig.shape[0]
    if switches:
        if filter is not None:
            raise ValueError("stack_frames: pass either filter= or the mfcc_window switch")
        if zero_padding:
            switches = dict(switches, framing="padded")
        flen = int(np.floor(np.float32(np.float32(sampling_frequency) * np.float32(frame_length)) + np.float32(0.5)))  # f32::round
        n_fft = 512
        while n_fft < flen and n_fft < 8192:  # the config needs an FFT length that holds the frame; nothing else uses it here
            n_fft *= 2
        config = _cfg(sampling_frequency, frame_length, frame_stride, 13, 40, n_fft, 0, None, True, switches, sig)
        T = config.num_frames(L)
        fl, st = C.c_size_t(), C.c_size_t()
        _lib.check(lib.ss_frame_sizes(C.byref(config.params), C.byref(fl), C.byref(st)))
        if _is_torch(sig):
            import torch

            x = sig.contiguous()
            out = torch.empty((T, fl.value), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(lib.ss_stack_frames_device(config.handle, x.data_ptr(), 1, L, L, out.data_ptr(), _stream_ptr()))
            return out
        x = np.ascontiguousarray(sig)
        out = np.empty((T, fl.value), dtype=np.float32)
        _lib.check(lib.ss_stack_frames(config.handle, x.ctypes.data, L, out.ctypes.data))
        return out
    T, fl = C.c_size_t(), C.c_size_t()
    _lib.check(lib.ss_stack_frames_shape(L, int(sampling_frequency), float(frame_length), float(frame_stride), int(bool(zero_padding)),
                                         C.byref(T), C.byref(fl)))
    win = None
    if filter is not None:
        w = filter(fl.value) if callable(filter) else filter
        w = w.detach().cpu().numpy() if _is_torch(w) else np.asarray(w)
        w = np.ascontiguousarray(w.reshape(-1)[: fl.value] if w.ndim == 1 or w.shape[0] != 1 else w[0], dtype=np.float32)
        if w.shape[0] != fl.value:
            raise ValueError(f"stack_frames: filter gave {w.shape[0]} values for frames of {fl.value} samples")
        win = w
    if _is_torch(sig):
        import torch

        x = sig.contiguous()
        out = torch.empty((T.value, fl.value), dtype=torch.float32, device=x.device)
        wd = torch.from_numpy(win).to(x.device) if win is not None else None
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_stack_frames_signal_device(x.data_ptr(), L, int(sampling_frequency), float(frame_length), float(frame_stride),
                                                         wd.data_ptr() if wd is not None else None, int(bool(zero_padding)), out.data_ptr(),
                                                         _stream_ptr()))
        if wd is not None:
            wd.record_stream(torch.cuda.current_stream(x.device))
        return out
    x = np.ascontiguousarray(sig)
    out = np.empty((T.value, fl.value), dtype=np.float32)
    _lib.check(lib.ss_stack_frames_signal(x.ctypes.data, L, int(sampling_frequency), float(frame_length), float(frame_stride),
                                          win.ctypes.data if win is not None else None, int(bool(zero_padding)), out.ctypes.data))
    return out


def power_spectrum(frames, fft_points=512):
    """``speechsauce::processing::power_spectrum(frames, fft_points)`` (processing.rs:179-181): |rfft(row)| / fft_points of
    every row of a 2-D float32 frames matrix (rows shorter than fft_points are zero-padded, processing.rs:147-156) ->
    (num_frames, fft_points // 2 + 1).  The name is the reference's; the values are magnitudes, as written there."""
    if _is_torch(frames):
        import torch

        if frames.dtype != torch.float32:
            raise TypeError("power_spectrum: frames must be float32")
        if frames.dim() != 2:
            raise ValueError("power_spectrum: frames must be 2d")
        fr = frames if frames.is_cuda else frames.detach().numpy()
    else:
        fr = np.asarray(frames)
        if fr.dtype != np.float32:
            raise TypeError(f"power_spectrum: frames must be float32, got {fr.dtype}")
        if fr.ndim != 2:
            raise ValueError("power_spectrum: frames must be 2d")
    # only fft_points of the config matters here; the other fields are chosen so that the config validates with it
    # (a frame of half the FFT length, a bank that fits the spectrum)
    n = int(fft_points)
    nf = max(1, min(40, n // 8))
    config = _cfg(16000, n / 32000.0, n / 64000.0, min(13, nf), nf, n, 0, None, True, {}, fr)
    lib = _lib.lib()
    rows, cols = fr.shape
    F = int(fft_points) // 2 + 1
    if _is_torch(fr):
        import torch

        x = fr if fr.stride(1) == 1 else fr.contiguous()
        out = torch.empty((rows, F), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_power_spectrum_frames_device(config.handle, x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols,
                                                           out.data_ptr(), _stream_ptr()))
        return out
    x = np.ascontiguousarray(fr)
    out = np.empty((rows, F), dtype=np.float32)
    _lib.check(lib.ss_power_spectrum_frames(config.handle, x.ctypes.data, rows, cols, out.ctypes.data))
    return out


def power_spectrum_of_signal(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, fft_length=512, **switches):
    """stack_frames + power_spectrum fused, as mfe uses them (feature.rs:203-214): 1-D -> (T, F), 2-D [B, L] -> (B, T, F)."""
    sig = _require_f32(signal, (1, 2), "power_spectrum_of_signal")
    config = _cfg(sampling_frequency, frame_length, frame_stride, 13, 40, fft_length, 0, None, True, switches, sig)
    lib = _lib.lib()
    one_d = sig.ndim == 1
    sig2 = sig[None, :] if one_d else sig
    B, L = sig2.shape
    T = config.num_frames(L)
    F = config.params.fft_points // 2 + 1
    if _is_torch(sig2):
        import torch

        x = sig2 if sig2.stride(1) == 1 else sig2.contiguous()
        out = torch.empty((B, T, F), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_power_spectrum_batch_device(config.handle, x.data_ptr(), B, L, x.stride(0) if B > 1 else L, out.data_ptr(), _stream_ptr()))
    else:
        x = np.ascontiguousarray(sig2)
        out = np.empty((B, T, F), dtype=np.float32)
        _lib.check(lib.ss_power_spectrum_batch(config.handle, x.ctypes.data, B, L, L, out.ctypes.data))
    return out[0] if one_d else out


def preemphasis(signal, shift=1, cof=0.98):
    """y[n] = x[n] - cof * x[(n - shift) mod N]  (processing.rs:31-53; py lib.rs:207-215)."""
    sig = _require_f32(signal, (1,), "preemphasis")
    lib = _lib.lib()
    n = sig.shape[0]
    if _is_torch(sig):
        import torch

        x = sig.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.ss_preemphasis_device(x.data_ptr(), n, int(shift), float(cof), y.data_ptr(), _stream_ptr()))
        return y
    x = np.ascontiguousarray(sig)
    y = np.empty_like(x)
    _lib.check(lib.ss_preemphasis(x.ctypes.data, n, int(shift), float(cof), y.ctypes.data))
    return y


# ---- post-processing on the feature matrix (processing.rs:222-371, feature.rs:253-269; `cmvn` is exported by the
# reference's Python package, py lib.rs:217-224).  numpy [rows, cols] in -> numpy out; a ROCm tensor of shape
# [rows, cols] or [batch, rows, cols] stays on the device (current stream). ----

def _feature_matrix(vec, what):
    if _is_torch(vec):
        import torch

        if vec.dtype != torch.float32:
            raise TypeError(f"{what}: expected float32")
        if vec.dim() not in (2, 3):
            raise ValueError(f"{what}: expected a [rows, cols] or [batch, rows, cols] tensor")
        if not vec.is_cuda:
            return np.ascontiguousarray(vec.numpy()), False
        return vec.contiguous(), True
    arr = np.asarray(vec)
    if arr.dtype != np.float32:
        raise TypeError(f"{what}: expected float32 (the reference binding takes PyReadonlyArray2<f32>)")
    if arr.ndim != 2:
        raise ValueError(f"{what}: expected a 2-D [rows, cols] array")
    return np.ascontiguousarray(arr), False


def _post(vec, what, host_fn, dev_fn, out_tail=()):
    x, on_device = _feature_matrix(vec, what)
    if on_device:
        import torch

        batch = x.shape[0] if x.dim() == 3 else 1
        rows, cols = x.shape[-2], x.shape[-1]
        out = torch.empty(tuple(x.shape) + tuple(out_tail), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(dev_fn(x.data_ptr(), batch, rows, cols, out.data_ptr(), _stream_ptr()))
        return out
    if x.ndim == 3:  # CPU tensor with a batch axis: one matrix at a time through the host entry point
        return np.stack([_post(m, what, host_fn, dev_fn, out_tail) for m in x])
    out = np.empty(x.shape + tuple(out_tail), dtype=np.float32)
    _lib.check(host_fn(x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data))
    return out


def cmvn(vec, variance_normalization=False):
    """Global cepstral mean (and variance) normalisation, one observation per row (processing.rs:265-300)."""
    lib, var = _lib.lib(), int(bool(variance_normalization))
    return _post(vec, "cmvn", lambda p, r, c, o: lib.ss_cmvn(p, r, c, var, o),
                 lambda p, b, r, c, o, s: lib.ss_cmvn_batch_device(p, b, r, c, var, o, s))


def cmvnw(vec, win_size=301, variance_normalization=False):
    """Sliding-window mean (and variance) normalisation over win_size rows (processing.rs:315-371)."""
    lib, var, w = _lib.lib(), int(bool(variance_normalization)), int(win_size)
    return _post(vec, "cmvnw", lambda p, r, c, o: lib.ss_cmvnw(p, r, c, w, var, o),
                 lambda p, b, r, c, o, s: lib.ss_cmvnw_batch_device(p, b, r, c, w, var, o, s))


def derivative_extraction(feat, delta_windows):
    """Derivative features along the feature axis (processing.rs:222-254)."""
    lib, dw = _lib.lib(), int(delta_windows)
    return _post(feat, "derivative_extraction", lambda p, r, c, o: lib.ss_derivative_extraction(p, r, c, dw, o),
                 lambda p, b, r, c, o, s: lib.ss_derivative_extraction_device(p, b * r, c, dw, o, s))


def extract_derivative_feature(feature):
    """[..., rows, cols] -> [..., rows, cols, 3]: static, first and second derivative features (feature.rs:253-269)."""
    lib = _lib.lib()
    return _post(feature, "extract_derivative_feature", lambda p, r, c, o: lib.ss_extract_derivative_feature(p, r, c, o),
                 lambda p, b, r, c, o, s: lib.ss_extract_derivative_feature_device(p, b * r, c, o, s), out_tail=(3,))

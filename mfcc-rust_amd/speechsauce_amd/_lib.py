"""ctypes binding of libspeechsauce_amd.so (the C ABI of include/speechsauce_amd.h).

This plays the role of the reference's PyO3 module ``speechsauce._internal``
(py-speechsauce/src/lib.rs:141-256).  There is no CPU fallback: if the shared library is missing
or no HIP device is usable, every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
# SS_LIB_PATH overrides the in-tree build (A/B runs of two builds in one process tree)
LIB_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(_PKG_ROOT, "lib", "libspeechsauce_amd.so")
# the lab build (same sources, -DSS_LAB=1): the only library that exports the ss_debug_* test aids
LAB_LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libspeechsauce_amd_lab.so")
ABI_VERSION = 7

SS_OK, SS_ERR_SHORT_SIGNAL, SS_ERR_BAD_CONFIG, SS_ERR_ARG, SS_ERR_HIP, SS_ERR_UNSUPPORTED, SS_ERR_DEVICE = range(7)
FRAMING = {"contract": 0, "literal": 1, "center": 2, "padded": 3}
MEL_SCALE = {"reference": 0, "slaney": 1, "htk": 2}
MEL_NORM = {"none": 0, "slaney": 1}
PAD_MODE = {"reflect": 0, "constant": 1}
DCT_NORM = {"reference": 0, "ortho": 1}
WINDOW = {"rect": 0, "hann": 1, "vorbis": 2}
DCT2_GAIN = 2.0


class SsParams(C.Structure):
    """ss_params (include/speechsauce_amd.h); field order is ABI."""

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


class SpeechSauceError(RuntimeError):
    """Raised where the reference would panic (or where HIP fails)."""

    def __init__(self, status: int, detail: str):
        super().__init__(f"speechsauce_amd status {status}: {detail}")
        self.status = status
        self.detail = detail


_lib = None

# name -> (restype, argtypes); every symbol include/speechsauce_amd.h declares (LAB_PROTOTYPES: include/speechsauce_amd_debug.h)
_P = C.POINTER
_cfg = C.c_void_p
_fp = C.c_void_p  # float* passed as an address (numpy .ctypes.data or a device pointer)
PROTOTYPES = {
    "ss_params_default": (C.c_int, [_P(SsParams), C.c_uint32]),
    "ss_config_create": (C.c_int, [_P(SsParams), _P(_cfg)]),
    "ss_config_destroy": (None, [_cfg]),
    "ss_config_params": (C.c_int, [_cfg, _P(SsParams)]),
    "ss_config_device_status": (C.c_int, [_cfg]),
    "ss_params_validate": (C.c_int, [_P(SsParams)]),
    "ss_frame_sizes": (C.c_int, [_P(SsParams), _P(C.c_size_t), _P(C.c_size_t)]),
    "ss_num_frames": (C.c_int, [_P(SsParams), C.c_size_t, _P(C.c_size_t)]),
    "ss_stft_sizes": (C.c_int, [_P(SsParams), _P(C.c_size_t), _P(C.c_size_t), _P(C.c_float)]),
    "ss_stft_rows": (C.c_int, [_P(SsParams), C.c_size_t, _P(C.c_size_t), _P(C.c_size_t)]),
    "ss_filterbank": (C.c_int, [_P(SsParams), _fp, C.c_void_p]),
    "ss_vorbis_window": (C.c_int, [C.c_size_t, _fp]),
    "ss_mfcc": (C.c_int, [_cfg, _fp, C.c_size_t, _fp]),
    "ss_mfe": (C.c_int, [_cfg, _fp, C.c_size_t, _fp, _fp]),
    "ss_mel_spectrogram": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, _fp]),
    "ss_preemphasis": (C.c_int, [_fp, C.c_size_t, C.c_long, C.c_float, _fp]),
    "ss_stft": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, _fp]),
    "ss_stack_frames": (C.c_int, [_cfg, _fp, C.c_size_t, _fp]),
    "ss_power_spectrum_frames": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, _fp]),
    "ss_power_spectrum": (C.c_int, [_cfg, _fp, C.c_size_t, _fp]),
    "ss_power_spectrum_batch": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp]),
    "ss_mfcc_batch": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp]),
    "ss_mfe_batch": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp]),
    "ss_mfcc_batch_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_mfe_batch_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp, C.c_void_p]),
    "ss_mfcc_batches_device": (C.c_int, [_cfg, C.c_size_t, _P(C.c_void_p), _P(C.c_size_t), C.c_size_t, C.c_size_t, _P(C.c_void_p), C.c_void_p]),
    "ss_mel_spectrogram_batches_device": (C.c_int, [_cfg, C.c_size_t, _P(C.c_void_p), _P(C.c_size_t), C.c_size_t, C.c_size_t, _P(C.c_void_p), C.c_void_p]),
    "ss_mfcc_timed_region": (C.c_int, [_cfg, _P(C.c_void_p), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _P(C.c_void_p), C.c_size_t, C.c_void_p,
                                       C.c_int, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "ss_mel_spectrogram_timed_region": (C.c_int, [_cfg, _P(C.c_void_p), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _P(C.c_void_p), C.c_size_t,
                                                  C.c_void_p, C.c_int, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "ss_mel_spectrogram_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_preemphasis_device": (C.c_int, [_fp, C.c_size_t, C.c_long, C.c_float, _fp, C.c_void_p]),
    "ss_lmfe": (C.c_int, [_cfg, _fp, C.c_size_t, _fp]),
    "ss_lmfe_batch": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp]),
    "ss_lmfe_batch_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp, C.c_void_p]),
    "ss_ln_device": (C.c_int, [_fp, C.c_size_t, C.c_void_p]),
    "ss_power_to_db": (C.c_int, [_fp, C.c_size_t, C.c_float, C.c_float, C.c_float, _fp]),
    "ss_power_to_db_device": (C.c_int, [_fp, C.c_size_t, C.c_float, C.c_float, C.c_float, _fp, C.c_void_p]),
    "ss_shard_bounds": (C.c_int, [C.c_size_t, C.c_int, C.c_int, _P(C.c_size_t), _P(C.c_size_t)]),
    "ss_all_gather_features": (C.c_int, [C.c_void_p, _fp, C.c_size_t, _fp, C.c_void_p]),
    "ss_gather_features": (C.c_int, [C.c_void_p, _fp, C.c_size_t, _fp, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ss_rccl_library": (C.c_int, [C.c_char_p]),
    "ss_power_spectrum_batch_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_power_spectrum_frames_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_stack_frames_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_stft_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_cmvn": (C.c_int, [_fp, C.c_size_t, C.c_size_t, C.c_int, _fp]),
    "ss_cmvn_batch_device": (C.c_int, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _fp, C.c_void_p]),
    "ss_cmvnw": (C.c_int, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _fp]),
    "ss_cmvnw_batch_device": (C.c_int, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _fp, C.c_void_p]),
    "ss_derivative_extraction": (C.c_int, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp]),
    "ss_derivative_extraction_device": (C.c_int, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_extract_derivative_feature": (C.c_int, [_fp, C.c_size_t, C.c_size_t, _fp]),
    "ss_extract_derivative_feature_device": (C.c_int, [_fp, C.c_size_t, C.c_size_t, _fp, C.c_void_p]),
    "ss_device_count": (C.c_int, [_P(C.c_int)]),
    "ss_set_device": (C.c_int, [C.c_int]),
    "ss_last_kernel_name": (C.c_char_p, []),
    "ss_time_mfcc_batch_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p, C.c_int, _P(C.c_float)]),
    "ss_time_mel_spectrogram_device": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p, C.c_int, _P(C.c_float)]),
    "ss_mfcc_shader_clock": (C.c_int, [_cfg, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, C.c_void_p, C.c_int, _P(C.c_float)]),
    "ss_shader_clock_probe": (C.c_int, [C.c_void_p, C.c_uint, _P(C.c_float)]),
    "ss_shader_clock_probe_async": (C.c_int, [C.c_void_p, C.c_uint, C.c_void_p]),
    "ss_stack_frames_shape": (C.c_int, [C.c_size_t, C.c_uint32, C.c_float, C.c_float, C.c_int, _P(C.c_size_t), _P(C.c_size_t)]),
    "ss_stack_frames_signal": (C.c_int, [_fp, C.c_size_t, C.c_uint32, C.c_float, C.c_float, _fp, C.c_int, _fp]),
    "ss_stack_frames_signal_device": (C.c_int, [_fp, C.c_size_t, C.c_uint32, C.c_float, C.c_float, _fp, C.c_int, _fp, C.c_void_p]),
    "ss_status_string": (C.c_char_p, [C.c_int]),
    "ss_last_error_string": (C.c_char_p, []),
    "ss_abi_version": (C.c_int, []),
}
# the process-wide test aids: exported by the lab library only (include/speechsauce_amd_debug.h)
LAB_PROTOTYPES = {
    "ss_debug_poison_lds": (C.c_int, [C.c_void_p]),
    "ss_debug_stamp_buffer": (C.c_int, [C.c_void_p]),
    "ss_debug_force_generic": (C.c_int, [C.c_int]),
    "ss_debug_mel_tile": (C.c_int, [C.c_int]),
    "ss_debug_tile_fault": (C.c_int, [C.c_int]),
}


def load(path: str, lab: bool = False):
    """dlopen one build of the library and give every entry point its prototype.  The ABI version is checked for EVERY build
    (an older library would be driven with the current SsParams layout and prototypes: silent struct-layout errors); a symbol
    the build lacks is an error unless SS_LIB_LENIENT=1 says the caller knows (A/B runs against an older library)."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python __graft_entry__.py build` "
            "(or `make -C mfcc-rust_amd/csrc [lab]`). speechsauce_amd has no CPU fallback."
        )
    try:
        import torch  # noqa: F401  (device memory / streams plumbing; also pins the HIP runtime)
    except Exception:  # pragma: no cover - torch is optional for the numpy host path
        pass
    handle = C.CDLL(path)
    lenient = os.environ.get("SS_LIB_LENIENT") == "1"
    protos = dict(PROTOTYPES, **LAB_PROTOTYPES) if lab else PROTOTYPES
    for name, (res, args) in protos.items():
        fn = getattr(handle, name, None)
        if fn is None:
            if lenient:
                continue
            raise ImportError(f"{path} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if not lab:  # a lab build handed in through SS_LIB_PATH (tools/): its test aids get their prototypes too
        for name, (res, args) in LAB_PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                continue
            fn.restype = res
            fn.argtypes = args
    got = handle.ss_abi_version()
    if got != ABI_VERSION:
        msg = f"{path}: ABI version {got}, this front speaks {ABI_VERSION}"
        if not lenient:
            raise ImportError(msg + " (SS_LIB_LENIENT=1 to drive it anyway)")
        import warnings

        warnings.warn(msg)
    return handle


def lib():
    """The library the front runs on, loaded once: the in-tree product build (or SS_LIB_PATH).  torch is imported first
    when present so that both share one HIP runtime (same SONAME libamdhip64.so.7; the loader reuses the copy torch mapped)."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


_lab = None
_on_switch = []  # callbacks of the front (its memoised configs belong to one library)


def lab():
    """The lab build (libspeechsauce_amd_lab.so), loaded once and only on request: the ss_debug_* test aids live there."""
    global _lab
    if _lab is None:
        _lab = load(LAB_LIB_PATH, lab=True)
    return _lab


class use_library:
    """Context manager for tests: run the Python front on another build (the lab library) inside the block.  Configs are
    per-library objects, so the front's memoised configs are dropped on the way in and out."""

    def __init__(self, handle):
        self.handle = handle

    def __enter__(self):
        global _lib
        self.prev = lib()
        for cb in _on_switch:
            cb()
        _lib = self.handle
        return self.handle

    def __exit__(self, *exc):
        global _lib
        for cb in _on_switch:
            cb()
        _lib = self.prev
        return False


def check(status: int) -> None:
    if status != SS_OK:
        detail = lib().ss_last_error_string().decode() or lib().ss_status_string(status).decode()
        raise SpeechSauceError(status, detail)


def make_params(sample_rate=16000, fft_points=512, frame_length=0.02, frame_stride=0.01, num_cepstral=13,
                num_filters=40, low_frequency=0.0, high_frequency=None, dc_elimination=True,
                framing="contract", spectrum_exponent=1, dct_norm="reference", dct2_gain=DCT2_GAIN,
                mfcc_window="rect", preemph_coef=0.0, preemph_shift=1, mel_scale="reference", mel_norm="none",
                pad_mode="reflect") -> SsParams:
    p = SsParams()
    check(lib().ss_params_default(C.byref(p), int(sample_rate)))
    p.fft_points = int(fft_points)
    p.frame_length = float(frame_length)
    p.frame_stride = float(frame_stride)
    p.num_cepstral = int(num_cepstral)
    p.num_filters = int(num_filters)
    p.low_frequency = float(low_frequency)
    # py-speechsauce/src/lib.rs:249: high_frequency.unwrap_or(sampling_frequency as f32 / 2.0)
    p.high_frequency = float(sample_rate) / 2.0 if high_frequency is None else float(high_frequency)
    p.dc_elimination = int(bool(dc_elimination))
    p.framing = FRAMING[framing] if isinstance(framing, str) else int(framing)
    p.spectrum_exponent = int(spectrum_exponent)
    p.dct_norm = DCT_NORM[dct_norm] if isinstance(dct_norm, str) else int(dct_norm)
    p.dct2_gain = float(dct2_gain)
    p.mfcc_window = WINDOW[mfcc_window] if isinstance(mfcc_window, str) else int(mfcc_window)
    p.preemph_coef = float(preemph_coef)
    p.preemph_shift = int(preemph_shift)
    p.mel_scale = MEL_SCALE[mel_scale] if isinstance(mel_scale, str) else int(mel_scale)
    p.mel_norm = MEL_NORM[mel_norm] if isinstance(mel_norm, str) else int(mel_norm)
    p.pad_mode = PAD_MODE[pad_mode] if isinstance(pad_mode, str) else int(pad_mode)
    return p

"""librosa-compatible variants: centred framing with reflect / zero padding, Slaney or HTK mel scale with
continuous triangles and area normalisation (librosa.filters.mel), on top of the existing periodic Hann window, power = 2
and ortho-DCT switches.

librosa is not installed in this image, so these are pinned the same way as the rest ("parity unpinned" against the real
library): the C oracle against an independent numpy restatement written the way librosa writes it (np.pad, rfftfreq,
subtract.outer ramps), plus closed-form known answers of the published algorithm.  GPU: the HIP path against the oracle.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

RTOL = 1e-4
# librosa.feature.melspectrogram-style front end at 16 kHz: n_fft = win_length = 512, hop 160, 40 mels, power 2
LIBROSA_LIKE = dict(sample_rate=16000, fft_points=512, frame_length=512 / 16000, frame_stride=0.01, num_cepstral=13,
                    num_filters=40, framing="center", pad_mode="reflect", mfcc_window="hann", spectrum_exponent=2,
                    mel_scale="slaney", mel_norm="slaney", dct_norm="ortho")


def _rel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def _signal(seed, n):
    return (np.random.default_rng(seed).standard_normal(n) * 0.1).astype(np.float32)


def _np_params(**kw):
    import oracle_np as on

    return on.Params(**kw)


# ------------------------------------------------------------------------------------------------------------ CPU pins

def test_slaney_scale_known_answers(oracle):
    """Slaney mel scale: linear 200/3 Hz per mel below 1 kHz, 27 mels per factor 6.4 above (mel(1000) = 15,
    mel(6400) = 42): with fmin = 0, fmax = 6400 and 41 filters the 43 mel points are the integers 0..42."""
    p = oracle.make_params(sample_rate=16000, fft_points=512, num_filters=41, num_cepstral=13, high_frequency=6400.0,
                           mel_scale="slaney")
    fb, _ = oracle.filterbank(p)
    freqs = np.arange(257) * 16000 / 512
    centre = lambda m: 200.0 / 3 * m if m <= 15 else 1000.0 * 6.4 ** ((m - 15) / 27.0)  # noqa: E731
    for m in (0, 7, 14, 15, 30, 40):
        lo, mid, hi = centre(m), centre(m + 1), centre(m + 2)
        want = np.maximum(0, np.minimum((freqs - lo) / (mid - lo), (hi - freqs) / (hi - mid)))
        assert np.abs(fb[m] - want).max() < 1e-6, m
    # area normalisation: each triangle integrates to 1 over Hz (bin spacing 31.25 Hz; trapezoid error for narrow filters)
    pn = oracle.make_params(sample_rate=16000, fft_points=4096, num_filters=41, high_frequency=6400.0, mel_scale="slaney",
                            mel_norm="slaney", frame_length=0.02)
    fbn, _ = oracle.filterbank(pn)
    area = fbn.sum(axis=1) * (16000 / 4096)
    assert np.abs(area - 1).max() < 2e-2
    # HTK scale: same construction on 2595 log10(1 + f/700)
    ph = oracle.make_params(sample_rate=16000, fft_points=512, num_filters=10, mel_scale="htk")
    fbh, _ = oracle.filterbank(ph)
    mel_pts = np.linspace(0, 2595 * np.log10(1 + 8000 / 700), 12)
    hz = 700 * (10 ** (mel_pts / 2595) - 1)
    want = np.maximum(0, np.minimum((freqs - hz[3]) / (hz[4] - hz[3]), (hz[5] - freqs) / (hz[5] - hz[4])))
    assert np.abs(fbh[3] - want).max() < 1e-6
    with pytest.raises(oracle.OracleError):  # area normalisation needs a continuous bank
        oracle.filterbank(oracle.make_params(mel_norm="slaney"))


def test_centre_framing_known_answers(oracle):
    """1 + n // hop frames; frame 0 of a ramp under reflect padding is the mirrored ramp."""
    p = oracle.make_params(**dict(LIBROSA_LIKE, mfcc_window="rect", spectrum_exponent=1))
    assert oracle.num_frames(p, 16000) == 101 and oracle.num_frames(p, 16159) == 101 and oracle.num_frames(p, 16160) == 102
    with pytest.raises(oracle.OracleError):
        oracle.num_frames(p, 256)  # reflect padding needs more than n_fft / 2 samples
    pc = oracle.make_params(**dict(LIBROSA_LIKE, pad_mode="constant"))
    assert oracle.num_frames(pc, 100) == 1
    x = np.arange(2000, dtype=np.float32) / 2000
    P = oracle.power_spectrum(p, x)
    frame0 = np.concatenate([x[256:0:-1], x[:256]])  # np.pad(x, 256, "reflect")[:512]
    assert _rel(P[0], np.abs(np.fft.rfft(frame0.astype(np.float64))) / 512) < 1e-12
    frame_last = np.pad(x.astype(np.float64), 256, mode="reflect")[12 * 160: 12 * 160 + 512]
    assert _rel(P[12], np.abs(np.fft.rfft(frame_last)) / 512) < 1e-12


def test_oracle_matches_numpy_restatement(oracle):
    import oracle_np as on

    x = _signal(61, 16000)
    for kw in (LIBROSA_LIKE, dict(LIBROSA_LIKE, pad_mode="constant", mel_scale="htk", mel_norm="none"),
               dict(LIBROSA_LIKE, mel_norm="none", preemph_coef=0.97),
               dict(sample_rate=16000, mel_scale="slaney", mel_norm="slaney"),       # reference framing, librosa bank
               dict(sample_rate=16000, framing="center")):                            # librosa framing, reference bank
        p, q = oracle.make_params(**kw), _np_params(**kw)
        fb, _ = oracle.filterbank(p)
        assert np.array_equal(fb, on.filterbank(q)[0]) or np.abs(fb - on.filterbank(q)[0]).max() < 1e-7
        assert oracle.num_frames(p, 16000) == on.num_frames(q, 16000)
        assert _rel(oracle.mfcc(p, x), on.mfcc(q, x)) < 1e-9, kw
        wf, we = on.mfe(q, x)
        gf, ge = oracle.mfe(p, x)
        assert _rel(gf, wf) < 1e-9 and _rel(ge, we) < 1e-9


def test_host_tables_follow_the_switches(sslib, oracle):
    """ss_filterbank / ss_num_frames (host side of the ABI, no device needed) against the oracle."""
    import ctypes as C

    from speechsauce_amd import make_params

    kw = dict(LIBROSA_LIKE)
    p, q = make_params(**kw), oracle.make_params(**kw)
    fb = np.zeros((40, 257), np.float32)
    assert sslib.ss_filterbank(C.byref(p), fb.ctypes.data, None) == 0
    assert np.abs(fb - oracle.filterbank(q)[0]).max() < 1e-7
    t = C.c_size_t()
    assert sslib.ss_num_frames(C.byref(p), 16000, C.byref(t)) == 0 and t.value == 101
    assert sslib.ss_num_frames(C.byref(p), 200, C.byref(t)) == 1  # SS_ERR_SHORT_SIGNAL
    bad = make_params(mel_norm="slaney")
    assert sslib.ss_params_validate(C.byref(bad)) == 2  # SS_ERR_BAD_CONFIG


# ------------------------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_librosa_like_front_end_gpu(ss, oracle, sslib):
    import torch

    x = _signal(62, (9, 16000))
    xd = torch.from_numpy(x).cuda()
    for kw in (LIBROSA_LIKE, dict(LIBROSA_LIKE, pad_mode="constant", mel_scale="htk", mel_norm="none"),
               dict(LIBROSA_LIKE, preemph_coef=0.97), dict(sample_rate=16000, mel_scale="slaney", mel_norm="slaney"),
               dict(sample_rate=16000, framing="center")):
        p = oracle.make_params(**kw)
        sw = {k: v for k, v in kw.items() if k not in ("sample_rate", "fft_points", "frame_length", "frame_stride",
                                                       "num_cepstral", "num_filters")}
        args = dict(frame_length=kw.get("frame_length", 0.02), frame_stride=kw.get("frame_stride", 0.01))
        got = ss.mfcc_batch(xd, 16000, **args, **sw).cpu().numpy()
        name = sslib.ss_last_kernel_name().decode()
        if kw.get("preemph_coef"):
            assert name.startswith("ss_mfcc_c256w<")  # fused pre-emphasis with centred frames: the wide-bank kernel
        else:  # builds of the 512-point kernel with centred frames and / or P rows over the whole spectrum
            assert name.startswith("ss_mfcc_c256<16") and ("center" in name) == (kw.get("framing") == "center") \
                and ("fullp" in name) == (kw.get("mel_scale", "reference") != "reference"), name
        assert got.shape[1] == oracle.num_frames(p, 16000)
        for b in (0, 4, 8):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (kw, b)
        feat, en = ss.mfe_batch(xd, 16000, **args, **sw)
        wf, we = oracle.mfe(p, x[8])
        assert _rel(feat[8].cpu().numpy(), wf) <= RTOL and _rel(en[8].cpu().numpy(), we) <= RTOL
    # a clip barely longer than half a frame: every frame touches both edges
    xs = _signal(63, 300)
    p = oracle.make_params(**LIBROSA_LIKE)
    sw = {k: v for k, v in LIBROSA_LIKE.items() if k in ("framing", "pad_mode", "mfcc_window", "spectrum_exponent",
                                                            "mel_scale", "mel_norm", "dct_norm")}
    got = ss.mfcc(xs, 16000, frame_length=512 / 16000, **sw)
    assert _rel(got, oracle.mfcc(p, xs)) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("nfft,hop,mels,kernel", [(2048, 512, 128, "ss_mfcc_c1024<"), (1024, 256, 80, "ss_mfcc_c512<")])
def test_librosa_like_large_fft_gpu(ss, oracle, sslib, nfft, hop, mels, kernel):
    """librosa.feature.melspectrogram's own defaults (22.05 kHz, n_fft = win_length = 2048, hop 512, 128 Slaney mels over
    0..fs/2, centred reflect-padded frames, Hann window, power 2) and the MFCCs on top, on the LIB builds of the 2048-point
    frame kernel -- and the same at n_fft = 1024 / hop 256 / 80 mels on the 1024-point one; each switch also on its own."""
    import torch

    sr, n = 22050, 22050
    base = dict(sample_rate=sr, fft_points=nfft, frame_length=nfft / sr, frame_stride=hop / sr, num_cepstral=20, num_filters=mels)
    lib = dict(framing="center", pad_mode="reflect", mfcc_window="hann", spectrum_exponent=2, mel_scale="slaney",
               mel_norm="slaney", dct_norm="ortho")
    x = _signal(64, (7, n))
    xd = torch.from_numpy(x).cuda()
    margs = dict(frame_length=base["frame_length"], frame_stride=base["frame_stride"], fft_length=nfft, num_filters=mels)
    args = dict(margs, num_cepstral=20)
    for sw in (lib, dict(lib, pad_mode="constant"), dict(framing="center"), dict(mel_scale="htk"),
               dict(mel_scale="slaney", mel_norm="slaney", spectrum_exponent=2)):
        p = oracle.make_params(**base, **sw)
        got = ss.mfcc_batch(xd, sr, **args, **sw).cpu().numpy()
        name = sslib.ss_last_kernel_name().decode()
        assert name.startswith(kernel) and "lib" in name and ("win" in name) == ("mfcc_window" in sw), name
        assert got.shape[1] == oracle.num_frames(p, n)
        for b in (0, 3, 6):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (sw, b)
        feat, en = ss.mfe_batch(xd, sr, **margs, **sw)
        assert "mfe" in sslib.ss_last_kernel_name().decode() and "lib" in sslib.ss_last_kernel_name().decode()
        wf, we = oracle.mfe(p, x[6])
        assert _rel(feat[6].cpu().numpy(), wf) <= RTOL and _rel(en[6].cpu().numpy(), we) <= RTOL, sw
    # a clip barely longer than half a frame: every frame touches both edges; odd frame count -> a dead half-wave
    xs = _signal(65, nfft // 2 + 176)
    p = oracle.make_params(**base, **lib)
    got = ss.mfcc(xs, sr, **args, **lib)
    assert _rel(got, oracle.mfcc(p, xs)) <= RTOL


@pytest.mark.gpu
def test_log_mel_80_centred_gpu(ss, oracle, sslib):
    """The modern speech front end at a power-of-two FFT: 16 kHz, n_fft = win_length = 512, hop 160, 80 Slaney mels over
    0..8 kHz, centred reflect-padded frames, Hann window, power 2 -- mel energies (mfe) and MFCCs -- on the wide-bank kernel."""
    import torch

    sr = 16000
    sw = dict(framing="center", pad_mode="reflect", mfcc_window="hann", spectrum_exponent=2, mel_scale="slaney", mel_norm="slaney",
              dct_norm="ortho")
    x = _signal(66, (6, sr))
    xd = torch.from_numpy(x).cuda()
    for flen, pad in ((512, "reflect"), (400, "constant")):
        s2 = dict(sw, pad_mode=pad)
        p = oracle.make_params(sample_rate=sr, fft_points=512, frame_length=flen / sr, frame_stride=0.01, num_cepstral=13, num_filters=80, **s2)
        margs = dict(frame_length=flen / sr, frame_stride=0.01, fft_length=512, num_filters=80)
        feat, en = ss.mfe_batch(xd, sr, **margs, **{k: v for k, v in s2.items() if k != "dct_norm"})
        name = sslib.ss_last_kernel_name().decode()
        assert name.startswith("ss_mfcc_c256w<") and "mfe" in name and "win" in name, name
        assert feat.shape == (6, oracle.num_frames(p, sr), 80)
        for b in (0, 5):
            wf, we = oracle.mfe(p, x[b])
            assert _rel(feat[b].cpu().numpy(), wf) <= RTOL and _rel(en[b].cpu().numpy(), we) <= RTOL, (flen, b)
        got = ss.mfcc_batch(xd, sr, num_cepstral=13, **margs, **s2).cpu().numpy()
        assert sslib.ss_last_kernel_name().startswith(b"ss_mfcc_c256w<")
        for b in (0, 5):
            assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (flen, b)
    # a clip barely longer than half a frame: every frame touches both edges
    xs = _signal(67, 300)
    p = oracle.make_params(sample_rate=sr, fft_points=512, frame_length=512 / sr, frame_stride=0.01, num_cepstral=13, num_filters=80, **sw)
    got = ss.mfcc(xs, sr, frame_length=512 / sr, num_filters=80, **sw)
    assert _rel(got, oracle.mfcc(p, xs)) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("sr,nfft,flen,hop,M,C,kernel", [(16000, 512, 401, 160, 40, 13, "ss_mfcc_c256<"), (16000, 512, 322, 161, 80, 13, "ss_mfcc_c256w<"),
                                                        (22050, 1024, 883, 221, 64, 20, "ss_mfcc_c512<"), (44100, 2048, 1765, 441, 128, 20, "ss_mfcc_c1024<")])
def test_centred_frames_of_any_length_gpu(ss, oracle, sslib, sr, nfft, flen, hop, M, C, kernel):
    """Centred frames whose length is odd or not a multiple of four (centre offset flen // 2, frame starts of either parity),
    reflect and zero padding, on the dedicated kernels; clips short enough that every frame touches an edge included."""
    import torch

    for n in (flen + 11 * hop + 1, flen // 2 + 3):
        x = _signal(68, (3, n))
        for pad, win in (("reflect", "hann"), ("constant", "hann"), ("reflect", "rect"), ("constant", "rect")):
            sw = dict(framing="center", pad_mode=pad, mfcc_window=win)  # (a window hides the sample behind an odd frame)
            kw = dict(frame_length=flen / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, fft_length=nfft)
            p = oracle.make_params(sample_rate=sr, fft_points=nfft, frame_length=flen / sr, frame_stride=hop / sr, num_cepstral=C, num_filters=M, **sw)
            got = ss.mfcc_batch(torch.from_numpy(x).cuda(), sr, **kw, **sw).cpu().numpy()
            assert sslib.ss_last_kernel_name().decode().startswith(kernel), sslib.ss_last_kernel_name()
            assert got.shape[1] == oracle.num_frames(p, n)
            for b in range(3):
                assert _rel(got[b], oracle.mfcc(p, x[b])) <= RTOL, (n, pad, b)

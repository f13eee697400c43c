"""One-off robustness run for the STFT path: 600 random mel-spectrogram configurations (fft sizes incl. lengths that are not powers
of two, odd hops / lengths / offsets, Slaney / HTK banks, every third one on poisoned LDS) against the oracle.
SS_SWEEP_SEED selects the seed."""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
for d in ('mfcc-rust_amd', 'oracle', 'tests'): sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
import speechsauce_amd as ss
import oracle_c as oracle
from test_gpu_sweep import _rel
lib = ss._lib.lib()
rng = np.random.default_rng(int(os.environ.get("SS_SWEEP_SEED", "777")))
ran = bad = 0; kernels = {}
for i in range(600):
    sr = int(rng.choice([8000, 16000, 22050, 44100]))
    n_fft = int(rng.choice([256, 400, 512, 512, 1000, 1024, 2048, 2048, 4096]))
    hop = int(rng.integers(max(4, n_fft // 16), n_fft // 2 + 1))
    if rng.random() < 0.5: hop &= ~1
    M = int(rng.choice([20, 40, 64, 80, 128, 23]))
    kw = dict(sample_rate=sr, fft_points=n_fft, frame_length=(hop + 0.5) / sr, frame_stride=(hop + 0.5) / sr, num_cepstral=13,
              num_filters=M, low_frequency=0.0, high_frequency=float(sr / 2 * rng.choice([1.0, 0.9])))
    sw = {}
    if rng.random() < 0.3: sw.update(mel_scale=str(rng.choice(["slaney", "htk"])), mel_norm=str(rng.choice(["none", "slaney"])))
    ch = int(rng.choice([1, 2, 5])); n = int(rng.integers(hop * 3, hop * 40))
    try:
        p = oracle.make_params(**kw, **sw); oracle.filterbank(p); oracle.stft_rows(p, n)
    except oracle.OracleError:
        continue
    x = (np.random.default_rng(100 + i).standard_normal((ch, n + 1)) * 0.1).astype(np.float32)
    xd = torch.from_numpy(x).cuda()[:, (i % 2):n + (i % 2)]
    if i % 3 == 0: ss._lib.lab().ss_debug_poison_lds(None)
    try:
        got = ss.mel_spectrogram(xd, sr, frame_length=kw["frame_length"], frame_stride=kw["frame_stride"], num_filters=M, fft_length=n_fft,
                                 high_frequency=kw["high_frequency"], **sw).cpu().numpy()
    except Exception as e:
        bad += 1; print("ERROR", i, e, kw, sw, ch, n); continue
    name = lib.ss_last_kernel_name().decode(); kernels[name] = kernels.get(name, 0) + 1
    want = oracle.mel_spectrogram(p, x[:, (i % 2):n + (i % 2)])
    ran += 1
    if got.shape != want.shape or not _rel(got, want) <= 1e-4:
        bad += 1; print("FAIL", i, name, kw, sw, ch, n, got.shape, want.shape, _rel(got, want) if got.shape == want.shape else None)
print("ran", ran, "bad", bad, kernels)

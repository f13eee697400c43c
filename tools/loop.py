"""Runs one device entry point in a loop (for rocprofv3 passes on stages bench.py has no workload for).
usage: python3 tools/loop.py <stft|power|power_frames|stack|cfg2|cfg3|cfg5|cfg2x4|cfg3x4|cfg5x4> [iterations]
(cfgNx4: four independent batches per call = one launch of the batch-table build, rotating over eight distinct batches)"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "mfcc-rust_amd"))
import torch

import speechsauce_amd as ss

what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
x = torch.randn(1024, 16000, device="cuda") * 0.1
if what == "stft":
    fn = lambda: ss.stft(x, 16000, frame_length=0.032, fft_length=2048)
elif what == "power":
    x1 = x.reshape(-1)
    fn = lambda: ss.power_spectrum_of_signal(x1, 16000)
elif what == "power_frames":
    fr = ss.stack_frames(x.reshape(-1), 16000, frame_length=0.02, frame_stride=0.01)
    fn = lambda: ss.power_spectrum(fr, 512)
elif what == "stack":
    x1 = x.reshape(-1)
    fn = lambda: ss.stack_frames(x1, 16000, frame_length=0.02, frame_stride=0.01)
elif what == "cfg2":
    fn = lambda: ss.mfcc_batch(x, 16000)
elif what == "cfg3":
    fn = lambda: ss.mel_spectrogram(x, 16000, frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0)
elif what == "cfg5":
    x5 = torch.randn(512, 44100, device="cuda") * 0.1
    fn = lambda: ss.mfcc_batch(x5, 44100, frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096)
elif what in ("cfg2x4", "cfg3x4", "cfg5x4"):
    n_s, clips = (44100, 512) if what == "cfg5x4" else (16000, 1024)
    xs = [torch.randn(clips, n_s, device="cuda") * 0.1 for _ in range(8)]
    kw = {"cfg2x4": {}, "cfg3x4": dict(frame_length=0.032, frame_stride=0.032, num_filters=128, fft_length=2048, high_frequency=8000.0),
          "cfg5x4": dict(frame_length=4096 / 44100, frame_stride=1024 / 44100, num_cepstral=40, num_filters=256, fft_length=4096)}[what]
    call = ss.mel_spectrogram if what == "cfg3x4" else ss.mfcc_batch
    state = [0]

    def fn():
        state[0] ^= 1
        return call(xs[4 * state[0]: 4 * state[0] + 4], n_s, **kw)
else:
    raise SystemExit(__doc__)
for _ in range(20):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    fn()
e1.record()
torch.cuda.synchronize()
print("%s: %.2f us per call, kernel %s" % (what, e0.elapsed_time(e1) * 1e3 / n, ss._lib.lib().ss_last_kernel_name().decode()))

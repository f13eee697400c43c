#!/bin/bash
# librosa.feature.melspectrogram-style front end at n_fft = 2048 (centred reflect-padded frames, Hann, power 2, 128 Slaney
# mels over 0..fs/2, ortho DCT) on 512 x 1 s clips @44.1 kHz: LIB build of ss_mfcc_c1024 vs the generic kernel, us per launch
P='{"fft_points": 2048, "frame_length": 0.046439909297052155, "frame_stride": 0.011609977324263039, "num_filters": 128, "num_cepstral": 20, "framing": "center", "pad_mode": "reflect", "mfcc_window": "hann", "spectrum_exponent": 2, "mel_scale": "slaney", "mel_norm": "slaney", "dct_norm": "ortho"}'
for g in 0 1; do
  G=""; if [ $g = 1 ]; then G="--force-generic"; fi
  python bench.py $G --workload cfg5 --params "$P" --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us', round(d['roofline']['frac'],3))"
done

#!/bin/bash
# mel_spectrogram at fft_points = 1024 (16 kHz, 80 mels, 32 ms and 16 ms chunks) on 1024 x 1 s clips: ss_mel_c512 vs the generic kernel
for P in '{"fft_points": 1024, "frame_length": 0.032, "frame_stride": 0.032, "num_filters": 80}' '{"fft_points": 1024, "frame_length": 0.016, "frame_stride": 0.016, "num_filters": 80}'; do
for g in 0 1; do
  G=""; if [ $g = 1 ]; then G="--force-generic"; fi
  python bench.py $G --workload cfg3 --params "$P" --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us', round(d['roofline']['frac'],3), d['config']['frames_per_clip'], 'rows/clip')"
done
done

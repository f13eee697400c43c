#!/bin/bash
# mel_spectrogram at fft_points = 4096 (44.1 kHz, 1024-sample chunks, 256 mels) on 512 x 1 s clips: ss_mel_c2048 vs the generic kernel
P='{"frame_length": 0.023219954648526078, "frame_stride": 0.023219954648526078}'
for g in 0 1; do
  G=""; if [ $g = 1 ]; then G="--force-generic"; fi
  python bench.py $G --workload cfg5 --kind mel --params "$P" --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us', round(d['roofline']['frac'],3), d['config']['frames_per_clip'], 'rows/clip')"
done

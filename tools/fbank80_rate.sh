#!/bin/bash
# 16 kHz log-mel front end (fft_points = 512, 25 ms frames, 10 ms hop, 80 mels; MFCC with 13 cepstra on top) on 1024 x 1 s
# clips: the wide-bank kernel ss_mfcc_c256w vs the generic kernel, us per launch
P='{"frame_length": 0.025, "num_filters": 80}'
for g in 0 1; do
  G=""; if [ $g = 1 ]; then G="--force-generic"; fi
  python bench.py $G --workload cfg2 --params "$P" --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us', round(d['roofline']['frac'],3))"
done

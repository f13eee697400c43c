#!/bin/bash
# 8 kHz telephony front end (fft_points = 256, 20 ms frames, 10 ms hop, 40 mels, 13 cepstra) on 1024 x 2 s clips (202 752 frames):
# ss_mfcc_c256x2 vs the generic kernel, us per launch
P='{"sample_rate": 8000, "fft_points": 256}'
for g in 0 1; do
  G=""; if [ $g = 1 ]; then G="--force-generic"; fi
  python bench.py $G --workload cfg2 --params "$P" --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us', round(d['roofline']['frac'],3), round(d['value']/1e9,2), 'G frames/s')"
done

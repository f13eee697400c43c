#!/bin/bash
# Interleaved A/B of two builds on ONE device (methodology rule: never rank builds across devices).
# usage: tools/ab_bench.sh <libA.so> <libB.so> [rounds] [extra bench args]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for L in $A $B; do
    SS_LIB_PATH=$L python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$L'.split('/')[-1], d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), 'us')"
  done
done

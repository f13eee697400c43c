#!/bin/bash
# On the GPU box: interleaved timing of several libraries on one device, any workload.
# usage: tools/ab_multi.sh "libA libB ..." rounds [bench args]     (names resolve to ab/lib_<name>.so)
V=$1; R=${2:-3}; shift 2
for i in $(seq $R); do
  for n in $V; do
    SS_LIB_PATH=$PWD/ab/lib_$n.so python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('variant', '$n', r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', round(r.get('clock_ghz_measured') or 0,3))"
  done
done

#!/bin/bash
# On the GPU box: interleaved timing of ab/lib_<name>.so builds with optional environment knobs, any workload.
# usage: tools/ab_env.sh "name1 name2@ENV=val ..." rounds [bench args]
V=$1; R=${2:-3}; shift 2
for i in $(seq $R); do
  for v in $V; do
    n=${v%%@*}; e=""; [ "$v" != "$n" ] && e=${v#*@}
    env $e SS_LIB_PATH=$PWD/ab/lib_$n.so python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('variant', '$v', r['kernel'], round(r['avg_launch_us'],2), 'us')"
  done
done

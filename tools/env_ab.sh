#!/bin/bash
# Same-box A/B over an environment knob: tools/env_ab.sh VAR "v1 v2 ..." rounds [bench args]
VAR=$1; VALS=$2; R=${3:-2}; shift 3
for i in $(seq $R); do for v in $VALS; do
  env $VAR=$v python bench.py --steps 3000 --warmup 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), 'us')"
done; done

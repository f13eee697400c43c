#!/bin/bash
# generic kernel on the three BASELINE shapes (--force-generic): us per launch
for w in cfg2 cfg3 cfg5; do
  python bench.py --force-generic --workload $w --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us')"
done

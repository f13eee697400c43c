#!/bin/bash
# r06: the exact commands behind profiles/r06/ -- one sub-command per gpurun call.  Usage on the GPU box:
#   gpurun -- 'bash tools/runs/r06.sh <name>'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
FILT='^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids'
summ() {  # one line per leg of a default bench line
python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('headline', r['kernel'], round(r['avg_launch_us'], 2), 'us frac', round(r['frac'], 4), 'clk', r.get('clock_ghz_measured'), 'value', '%.4g' % d['value'])
print('pipelined', d.get('value_pipelined'), {k: v for k, v in (d.get('pipelined') or {}).items() if k != 'note'})
for k, v in (d.get('secondary') or {}).items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ('kernel', 'avg_launch_us', 'launch_us', 'frac', 'clock_ghz_measured', 'cycles_per_launch', 'valu_floor_frac', 'energy_mj_per_launch', 'value', 'error')})
if 'cpu_baseline' in d: print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])
PY
}
case "${1:-}" in
first)   # the new tests, then the driver's command
timeout 1500 python -m pytest tests/test_batches.py tests/test_concurrency.py -m gpu -x -q 2>&1 | grep -v "$FILT" | tail -15
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06/bench_default_first.json 2> gpurun_out/r06/bench_default_first.err
tail -4 gpurun_out/r06/bench_default_first.err
summ gpurun_out/r06/bench_default_first.json
;;
tests)   # the whole GPU suite
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$FILT" | tail -15
;;
spread)  # three runs of the driver's command on ONE box: do the secondary.cfg2 cycles agree?
for i in 1 2 3; do
  ( time python bench.py --steps 20 --warmup 5 --cpu-seconds 2 ) > gpurun_out/r06/bench_default_$i.json 2> gpurun_out/r06/bench_default_$i.err
  grep real gpurun_out/r06/bench_default_$i.err
  summ gpurun_out/r06/bench_default_$i.json
done 2>&1 | tee gpurun_out/r06/box_spread.txt
;;
stampcost)  # do the stamps cost launch time?
python tools/stamp_cost.py 5 2>&1 | grep -v "$FILT" | tee gpurun_out/r06/stamp_cost.txt
;;
*)
echo "usage: $0 {first|tests|spread|stampcost}" >&2
exit 2
;;
esac

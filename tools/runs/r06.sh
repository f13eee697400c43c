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
profile)  # bench lines first (un-profiled), then traces + PMC passes of the same commands, all on this ONE box
O=gpurun_out/r06
python bench.py --no-cpu-baseline > $O/bench_cfg2.json 2>/dev/null
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --workload cfg3 --steps 1000 --warmup 100 > $O/bench_cfg3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg5 --steps 1000 --warmup 100 > $O/bench_cfg5.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg4 --steps 10 --warmup 2 > $O/bench_cfg4.json 2>/dev/null
tools/profile.sh r06_cfg2 > $O/cfg2_pmc_summary.txt 2>&1
tools/profile.sh r06_cfg3 --workload cfg3 --steps 1000 --warmup 100 > $O/cfg3_pmc_summary.txt 2>&1
tools/profile.sh r06_cfg5 --workload cfg5 --steps 1000 --warmup 100 > $O/cfg5_pmc_summary.txt 2>&1
for w in cfg2 cfg3 cfg5; do cp gpurun_out/prof_r06_$w/trace/*/*kernel_stats.csv $O/${w}_kernel_stats.csv; cp gpurun_out/prof_r06_$w/summary.json $O/${w}_pmc_summary.json; done
# the batch-table builds: four batches per launch (bench.py --leg cfgN_x4: the default run's secondary legs, alone)
for w in cfg2 cfg3 cfg5; do
  python bench.py --leg ${w}_x4 > $O/bench_${w}_x4.json 2>/dev/null
  tools/profile.sh r06_${w}_x4 --leg ${w}_x4 > $O/${w}_x4_pmc_summary.txt 2>&1
  cp gpurun_out/prof_r06_${w}_x4/trace/*/*kernel_stats.csv $O/${w}_x4_kernel_stats.csv; cp gpurun_out/prof_r06_${w}_x4/summary.json $O/${w}_x4_pmc_summary.json
done
rm -rf gpurun_out/prof_r06_*  # raw traces: more than gpurun copies back; the summaries above are what is kept
for f in $O/bench_cfg?.json; do python -c "
import json,sys;d=json.load(open('$f'));r=d['roofline'];print('$f', r['kernel'], round(r['avg_launch_us'],2), 'us frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'valu', r.get('valu_floor_frac'))"; done
grep real $O/bench_default.err; summ $O/bench_default.json
for w in cfg2 cfg3 cfg5; do head -2 $O/${w}_kernel_stats.csv | cut -c1-220; head -2 $O/${w}_x4_kernel_stats.csv | cut -c1-220; done
for w in cfg2 cfg3 cfg5; do python tools/power_probe.py --workload $w --inputs ring,zeros --seconds 1.5 2>&1 | grep -v "$FILT"; done > $O/power_probe.txt
SS_PROFILE_TAG="round 6 (final code)" SS_PROFILE_CLOCK_GHZ=$(python -c "import json;print(json.load(open('$O/bench_cfg2.json'))['roofline']['clock_ghz_measured'])") python tools/make_traffic_json.py cfg2=$O/cfg2_pmc_summary.json cfg3=$O/cfg3_pmc_summary.json cfg5=$O/cfg5_pmc_summary.json cfg2_x4=$O/cfg2_x4_pmc_summary.json cfg3_x4=$O/cfg3_x4_pmc_summary.json cfg5_x4=$O/cfg5_x4_pmc_summary.json > /dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/stage_rate.py 2>&1 | grep -v "$FILT" > $O/stage_rate.txt
;;
profile_x4)  # the batch-table part of `profile` alone
O=gpurun_out/r06
for w in cfg2 cfg3 cfg5; do
  python bench.py --leg ${w}_x4 > $O/bench_${w}_x4.json 2>/dev/null
  tools/profile.sh r06_${w}_x4 --leg ${w}_x4 > $O/${w}_x4_pmc_summary.txt 2>&1
  cp gpurun_out/prof_r06_${w}_x4/trace/*/*kernel_stats.csv $O/${w}_x4_kernel_stats.csv; cp gpurun_out/prof_r06_${w}_x4/summary.json $O/${w}_x4_pmc_summary.json
  python -c "
import json;d=json.load(open('$O/bench_${w}_x4.json'));print('${w}_x4', d['kernel'], round(d['avg_launch_us'],2), 'us per batch,', round(d['launch_us'],1), 'us per launch, frac', round(d['frac'],4))"
  head -2 $O/${w}_x4_kernel_stats.csv | tail -1 | cut -c1-200
done
rm -rf gpurun_out/prof_r06_*
SS_PROFILE_TAG="round 6 (final code)" SS_PROFILE_CLOCK_GHZ=2.0 python tools/make_traffic_json.py cfg2_x4=$O/cfg2_x4_pmc_summary.json cfg3_x4=$O/cfg3_x4_pmc_summary.json cfg5_x4=$O/cfg5_x4_pmc_summary.json > /dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
;;
sweeps)  # robustness: random configurations against the oracle (every third on poisoned LDS)
SS_SWEEP_SEED=10000 timeout 1500 python tools/bigsweep.py 2>&1 | grep -v "$FILT" | tail -6 | tee gpurun_out/r06/bigsweep.txt
SS_SWEEP_SEED=10000 timeout 1500 python tools/melsweep.py 2>&1 | grep -v "$FILT" | tail -6 | tee gpurun_out/r06/melsweep.txt
;;
groups)  # 1 / 2 / 4 / 8 batches per launch
python tools/group_sweep.py 2>&1 | grep -v "$FILT" | tee gpurun_out/r06/group_sweep.txt
;;
trace_default)  # rocprofv3 kernel trace of the driver's own command: every kernel of the default line in one trace
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_default && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r06/bench_default_traced.json 2>/dev/null )
cp gpurun_out/prof_default/*/*kernel_stats.csv gpurun_out/r06/default_run_kernel_stats.csv
rm -rf gpurun_out/prof_default
grep "ss::" gpurun_out/r06/default_run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-220
;;
*)
echo "usage: $0 {first|tests|spread|stampcost|profile|profile_x4|sweeps|groups|trace_default}" >&2
exit 2
;;
esac

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -3 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline', r['kernel'], round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'cyc', r.get('cycles_per_launch'), 'valu', r.get('valu_floor_frac'))
print('pipelined', d.get('value_pipelined'), d['value'])
for k,v in (d.get('secondary') or {}).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a not in ('workload','traffic_source','board')}, v.get('board',{}).get('sclk_mhz_mean'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
PY
for w in cfg3 cfg5; do python bench.py --workload $w --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$w 1000 steps', r['kernel'], round(r['avg_launch_us'],2), 'us')"; done
for w in cfg3 cfg5; do python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$w 200 steps', r['kernel'], round(r['avg_launch_us'],2), 'us')"; done

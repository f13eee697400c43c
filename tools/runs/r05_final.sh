cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -3 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline', r['kernel'], round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'clk', round(r['clock_ghz_measured'],3), 'cyc', round(r['cycles_per_launch']), 'mJ', round(r.get('energy_mj_per_launch',0),1), 'valu', round(r.get('valu_floor_frac',0),3))
print('value', d['value'], 'pipelined', d['value_pipelined'], d['pipelined']['streams'], d['pipelined']['ms_per_step'])
for k,v in d['secondary'].items(): print(k, round(v['avg_launch_us'],2), round(v['frac'],4), v.get('clock_ghz_measured'), v.get('cycles_per_launch'), v.get('energy_mj_per_launch'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/r04/bench_final_steps20.json | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', d['value'], d['ms_per_step'], r['kernel'], round(r['avg_launch_us'],2), 'us frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'valu', r.get('valu_floor_frac'), 'cpu', d['cpu_baseline']['value'])"

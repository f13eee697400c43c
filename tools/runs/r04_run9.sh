cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25
python bench.py --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', r.get('clock_ghz_measured'), 'frac', round(r['frac'],4), 'valu', r.get('valu_floor_frac'))"

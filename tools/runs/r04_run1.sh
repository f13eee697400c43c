set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do
  for n in prev new; do
    SS_LIB_PATH=$PWD/ab/lib_$n.so python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('variant', '$n', r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', round(r.get('clock_ghz_measured') or 0,3), 'frac', round(r['frac'],4))"
  done
done
for n in prev new prev new; do echo "== $n"; SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/stage_rate.py 2>&1 | grep -v amdgpu.ids; done

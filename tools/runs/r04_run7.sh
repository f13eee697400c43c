cd $GRAFT_REPO_ROOT
for r in 300 1200 300 1200 60 2400; do
 SS_BENCH_RING_MIB=$r SS_LIB_PATH=$PWD/ab/lib_new.so python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('ring', $r, r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', round(r.get('clock_ghz_measured') or 0,3), 'frac', round(r['frac'],4))"
done
for r in 300 1200 300 1200; do
 SS_BENCH_RING_MIB=$r SS_LIB_PATH=$PWD/ab/lib_new.so python bench.py --workload cfg3 --no-cpu-baseline --steps 500 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('cfg3 ring', $r, r['kernel'], round(r['avg_launch_us'],2), 'us', 'frac', round(r['frac'],4))"
 SS_BENCH_RING_MIB=$r SS_LIB_PATH=$PWD/ab/lib_new.so python bench.py --workload cfg5 --no-cpu-baseline --steps 500 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('cfg5 ring', $r, r['kernel'], round(r['avg_launch_us'],2), 'us', 'frac', round(r['frac'],4))"
done

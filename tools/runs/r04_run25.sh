cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for ms in 200 500 1000; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms $ms 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('prewarm', $ms, round(r['avg_launch_us'],2), r.get('clock_ghz_measured'), r.get('board'))"; done
bash tools/runs/r04_run24.sh

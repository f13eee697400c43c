cd $GRAFT_REPO_ROOT
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
for i in 1 2 3; do for w in 8 9 10 11 12; do SS_WAVES=$w python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('waves', $w, r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', round(r.get('clock_ghz_measured') or 0,3))"; done; done

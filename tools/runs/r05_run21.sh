cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for s in 1 2 3 4 1 2 3 4; do python bench.py --no-cpu-baseline --no-secondary --streams $s --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('streams', $s, 'frames/s %.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3))"; done | tee gpurun_out/r05/streams_cfg2.txt
for s in 1 2 3; do for w in cfg3 cfg5; do python bench.py --no-cpu-baseline --no-secondary --workload $w --streams $s --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w streams', $s, 'rows/s %.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3))"; done; done | tee -a gpurun_out/r05/streams_cfg2.txt

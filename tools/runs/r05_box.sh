cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 --cpu-seconds 2 2>/dev/null | python -c "
import json,sys,os,time
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['secondary']
print('box', os.uname().nodename, time.strftime('%H:%M:%S'), '| cfg2 %.2f us frac %.4f clk %.3f cyc %.0f | pipelined x%.3f | cfg3 %.2f us %.0f cyc | cfg5 %.2f us %.0f cyc | cfg4 %.0f us frac %.4f | board %s W %s MHz' % (r['avg_launch_us'], r['frac'], r['clock_ghz_measured'], r['cycles_per_launch'], d['value_pipelined']/d['value'], s['cfg3']['avg_launch_us'], s['cfg3'].get('cycles_per_launch') or 0, s['cfg5']['avg_launch_us'], s['cfg5'].get('cycles_per_launch') or 0, s['cfg4']['avg_launch_us'], s['cfg4']['frac'], r['board']['power_w_mean'], r['board']['sclk_mhz_mean']))" | tee -a gpurun_out/r05/box_spread_$(date +%s).txt

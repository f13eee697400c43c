cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
# bench lines first (un-profiled), then the traces + PMC passes of the same commands, all on this one box
python bench.py --no-cpu-baseline > gpurun_out/r05/bench_cfg2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --cpu-seconds 4 > gpurun_out/r05/bench_cfg2_steps20.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r05/bench_cfg3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg5 --steps 1000 --warmup 100 > gpurun_out/r05/bench_cfg5.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg4 --steps 10 --warmup 2 > gpurun_out/r05/bench_cfg4.json 2>/dev/null
tools/profile.sh r05_cfg2 > gpurun_out/r05/cfg2_pmc_summary.txt 2>&1
tools/profile.sh r05_cfg3 --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r05/cfg3_pmc_summary.txt 2>&1
tools/profile.sh r05_cfg5 --workload cfg5 --steps 1000 --warmup 100 > gpurun_out/r05/cfg5_pmc_summary.txt 2>&1
for w in cfg2 cfg3 cfg5; do cp gpurun_out/prof_r05_$w/trace/*/*kernel_stats.csv gpurun_out/r05/${w}_kernel_stats.csv; cp gpurun_out/prof_r05_$w/summary.json gpurun_out/r05/${w}_pmc_summary.json; done
rm -rf gpurun_out/prof_r05_*  # raw traces: more than gpurun copies back; the summaries above are what is kept
python tools/stage_rate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/stage_rate.txt
python tools/stft_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/stft_vs_batch.txt
for f in gpurun_out/r05/bench_cfg*.json; do python -c "
import json,sys;d=json.load(open('$f'));r=d['roofline'];print('$f', r['kernel'], round(r['avg_launch_us'],2), 'us frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'valu', r.get('valu_floor_frac'))"; done
head -3 gpurun_out/r05/cfg2_kernel_stats.csv | cut -c1-200
for w in cfg2 cfg3 cfg5; do python tools/power_probe.py --workload $w --inputs ring,zeros --seconds 1.5 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05/power_probe.txt
SS_PROFILE_TAG="round 5 (final code)" SS_PROFILE_CLOCK_GHZ=$(python -c "import json;print(json.load(open('gpurun_out/r05/bench_cfg2.json'))['roofline']['clock_ghz_measured'])") python tools/make_traffic_json.py cfg2=gpurun_out/r05/cfg2_pmc_summary.json cfg3=gpurun_out/r05/cfg3_pmc_summary.json cfg5=gpurun_out/r05/cfg5_pmc_summary.json > /dev/null
cp profiles/pmc_traffic.json gpurun_out/r05/pmc_traffic.json

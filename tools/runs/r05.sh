#!/bin/bash
# r05: the exact commands behind profiles/r05/ -- one sub-command per gpurun call of that round (formerly one file each:
# tools/runs/r05_<name>.sh).  Usage on the GPU box:  gpurun -- 'bash tools/runs/r05.sh <name>'.  A command log, not a maintained tool:
# some steps name lab builds under ab/ that tools/ablate.sh made at the time.
set -u
case "${1:-}" in
run1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -3 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline', r['kernel'], round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'cyc', r.get('cycles_per_launch'), 'valu', r.get('valu_floor_frac'))
print('pipelined', d.get('value_pipelined'), d.get('pipelined'))
for k,v in (d.get('secondary') or {}).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a not in ('workload','traffic_source')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
PY
SS_LIB_PATH=$PWD/ab/lib_prof5.so python tools/prof5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/phase_profile_cfg5_base.txt
;;
run2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -3 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline', r['kernel'], round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'cyc', r.get('cycles_per_launch'), 'valu', r.get('valu_floor_frac'))
print('pipelined', d.get('value_pipelined'), d['value'])
for k,v in (d.get('secondary') or {}).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a not in ('workload','traffic_source','board')}, v.get('board',{}).get('sclk_mhz_mean'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
PY
for w in cfg3 cfg5; do python bench.py --workload $w --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$w 1000 steps', r['kernel'], round(r['avg_launch_us'],2), 'us')"; done
for w in cfg3 cfg5; do python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$w 200 steps', r['kernel'], round(r['avg_launch_us'],2), 'us')"; done
;;
run3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -m gpu -q -k "mel or stft or cfg3" 2>&1 | tail -3
bash tools/ab_multi.sh "nozskip zskip" 4 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_zskip.txt
python tools/secondary_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/secondary_probe.txt
;;
run4)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in nozskip zskip nozskip zskip; do echo "== $v"; SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/power_probe.py --workload cfg3 --inputs ring,zeros --seconds 1.5 2>&1 | grep -v "amdgpu.ids\|^#"; done | tee gpurun_out/r05/power_cfg3_zskip.txt
;;
run5)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
for v in nozskip zskip; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/prof_r05_$v; mkdir -p $OUT
  i=0
  for set in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
    "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TA_BUSY" \
    "TCC_HIT TCC_MISS TCC_REQ"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --workload cfg3 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $OUT > $R/gpurun_out/r05/pmc_cfg3_$v.txt 2>&1
  rm -rf $OUT
done
cd $R
paste gpurun_out/r05/pmc_cfg3_nozskip.txt gpurun_out/r05/pmc_cfg3_zskip.txt | cut -c1-200
;;
run6)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab_multi.sh "nozskip zskip nozskip_samerow zskip_samerow zskip_noload" 3 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_loads.txt
;;
run7)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in p3n p3z; do for inp in ring zeros; do echo "== $v $inp"; SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/prof3.py $inp 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r05/unit_timeline_cfg3.txt
;;
run8)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in p3n p3z; do for inp in ring; do echo "== $v $inp"; SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/prof3.py $inp 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r05/unit_timeline_cfg3_by_wave.txt
;;
run9)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -m gpu -q -k "mel or stft or cfg3" 2>&1 | tail -3
bash tools/ab_multi.sh "old nofair fair fair_nozskip" 4 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_fair.txt
SS_LIB_PATH=$PWD/ab/lib_p3fair.so python tools/prof3.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/unit_timeline_cfg3_fair.txt
;;
run10)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab_multi.sh "old nofair fairb" 4 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_fairb.txt
SS_LIB_PATH=$PWD/ab/lib_p3fairb.so python tools/prof3.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/unit_timeline_cfg3_fairb.txt
;;
run11)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SS_LIB_PATH=$PWD/ab/lib_prof5.so python tools/prof5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/phase_profile_cfg5_by_wave.txt
;;
run12)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -m gpu -q -x -k "mel or stft or cfg3 or graph" 2>&1 | tail -3
bash tools/ab_env.sh "pool@SS_NOPOOL=1 pool" 5 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_pool.txt
for e in 1 0; do echo "== SS_NOPOOL=$e"; SS_NOPOOL=$e SS_LIB_PATH=$PWD/ab/lib_p3pool.so python tools/prof3.py ring 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/unit_timeline_cfg3_pool.txt
;;
run13)
cd $GRAFT_REPO_ROOT
python tools/secondary_probe.py 2>&1 | grep -v amdgpu.ids | head -6
for w in cfg3 cfg5; do python tools/power_probe.py --workload $w --inputs ring --seconds 1.0 2>&1 | grep -v "amdgpu.ids\|^#"; done
;;
run14)
cd $GRAFT_REPO_ROOT
python tools/clock_probe_ab.py 2>&1 | grep -v amdgpu.ids
;;
run15)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/clock_probe_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/clock_probe_check.txt
;;
run16)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{ bash tools/ab_multi.sh "base ntload ntstore" 3 --workload cfg3
bash tools/traffic_ab.sh "base ntload ntstore" --workload cfg3 --no-secondary ; } 2>&1 | tee gpurun_out/r05/ab_cfg3_nt.txt
rm -rf gpurun_out/tab_*
;;
run17)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SS_LIB_PATH=$PWD/ab/lib_prof2.so python tools/prof2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/phase_profile_cfg2_by_wave.txt
;;
run18)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bench_default" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
;;
run19)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
bash tools/ab_multi.sh "c5_base c5_s1 c5_s2 c5_s4 c5_s5 c5_s6 c5_s9" 2 --workload cfg5
bash tools/ab_multi.sh "c3_base c3_s1 c3_s2 c3_s4 c3_s5 c3_s6 c3_s9" 2 --workload cfg3
bash tools/ab_multi.sh "c2_base c2_s1 c2_s2 c2_s4 c2_s5 c2_s6 c2_s9" 2
} 2>&1 | tee gpurun_out/r05/ab_sched_strategies.txt
;;
run20)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "nan_sample" 2>&1 | grep -E "passed|failed|Error|assert|cfg" | tail -12
;;
run21)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for s in 1 2 3 4 1 2 3 4; do python bench.py --no-cpu-baseline --no-secondary --streams $s --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('streams', $s, 'frames/s %.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3))"; done | tee gpurun_out/r05/streams_cfg2.txt
for s in 1 2 3; do for w in cfg3 cfg5; do python bench.py --no-cpu-baseline --no-secondary --workload $w --streams $s --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w streams', $s, 'rows/s %.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3))"; done; done | tee -a gpurun_out/r05/streams_cfg2.txt
;;
run22)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for w in 12 16 12 16; do echo "== SS_WAVES=$w"; SS_WAVES=$w SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so python tools/power_probe.py --workload cfg2 --inputs ring --seconds 1.5 2>&1 | grep -v "amdgpu.ids\|^#"; done | tee gpurun_out/r05/power_cfg2_waves16.txt
;;
run23)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
tools/ubench/bin/lds_exec_groups 2>&1 | tee gpurun_out/r05/lds_exec_groups_ubench.txt
;;
box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 --cpu-seconds 2 2>/dev/null | python -c "
import json,sys,os,time
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['secondary']
print('box', os.uname().nodename, time.strftime('%H:%M:%S'), '| cfg2 %.2f us frac %.4f clk %.3f cyc %.0f | pipelined x%.3f | cfg3 %.2f us %.0f cyc | cfg5 %.2f us %.0f cyc | cfg4 %.0f us frac %.4f | board %s W %s MHz' % (r['avg_launch_us'], r['frac'], r['clock_ghz_measured'], r['cycles_per_launch'], d['value_pipelined']/d['value'], s['cfg3']['avg_launch_us'], s['cfg3'].get('cycles_per_launch') or 0, s['cfg5']['avg_launch_us'], s['cfg5'].get('cycles_per_launch') or 0, s['cfg4']['avg_launch_us'], s['cfg4']['frac'], r['board']['power_w_mean'], r['board']['sclk_mhz_mean']))" | tee -a gpurun_out/r05/box_spread_$(date +%s).txt
;;
check)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
python __graft_entry__.py smoke 2>&1 | tail -1
;;
final)
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
;;
profile)
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
;;
profile_cfg3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r05/bench_cfg3.json 2>/dev/null
tools/profile.sh r05_cfg3 --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r05/cfg3_pmc_summary.txt 2>&1
for w in cfg3; do cp gpurun_out/prof_r05_$w/trace/*/*kernel_stats.csv gpurun_out/r05/${w}_kernel_stats.csv; cp gpurun_out/prof_r05_$w/summary.json gpurun_out/r05/${w}_pmc_summary.json; done
rm -rf gpurun_out/prof_r05_*
SS_PROFILE_TAG="round 5 (final code)" SS_PROFILE_CLOCK_GHZ=$(python -c "import json;print(json.load(open('gpurun_out/r05/bench_cfg3.json'))['roofline']['clock_ghz_measured'])") python tools/make_traffic_json.py cfg3=gpurun_out/r05/cfg3_pmc_summary.json > /dev/null
cp profiles/pmc_traffic.json gpurun_out/r05/pmc_traffic.json
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; tail -2 gpurun_out/r05/bench_default.err
grep -c secondary gpurun_out/r05/bench_default.json
;;
sweeps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SS_SWEEP_SEED=9000 timeout 1500 python tools/bigsweep.py 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r05/bigsweep.txt
SS_SWEEP_SEED=9000 timeout 1500 python tools/melsweep.py 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r05/melsweep.txt
cat gpurun_out/r05/bigsweep.txt gpurun_out/r05/melsweep.txt
;;
torchrun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/bench_torchrun_2ranks_one_gpu.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --workload cfg4 --corpus-clips 20000 2>/dev/null | tail -1 > gpurun_out/r05/bench_torchrun_cfg4_2ranks_one_gpu.json
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_torchrun_2ranks_one_gpu.json", "gpurun_out/r05/bench_torchrun_cfg4_2ranks_one_gpu.json"):
    d = json.loads(open(f).read())
    print(f, d["n_gpus"], d["scaling"], d["backend"], round(d["value"] / 1e9, 3), round(d["value_path_only"] / 1e9, 3), d["gather"]["mode"], d["gather"].get("chunks_per_step"), d["gather"]["collectives_timed"], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["scaling_model"].items() if k != "note"})
PY
;;
trace_default)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r05/bench_default_traced.json 2>/dev/null
cp $R/gpurun_out/prof_default/*/*kernel_stats.csv $R/gpurun_out/r05/default_run_kernel_stats.csv
rm -rf $R/gpurun_out/prof_default
grep "ss::" $R/gpurun_out/r05/default_run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
;;
*)
echo "usage: $0 {run1|run2|run3|run4|run5|run6|run7|run8|run9|run10|run11|run12|run13|run14|run15|run16|run17|run18|run19|run20|run21|run22|run23|box|check|final|profile|profile_cfg3|sweeps|torchrun|trace_default}" >&2
exit 2
;;
esac

#!/bin/bash
# r04: the exact commands behind profiles/r04/ -- one sub-command per gpurun call of that round (formerly one file each:
# tools/runs/r04_<name>.sh).  Usage on the GPU box:  gpurun -- 'bash tools/runs/r04.sh <name>'.  A command log, not a maintained tool:
# some steps name lab builds under ab/ that tools/ablate.sh made at the time.
set -u
case "${1:-}" in
run1)
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
;;
run2)
cd $GRAFT_REPO_ROOT
for n in stft2 stft4 stft2 stft4; do SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/loop.py stft 300 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done
SS_LIB_PATH=$PWD/ab/lib_stft2.so tools/prof_loop.sh stft stft_r04a 2>&1 | grep -v amdgpu.ids
;;
run3)
cd $GRAFT_REPO_ROOT
for n in stft2 stftA stftB stft2 stftA stftB; do SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/loop.py stft 300 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done
;;
run4)
cd $GRAFT_REPO_ROOT
for n in stft2 stftC stftD stft2 stftC stftD; do SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/loop.py stft 300 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done
;;
run5)
cd $GRAFT_REPO_ROOT
for n in stft2 stftN stftS stftT stft2 stftN stftS stftT; do SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/loop.py stft 300 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done
;;
run6)
cd $GRAFT_REPO_ROOT
SS_LIB_PATH=$PWD/ab/lib_stft2.so python tools/stft_sweep.py 2>&1 | grep -v amdgpu.ids
;;
run7)
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
;;
run8)
cd $GRAFT_REPO_ROOT
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
for w in 8 12 8 12; do echo "== stft waves $w"; SS_STFT_WAVES=$w python tools/stft_sweep.py 768 1024 2048 4096 2>&1 | grep -v amdgpu.ids; done
;;
run9)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25
python bench.py --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', r.get('clock_ghz_measured'), 'frac', round(r['frac'],4), 'valu', r.get('valu_floor_frac'))"
;;
run10)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_prof5.so python tools/prof5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/phase_profile_cfg5.txt
SS_LIB_PATH=$PWD/ab/lib_prof2.so python tools/prof2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/phase_profile_cfg2.txt
;;
run11)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg3 or mel or stage or sweep" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
for r in 0 1 0 1 0 1; do SS_MEL_ROWS4=$r python tools/loop.py cfg3 500 2>&1 | grep -v amdgpu.ids | sed "s/^/rows4=$r /"; done
;;
run12)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
;;
run13)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg1 or cfg2 or cfg4 or golden or strict or edge or sweep or lds or switches or front or mfe or variants" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
tools/ab_multi.sh "head pair" 6 2>&1 | grep variant
;;
run14)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg5 or 4096 or sweep or lds or golden" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
tools/ab_multi.sh "c5nopf c5pf" 5 --workload cfg5 2>&1 | grep variant
;;
run15)
cd $GRAFT_REPO_ROOT
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
for i in 1 2 3; do for w in 8 9 10 11 12; do SS_WAVES=$w python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('waves', $w, r['kernel'], round(r['avg_launch_us'],2), 'us', 'clk', round(r.get('clock_ghz_measured') or 0,3))"; done; done
;;
run16)
cd $GRAFT_REPO_ROOT
python tools/power_probe.py 2>&1 | grep -v amdgpu.ids
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
echo "== 8 waves per CU"; SS_WAVES=8 python tools/power_probe.py 2>&1 | grep -v "amdgpu.ids\|hwmon"
;;
run17)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(SS_SWEEP_SEED=9000 timeout 1200 python tools/bigsweep.py 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/r04/bigsweep.txt
(SS_SWEEP_SEED=4242 timeout 900 python tools/melsweep.py 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/r04/melsweep.txt
;;
run18)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg3 or mel or stft or stage or 2048 or sweep or lds or golden" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
tools/ab_multi.sh "head winfuse" 5 --workload cfg3 2>&1 | grep variant
;;
run19)
cd $GRAFT_REPO_ROOT
tools/ab_multi.sh "p0 pa pb pc pd pe pf" 3 2>&1 | grep variant
;;
run20)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_multiproc.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -12
;;
run21)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_c5swapb.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4096 or cfg5 or 44" 2>&1 | tail -3
bash tools/ablate_run.sh "c5base c5swapb" 3 --workload cfg5 2>&1 | tee gpurun_out/r04/ab_cfg5_swapb.txt
;;
run22)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
{
python -c "
import torch
p=torch.cuda.get_device_properties(0); print([a for a in dir(p) if 'pci' in a], p.pci_bus_id, p.pci_device_id)
import glob; print(glob.glob('/sys/bus/pci/devices/*/hwmon/hwmon*'))
" 2>&1 | tail -3
python tools/power_probe.py 2>&1 | grep -v "^RCCL\|amdgpu.ids"
SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so SS_WAVES=8 python tools/power_probe.py 2>&1 | grep -v "^RCCL\|amdgpu.ids"
rocm-smi --showpower --showclocks 2>&1 | head -40
} | tee gpurun_out/r04/power_probe.txt
;;
run23)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
LAB=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
{
python tools/power_probe.py 2>&1 | grep -v "^RCCL\|amdgpu.ids"
for w in 8 10; do SS_LIB_PATH=$LAB SS_WAVES=$w python tools/power_probe.py --inputs ring,one,zeros 2>&1 | grep -v "^RCCL\|amdgpu.ids"; done
python tools/power_probe.py --workload cfg3 --inputs ring,pcm16,zeros 2>&1 | grep -v "^RCCL\|amdgpu.ids"
SS_LIB_PATH=$LAB SS_MEL_WAVES=8 python tools/power_probe.py --workload cfg3 --inputs ring 2>&1 | grep -v "^RCCL\|amdgpu.ids"
python tools/power_probe.py --workload cfg5 --inputs ring,pcm16,zeros 2>&1 | grep -v "^RCCL\|amdgpu.ids"
} | tee gpurun_out/r04/power_probe.txt
;;
run24)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
{
for v in a0 a1 a2 a4 a8 a16 a32 a64 a128 a0; do
  echo "## variant $v"
  SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/power_probe.py --inputs ring,zeros 2>&1 | grep -v "^RCCL\|amdgpu.ids\|^# "
done
} | tee gpurun_out/r04/energy_by_stage_cfg2.txt
;;
run25)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for ms in 200 500 1000; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms $ms 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('prewarm', $ms, round(r['avg_launch_us'],2), r.get('clock_ghz_measured'), r.get('board'))"; done
bash tools/runs/r04.sh run24
;;
run26)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
{
bash tools/ablate_run.sh "head scalar" 3 --workload cfg2
bash tools/ablate_run.sh "head scalar" 3 --workload cfg5
bash tools/ablate_run.sh "head scalar" 2 --workload cfg3
} 2>&1 | tee gpurun_out/r04/ab_scalar_work_index.txt
;;
run27)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/ablate_run.sh "t0 t1" 5 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_tight_taps.txt
;;
run28)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
bash tools/ablate_run.sh "head tight" 6 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_tight_taps2.txt
;;
run29)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
{
bash tools/ablate_run.sh "head saddr" 8 --workload cfg2
bash tools/ablate_run.sh "head saddr" 5 --workload cfg3
bash tools/ablate_run.sh "head saddr" 5 --workload cfg5
} 2>&1 | tee gpurun_out/r04/ab_saddr.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
;;
run30)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
{
bash tools/ablate_run.sh "head noatom" 8 --workload cfg2
bash tools/ablate_run.sh "head noatom" 5 --workload cfg3
bash tools/ablate_run.sh "head noatom" 5 --workload cfg5
} 2>&1 | tee gpurun_out/r04/ab_noatom.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
;;
run31)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_p1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -3
bash tools/ablate_run.sh "p0 p1" 8 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_partner_lds.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++; print} END{for(k in a) print k, a[k]/n[k], n[k]}' | tail -6
;;
run32)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for v in c_eeef5a5 c_448cd06 c_c279462 noatom; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/pmcx_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc1.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $OUT 2>&1 | grep -A6 "mfcc_c256" | head -8
  rm -rf $OUT
done
;;
run33)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -2
bash tools/ablate_run.sh "head b128" 10 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_tight_b128.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
cd /tmp && export TMPDIR=/tmp
for v in b128; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/pmcx_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc1.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $OUT 2>&1 | grep -A6 "mfcc_c256" | head -8
  rm -rf $OUT
done
;;
run34)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for v in b0 b1 b2 b4 b8; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/pmcx_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc1.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $OUT 2>&1 | grep -A6 "mfcc_c2048" | head -8
  rm -rf $OUT
done
;;
run35)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
{
bash tools/ablate_run.sh "head mask" 6 --workload cfg5
bash tools/ablate_run.sh "head mask" 6 --workload cfg3
} 2>&1 | tee gpurun_out/r04/ab_lane_mask.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
;;
run36)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1500 python tools/bigsweep.py 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee gpurun_out/r04/bigsweep.txt
timeout 900 python tools/melsweep.py 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee gpurun_out/r04/melsweep.txt
;;
run37)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_prof5.so python tools/prof5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/phase_profile_cfg5.txt
SS_LIB_PATH=$PWD/ab/lib_prof2.so python tools/prof2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/phase_profile_cfg2.txt
;;
run38)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_e1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -2
bash tools/ablate_run.sh "e0 e1" 10 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_early_prefetch.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
;;
run39)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
{
bash tools/ablate_run.sh "head off32" 8 --workload cfg5
bash tools/ablate_run.sh "head off32" 6 --workload cfg3
} 2>&1 | tee gpurun_out/r04/ab_off32.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
;;
run40)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
{
bash tools/ablate_run.sh "head cst" 8 --workload cfg3
bash tools/ablate_run.sh "head cst" 8 --workload cfg5
} 2>&1 | tee gpurun_out/r04/ab_cst.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
;;
final)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/r04/bench_final_steps20.json | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('bench', d['value'], d['ms_per_step'], r['kernel'], round(r['avg_launch_us'],2), 'us frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'valu', r.get('valu_floor_frac'), 'cpu', d['cpu_baseline']['value'])"
;;
profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
# bench lines first (un-profiled), then the traces + PMC passes of the same commands, all on this one box
python bench.py --no-cpu-baseline > gpurun_out/r04/bench_cfg2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --cpu-seconds 4 > gpurun_out/r04/bench_cfg2_steps20.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r04/bench_cfg3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg5 --steps 1000 --warmup 100 > gpurun_out/r04/bench_cfg5.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg4 --steps 10 --warmup 2 > gpurun_out/r04/bench_cfg4.json 2>/dev/null
tools/profile.sh r04_cfg2 > gpurun_out/r04/cfg2_pmc_summary.txt 2>&1
tools/profile.sh r04_cfg3 --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r04/cfg3_pmc_summary.txt 2>&1
tools/profile.sh r04_cfg5 --workload cfg5 --steps 1000 --warmup 100 > gpurun_out/r04/cfg5_pmc_summary.txt 2>&1
for w in cfg2 cfg3 cfg5; do cp gpurun_out/prof_r04_$w/trace/*/*kernel_stats.csv gpurun_out/r04/${w}_kernel_stats.csv; cp gpurun_out/prof_r04_$w/summary.json gpurun_out/r04/${w}_pmc_summary.json; done
rm -rf gpurun_out/prof_r04_*  # raw traces: more than gpurun copies back; the summaries above are what is kept
python tools/stage_rate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/stage_rate.txt
python tools/stft_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/stft_vs_batch.txt
for f in gpurun_out/r04/bench_cfg*.json; do python -c "
import json,sys;d=json.load(open('$f'));r=d['roofline'];print('$f', r['kernel'], round(r['avg_launch_us'],2), 'us frac', round(r['frac'],4), 'clk', r.get('clock_ghz_measured'), 'valu', r.get('valu_floor_frac'))"; done
head -3 gpurun_out/r04/cfg2_kernel_stats.csv | cut -c1-200
;;
*)
echo "usage: $0 {run1|run2|run3|run4|run5|run6|run7|run8|run9|run10|run11|run12|run13|run14|run15|run16|run17|run18|run19|run20|run21|run22|run23|run24|run25|run26|run27|run28|run29|run30|run31|run32|run33|run34|run35|run36|run37|run38|run39|run40|final|profile}" >&2
exit 2
;;
esac

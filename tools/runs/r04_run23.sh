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

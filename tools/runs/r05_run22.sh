cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for w in 12 16 12 16; do echo "== SS_WAVES=$w"; SS_WAVES=$w SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so python tools/power_probe.py --workload cfg2 --inputs ring --seconds 1.5 2>&1 | grep -v "amdgpu.ids\|^#"; done | tee gpurun_out/r05/power_cfg2_waves16.txt

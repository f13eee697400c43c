cd $GRAFT_REPO_ROOT
python tools/power_probe.py 2>&1 | grep -v amdgpu.ids
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
echo "== 8 waves per CU"; SS_WAVES=8 python tools/power_probe.py 2>&1 | grep -v "amdgpu.ids\|hwmon"

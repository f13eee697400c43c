cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg3 or mel or stage or sweep" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
for r in 0 1 0 1 0 1; do SS_MEL_ROWS4=$r python tools/loop.py cfg3 500 2>&1 | grep -v amdgpu.ids | sed "s/^/rows4=$r /"; done

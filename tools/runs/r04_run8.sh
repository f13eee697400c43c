cd $GRAFT_REPO_ROOT
export SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so
for w in 8 12 8 12; do echo "== stft waves $w"; SS_STFT_WAVES=$w python tools/stft_sweep.py 768 1024 2048 4096 2>&1 | grep -v amdgpu.ids; done

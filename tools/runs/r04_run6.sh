cd $GRAFT_REPO_ROOT
SS_LIB_PATH=$PWD/ab/lib_stft2.so python tools/stft_sweep.py 2>&1 | grep -v amdgpu.ids

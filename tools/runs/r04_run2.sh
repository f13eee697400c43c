cd $GRAFT_REPO_ROOT
for n in stft2 stft4 stft2 stft4; do SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/loop.py stft 300 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done
SS_LIB_PATH=$PWD/ab/lib_stft2.so tools/prof_loop.sh stft stft_r04a 2>&1 | grep -v amdgpu.ids

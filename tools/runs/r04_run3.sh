cd $GRAFT_REPO_ROOT
for n in stft2 stftA stftB stft2 stftA stftB; do SS_LIB_PATH=$PWD/ab/lib_$n.so python tools/loop.py stft 300 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done

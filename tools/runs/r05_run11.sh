cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SS_LIB_PATH=$PWD/ab/lib_prof5.so python tools/prof5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/phase_profile_cfg5_by_wave.txt

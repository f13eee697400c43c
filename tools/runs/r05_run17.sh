cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SS_LIB_PATH=$PWD/ab/lib_prof2.so python tools/prof2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/phase_profile_cfg2_by_wave.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab_multi.sh "old nofair fairb" 4 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_fairb.txt
SS_LIB_PATH=$PWD/ab/lib_p3fairb.so python tools/prof3.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/unit_timeline_cfg3_fairb.txt

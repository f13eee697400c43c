cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in p3n p3z; do for inp in ring; do echo "== $v $inp"; SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/prof3.py $inp 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r05/unit_timeline_cfg3_by_wave.txt

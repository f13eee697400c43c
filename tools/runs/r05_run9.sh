cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -m gpu -q -k "mel or stft or cfg3" 2>&1 | tail -3
bash tools/ab_multi.sh "old nofair fair fair_nozskip" 4 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_fair.txt
SS_LIB_PATH=$PWD/ab/lib_p3fair.so python tools/prof3.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/unit_timeline_cfg3_fair.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -m gpu -q -x -k "mel or stft or cfg3 or graph" 2>&1 | tail -3
bash tools/ab_env.sh "pool@SS_NOPOOL=1 pool" 5 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_pool.txt
for e in 1 0; do echo "== SS_NOPOOL=$e"; SS_NOPOOL=$e SS_LIB_PATH=$PWD/ab/lib_p3pool.so python tools/prof3.py ring 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/unit_timeline_cfg3_pool.txt

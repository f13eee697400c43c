cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_e1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -2
bash tools/ablate_run.sh "e0 e1" 10 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_early_prefetch.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_c5swapb.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4096 or cfg5 or 44" 2>&1 | tail -3
bash tools/ablate_run.sh "c5base c5swapb" 3 --workload cfg5 2>&1 | tee gpurun_out/r04/ab_cfg5_swapb.txt

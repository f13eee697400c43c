cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SS_LIB_PATH=$PWD/ab/lib_p1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -3
bash tools/ablate_run.sh "p0 p1" 8 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_partner_lds.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++; print} END{for(k in a) print k, a[k]/n[k], n[k]}' | tail -6

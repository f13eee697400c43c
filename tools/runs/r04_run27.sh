cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/ablate_run.sh "t0 t1" 5 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_tight_taps.txt

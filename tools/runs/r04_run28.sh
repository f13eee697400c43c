cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
bash tools/ablate_run.sh "head tight" 6 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_tight_taps2.txt

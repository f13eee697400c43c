cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
{
bash tools/ablate_run.sh "head scalar" 3 --workload cfg2
bash tools/ablate_run.sh "head scalar" 3 --workload cfg5
bash tools/ablate_run.sh "head scalar" 2 --workload cfg3
} 2>&1 | tee gpurun_out/r04/ab_scalar_work_index.txt

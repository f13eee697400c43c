cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
{
bash tools/ablate_run.sh "head cst" 8 --workload cfg3
bash tools/ablate_run.sh "head cst" 8 --workload cfg5
} 2>&1 | tee gpurun_out/r04/ab_cst.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'

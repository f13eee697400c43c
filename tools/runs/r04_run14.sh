cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg5 or 4096 or sweep or lds or golden" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
tools/ab_multi.sh "c5nopf c5pf" 5 --workload cfg5 2>&1 | grep variant

cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6

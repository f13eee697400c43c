cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_multiproc.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -12

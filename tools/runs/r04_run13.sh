cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg1 or cfg2 or cfg4 or golden or strict or edge or sweep or lds or switches or front or mfe or variants" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
tools/ab_multi.sh "head pair" 6 2>&1 | grep variant

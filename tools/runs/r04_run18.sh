cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg3 or mel or stft or stage or 2048 or sweep or lds or golden" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
tools/ab_multi.sh "head winfuse" 5 --workload cfg3 2>&1 | grep variant

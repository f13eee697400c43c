cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "nan_sample" 2>&1 | grep -E "passed|failed|Error|assert|cfg" | tail -12

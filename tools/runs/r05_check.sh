cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
python __graft_entry__.py smoke 2>&1 | tail -1

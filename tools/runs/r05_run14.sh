cd $GRAFT_REPO_ROOT
python tools/probe_test.py 2>&1 | grep -v amdgpu.ids

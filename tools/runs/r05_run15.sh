cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/clock_probe_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/clock_probe_check.txt

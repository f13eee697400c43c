cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1500 python tools/bigsweep.py 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee gpurun_out/r04/bigsweep.txt
timeout 900 python tools/melsweep.py 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee gpurun_out/r04/melsweep.txt

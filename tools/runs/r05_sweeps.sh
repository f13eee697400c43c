cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SS_SWEEP_SEED=9000 timeout 1500 python tools/bigsweep.py 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r05/bigsweep.txt
SS_SWEEP_SEED=9000 timeout 1500 python tools/melsweep.py 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r05/melsweep.txt
cat gpurun_out/r05/bigsweep.txt gpurun_out/r05/melsweep.txt

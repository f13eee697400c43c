cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(SS_SWEEP_SEED=9000 timeout 1200 python tools/bigsweep.py 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/r04/bigsweep.txt
(SS_SWEEP_SEED=4242 timeout 900 python tools/melsweep.py 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/r04/melsweep.txt

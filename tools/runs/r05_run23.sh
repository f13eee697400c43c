cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
tools/ubench/bin/lds_exec_groups 2>&1 | tee gpurun_out/r05/lds_exec_groups_ubench.txt

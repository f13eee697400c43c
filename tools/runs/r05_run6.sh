cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab_multi.sh "nozskip zskip nozskip_samerow zskip_samerow zskip_noload" 3 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_loads.txt

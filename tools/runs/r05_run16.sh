cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{ bash tools/ab_multi.sh "base ntload ntstore" 3 --workload cfg3
bash tools/traffic_ab.sh "base ntload ntstore" --workload cfg3 --no-secondary ; } 2>&1 | tee gpurun_out/r05/ab_cfg3_nt.txt
rm -rf gpurun_out/tab_*

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
bash tools/ab_multi.sh "c5_base c5_s1 c5_s2 c5_s4 c5_s5 c5_s6 c5_s9" 2 --workload cfg5
bash tools/ab_multi.sh "c3_base c3_s1 c3_s2 c3_s4 c3_s5 c3_s6 c3_s9" 2 --workload cfg3
bash tools/ab_multi.sh "c2_base c2_s1 c2_s2 c2_s4 c2_s5 c2_s6 c2_s9" 2
} 2>&1 | tee gpurun_out/r05/ab_sched_strategies.txt

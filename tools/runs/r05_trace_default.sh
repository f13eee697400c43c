cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r05/bench_default_traced.json 2>/dev/null
cp $R/gpurun_out/prof_default/*/*kernel_stats.csv $R/gpurun_out/r05/default_run_kernel_stats.csv
rm -rf $R/gpurun_out/prof_default
grep "ss::" $R/gpurun_out/r05/default_run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200

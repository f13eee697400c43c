cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r05/bench_cfg3.json 2>/dev/null
tools/profile.sh r05_cfg3 --workload cfg3 --steps 1000 --warmup 100 > gpurun_out/r05/cfg3_pmc_summary.txt 2>&1
for w in cfg3; do cp gpurun_out/prof_r05_$w/trace/*/*kernel_stats.csv gpurun_out/r05/${w}_kernel_stats.csv; cp gpurun_out/prof_r05_$w/summary.json gpurun_out/r05/${w}_pmc_summary.json; done
rm -rf gpurun_out/prof_r05_*
SS_PROFILE_TAG="round 5 (final code)" SS_PROFILE_CLOCK_GHZ=$(python -c "import json;print(json.load(open('gpurun_out/r05/bench_cfg3.json'))['roofline']['clock_ghz_measured'])") python tools/make_traffic_json.py cfg3=gpurun_out/r05/cfg3_pmc_summary.json > /dev/null
cp profiles/pmc_traffic.json gpurun_out/r05/pmc_traffic.json
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; tail -2 gpurun_out/r05/bench_default.err
grep -c secondary gpurun_out/r05/bench_default.json

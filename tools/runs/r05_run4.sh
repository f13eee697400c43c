cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in nozskip zskip nozskip zskip; do echo "== $v"; SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/power_probe.py --workload cfg3 --inputs ring,zeros --seconds 1.5 2>&1 | grep -v "amdgpu.ids\|^#"; done | tee gpurun_out/r05/power_cfg3_zskip.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
{
for v in a0 a1 a2 a4 a8 a16 a32 a64 a128 a0; do
  echo "## variant $v"
  SS_LIB_PATH=$PWD/ab/lib_$v.so python tools/power_probe.py --inputs ring,zeros 2>&1 | grep -v "^RCCL\|amdgpu.ids\|^# "
done
} | tee gpurun_out/r04/energy_by_stage_cfg2.txt

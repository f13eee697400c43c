cd $GRAFT_REPO_ROOT
python tools/secondary_probe.py 2>&1 | grep -v amdgpu.ids | head -6
for w in cfg3 cfg5; do python tools/power_probe.py --workload $w --inputs ring --seconds 1.0 2>&1 | grep -v "amdgpu.ids\|^#"; done

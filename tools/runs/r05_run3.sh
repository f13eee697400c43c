cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -m gpu -q -k "mel or stft or cfg3" 2>&1 | tail -3
bash tools/ab_multi.sh "nozskip zskip" 4 --workload cfg3 2>&1 | tee gpurun_out/r05/ab_cfg3_zskip.txt
python tools/secondary_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/secondary_probe.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/bench_torchrun_2ranks_one_gpu.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --workload cfg4 --corpus-clips 20000 2>/dev/null | tail -1 > gpurun_out/r05/bench_torchrun_cfg4_2ranks_one_gpu.json
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_torchrun_2ranks_one_gpu.json", "gpurun_out/r05/bench_torchrun_cfg4_2ranks_one_gpu.json"):
    d = json.loads(open(f).read())
    print(f, d["n_gpus"], d["scaling"], d["backend"], round(d["value"] / 1e9, 3), round(d["value_path_only"] / 1e9, 3), d["gather"]["mode"], d["gather"].get("chunks_per_step"), d["gather"]["collectives_timed"], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["scaling_model"].items() if k != "note"})
PY

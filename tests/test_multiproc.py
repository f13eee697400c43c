"""world_size-2 (and 3) gloo runs of the clip-sharding / gather logic on CPU.  The per-rank compute here is
the oracle's f32 port (tests may use the oracle); on GPUs the same code path calls the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, result_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mfcc-rust_amd"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    import oracle_c
    from speechsauce_amd.distributed import all_gather_features, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = (np.random.default_rng(42).standard_normal((n_clips, 4000)) * 0.1).astype(np.float32)  # same on all ranks
        p = oracle_c.make_params()
        lo, hi = shard_bounds(n_clips, world, rank)
        local = np.stack([oracle_c.port_mfcc(p, x[b]) for b in range(lo, hi)]) if hi > lo else np.zeros((0, 23, 13), np.float32)
        full = all_gather_features(torch.from_numpy(local), n_clips)
        dist.barrier()
        np.save(os.path.join(result_dir, f"rank{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_clips", [(2, 6), (2, 7), (3, 7)])
def test_sharded_gather_matches_single_process(tmp_path, oracle, world, n_clips):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_clips, str(tmp_path)), nprocs=world, join=True)
    x = (np.random.default_rng(42).standard_normal((n_clips, 4000)) * 0.1).astype(np.float32)
    p = oracle.make_params()
    want = np.stack([oracle.port_mfcc(p, x[b]) for b in range(n_clips)])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)


def test_shard_bounds_cover_exactly():
    from speechsauce_amd.distributed import shard_bounds, shard_sizes

    for n in (0, 1, 7, 8, 1024, 360000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = shard_sizes(n, w)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)

"""Clip-batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

Clips are independent units of the hot path (no cross-clip state: the STFT carry-over is defined per
clip from zero state), so the batch axis shards with NO data-path collective.  The only exchange the
north-star names is the final gather of the [n_frames x n_mfcc] blocks: `gather_features` brings them to
one rank (grouped send / recv: one direct xGMI link per peer), `all_gather_features` to every rank
(RCCL over xGMI on GPUs; gloo on CPU for the tests).
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition: ranks [0, n % world) get one extra item."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int):
    return [shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0] for r in range(world)]


def _backend(group=None) -> str:
    import torch.distributed as dist

    return str(dist.get_backend(group)).lower()


def all_gather_into(out, block, group=None, parts=None):
    """One all-gather of equal blocks: `out` [world * B, ...] <- `block` [B, ...] of every rank.  `parts` (instead of `out`): one
    receive view per rank, e.g. the slices [r, c0:c1] of a [world, shard, ...] result when a shard is gathered chunk by chunk.

    RCCL (backend "nccl") moves device tensors directly over xGMI.  Under gloo (the CPU tests, or several ranks sharing one
    GPU, which RCCL refuses) device blocks are staged through host memory: same result, no claim about speed."""
    import torch
    import torch.distributed as dist

    if _backend(group) == "nccl" or not block.is_cuda:
        if parts is not None:
            dist.all_gather(list(parts), block, group=group)
            return parts
        dist.all_gather_into_tensor(out, block, group=group)
        return out
    world = dist.get_world_size(group)
    host = block.detach().cpu()
    hparts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(hparts, host, group=group)
    if parts is not None:
        for dst_view, h in zip(parts, hparts):
            dst_view.copy_(h.to(dst_view.device, non_blocking=False))
        return parts
    out.copy_(torch.cat(hparts, dim=0).to(out.device, non_blocking=False))
    return out


def gather_into(out, block, dst: int = 0, group=None, parts=None):
    """The north-star's collective: rank `dst` receives `out` [world * B, ...] <- `block` [B, ...] of every rank, in rank order;
    the other ranks only send (`out` is ignored there and may be None).  `parts` (on `dst`, instead of `out`): one receive view
    per rank -- a shard gathered chunk by chunk lands in the slices [r, c0:c1] of one [world, shard, ...] result.

    RCCL (backend "nccl"): torch.distributed.gather = grouped ncclSend / ncclRecv -- every peer has a direct xGMI link to the
    root, so the blocks arrive concurrently and nobody but the root spends HBM on them (an all-gather writes world - 1 foreign
    blocks into every rank's memory).  Under gloo device blocks are staged through host memory (tests; ranks sharing a GPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    if _backend(group) == "nccl" or not block.is_cuda:
        recv = (list(parts) if parts is not None else list(out.chunk(world, dim=0))) if rank == dst else None
        dist.gather(block, gather_list=recv, dst=dst_global, group=group)
        return (parts if parts is not None else out) if rank == dst else None
    host = block.detach().cpu()
    hparts = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
    dist.gather(host, gather_list=hparts, dst=dst_global, group=group)
    if rank != dst:
        return None
    if parts is not None:
        for dst_view, h in zip(parts, hparts):
            dst_view.copy_(h.to(dst_view.device, non_blocking=False))
        return parts
    out.copy_(torch.cat(hparts, dim=0).to(out.device, non_blocking=False))
    return out


def gather_features(local, n_total: int, dst: int = 0, group=None):
    """local: [B_rank, ...] feature block of this rank's shard -> [n_total, ...] on rank `dst`, None elsewhere.

    Uneven shards are padded to the largest shard for the collective and trimmed on the root."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError("local block does not match this rank's shard")
    bmax = max(sizes)
    tail = tuple(local.shape[1:])
    if local.shape[0] == bmax:
        padded = local.contiguous()
    else:
        padded = torch.zeros((bmax,) + tail, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    out = torch.empty((world * bmax,) + tail, dtype=local.dtype, device=local.device) if rank == dst else None
    gather_into(out, padded, dst=dst, group=group)
    if rank != dst:
        return None
    if all(s == bmax for s in sizes):
        return out
    return torch.cat([out[r * bmax: r * bmax + sizes[r]] for r in range(world)], dim=0)


def all_gather_features(local, n_total: int, group=None):
    """local: [B_rank, ...] feature block of this rank's shard -> [n_total, ...] on every rank.

    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match this rank's shard")
    bmax = max(sizes)
    tail = tuple(local.shape[1:])
    if local.shape[0] == bmax:
        padded = local.contiguous()
    else:
        padded = torch.zeros((bmax,) + tail, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    out = torch.empty((world * bmax,) + tail, dtype=local.dtype, device=local.device)
    all_gather_into(out, padded, group=group)
    if all(s == bmax for s in sizes):
        return out
    return torch.cat([out[r * bmax: r * bmax + sizes[r]] for r in range(world)], dim=0)


def mfcc_sharded(signals, sampling_frequency, gather=True, group=None, dst: int = 0, **kwargs):
    """signals: the FULL [B, L] batch (every rank sees the same view, e.g. a memory-mapped corpus);
    each rank computes its contiguous shard on its own GPU.  gather = "root": rank `dst` receives [B, T, C] (the others
    None) -- the north-star's gather; gather = True / "all": every rank receives it (all-gather); False: the local shard."""
    import torch.distributed as dist

    from . import mfcc_batch

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(signals.shape[0], world, rank)
    local = mfcc_batch(signals[lo:hi], sampling_frequency, **kwargs)
    if gather and world > 1:
        if gather == "root":
            return gather_features(local, signals.shape[0], dst, group)
        return all_gather_features(local, signals.shape[0], group)
    return local

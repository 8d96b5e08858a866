"""Multi-GPU sharding of a batch of comparisons: comparisons are independent, so every rank runs all
steps on its own contiguous block with no traffic during compute; ONE all-gather (RCCL over xGMI with the "nccl"
backend, gloo in the CPU tests) reassembles the [[x <= y]] batch."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_results(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Concatenate the ranks' result blocks ([count_r][words]) in rank order into [total][words]."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    width = local.shape[-1]
    if all(hi - lo == sizes[0][1] - sizes[0][0] for lo, hi in sizes):
        out = torch.empty((total, width), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    parts = [torch.empty((hi - lo, width), dtype=local.dtype, device=local.device) for lo, hi in sizes]
    pad = max(hi - lo for lo, hi in sizes)
    padded = [torch.empty((pad, width), dtype=local.dtype, device=local.device) for _ in sizes]
    mine = torch.zeros((pad, width), dtype=local.dtype, device=local.device)
    mine[: local.shape[0]] = local
    dist.all_gather(padded, mine, group=group)
    for p, full in zip(parts, padded):
        p.copy_(full[: p.shape[0]])
    return torch.cat(parts, dim=0)


def all_gather_planes(local: torch.Tensor, group=None) -> torch.Tensor:
    """Gather a bit-major per-bit vector (the wire batch [c_i] or [d],[beta_i]: [planes][B_r][words] on rank r) into the
    BLOCKED global layout [rank][planes][B_r][words] of SURVEY 8(e): every rank's block is one contiguous piece, so a single
    all-gather fills it and no transpose follows; comparison b of rank r, plane i is `out[r, i, b]`.  Equal shard sizes only
    (pad the last shard otherwise: a ragged wire batch has no single rectangular layout)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local.reshape((1,) + tuple(local.shape))
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    # (the collective wants the ranks concatenated along the first axis; the blocked layout is that very memory)
    dist.all_gather_into_tensor(out.view((world * local.shape[0],) + tuple(local.shape[1:])), local.contiguous(), group=group)
    return out


def comm_init_from_torch(engine, group=None) -> None:
    """Give `engine` (this rank's library context) an RCCL communicator of its own through the C ABI (sc_comm_init), using an
    existing torch.distributed group only to ship the 128-byte rendezvous id from rank 0 to the others.  Afterwards
    `engine.allgather(local)` reassembles result blocks without torch (what a non-torch host would call)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], rank, world)

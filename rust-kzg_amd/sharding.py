"""Sharding across the GPUs of one node: whole blobs for batched commit / prove, index ranges
for one large MSM.

Blobs are independent units: rank r takes a contiguous slab of the batch, runs the single-GPU
pipeline on it with its own replica of the fixed-base table, and the 48-byte results are gathered.
There is no data-path collective (nothing is exchanged while computing); the only communication
is one all_gather of the results at the end of a batch, and only if every rank needs all of them.
"""
from typing import Callable, List, Sequence, Tuple


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slab [lo, hi) of rank `rank`; slab sizes differ by at most one; empty slabs allowed."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def commit_sharded(blobs: Sequence[bytes], commit_batch: Callable[[List[bytes]], List[bytes]], dist=None) -> List[bytes]:
    """Commit to `blobs` with the work split over the ranks of `dist` (a torch.distributed-like module
    that is already initialised; None = single process).  `commit_batch` is the per-rank engine
    (the GPU pipeline in production, any equivalent in tests).  Returns all results on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return commit_batch(list(blobs))
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(len(blobs), world, rank)
    mine = commit_batch(list(blobs[lo:hi])) if hi > lo else []
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    out: List[bytes] = []
    for part in gathered:
        out.extend(part)
    if len(out) != len(blobs):
        raise RuntimeError("sharded commit lost results")
    return out


def msm_sharded(n_points: int, msm_partial: Callable[[int, int], bytes], g1_sum: Callable[[List[bytes]], bytes],
                dist=None) -> bytes:
    """One large MSM split by index range: rank r computes the partial sum over points/scalars [lo, hi) with
    `msm_partial(lo, hi)` (a 144-byte Jacobian blst_p1; the GPU engine in production), the partials are
    all-gathered (world x 144 bytes, the only exchange step) and every rank adds them with `g1_sum`
    (kzgamd_g1_sum: a group addition is not a reduction op a collective library offers)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return msm_partial(0, n_points)
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_points, world, rank)
    mine = msm_partial(lo, hi) if hi > lo else bytes(144)  # empty slice: the point at infinity (Z == 0)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if any(len(p) != 144 for p in gathered):
        raise RuntimeError("sharded MSM: malformed partial")
    return g1_sum(gathered)


def gather_results(mine, n_items: int, item_bytes: int, dist, device=None):
    """The one collective of a sharded batch: every rank contributes its slab of fixed-size results (a uint8 tensor of
    (hi - lo) * item_bytes bytes, on `device`) and receives all n_items * item_bytes bytes in batch order.  One
    all_gather of equal-sized pieces — ncclAllGather over RCCL when the tensors live on the GPUs, gloo on the host
    (256 commitments x 48 B = 12 KiB: link bandwidth is irrelevant, it is one latency).  Slabs differ by at most one item
    (shard_range), so every piece is padded to the largest slab and trimmed after the gather."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    spans = [shard_range(n_items, world, r) for r in range(world)]
    cap = max(hi - lo for lo, hi in spans) * item_bytes
    lo, hi = spans[rank]
    if mine.numel() != (hi - lo) * item_bytes or mine.dtype != torch.uint8:
        raise ValueError("gather_results: rank %d must contribute %d bytes" % (rank, (hi - lo) * item_bytes))
    dev = mine.device if device is None else device
    piece = torch.zeros(cap, dtype=torch.uint8, device=dev)
    piece[: mine.numel()] = mine
    out = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, piece)
    parts = [out[r * cap: r * cap + (spans[r][1] - spans[r][0]) * item_bytes] for r in range(world)]
    return torch.cat(parts)

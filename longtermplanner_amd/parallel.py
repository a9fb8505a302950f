"""Multi-GPU sharding of a query batch.

Queries are independent — the only cross-lane step of the hot path, the slowest-joint reduction, is inside a
query — so a batch shards as contiguous query ranges, one rank (process) per GPU, limits replicated, with NO
collective on the data path. Optional, both opt-in: an all_gather of the small switching-time records
(gather_records) and point-to-point sends of trajectory tiles to one rank (gather_trajectories_to_root) — RCCL when
the tensors live on GPUs, gloo on CPU tensors. By default dense trajectories stay on the GPU that sampled them (a
gather into one GPU is bounded by its 7 inbound xGMI links and by 288 GB of HBM).
"""
from typing import Dict, List, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first, count) of this rank's contiguous query range; the remainder goes to the lowest ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def shard_counts(n_total: int, world: int) -> List[int]:
    return [shard_range(n_total, r, world)[1] for r in range(world)]


def gather_records(local: Dict[str, "torch.Tensor"], n_total: int, group=None) -> Dict[str, "torch.Tensor"]:
    """all_gather per-query record tensors (first dim = this rank's shard) into full-batch tensors on every rank.

    Shards may differ by one query, so every shard is padded to the largest one for the collective and the
    padding is dropped afterwards. Works with the nccl backend (= RCCL on ROCm; tensors on the rank's GPU) and
    with gloo (CPU tensors).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = shard_counts(n_total, world)
    longest = max(counts)
    out = {}
    for key, t in local.items():
        if t.shape[0] != counts[dist.get_rank(group)]:
            raise ValueError(f"{key}: shard has {t.shape[0]} queries, expected {counts[dist.get_rank(group)]}")
        pad = torch.zeros((longest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out[key] = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out


def gather_trajectories_to_root(tile: "torch.Tensor", elements: int, root: int = 0, group=None):
    """OPT-IN: send the first `elements` elements of every rank's packed trajectory tile to `root`.

    Point-to-point sends (RCCL send/recv over xGMI with the nccl backend, gloo on CPU tensors), one message per rank,
    after an all_gather of the sizes. Returns, on `root`, a list with one 1-D tensor per rank (its own tile slice for
    itself, not copied) and None elsewhere. The root must have room for the sum: 1 M x 7-DoF trajectories are 386 GB,
    and a gather is bounded by the root's seven inbound xGMI links (~1 TB/s) — this is for small result sets, e.g.
    the first-N-samples rows, not for whole batches.
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    size = torch.tensor([int(elements)], dtype=torch.int64, device=tile.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    if rank != root:
        if sizes[rank]:
            dist.send(tile[: sizes[rank]].contiguous(), dst=root, group=group)
        return None
    parts = []
    for r in range(world):
        if r == root:
            parts.append(tile[: sizes[r]])
            continue
        buf = torch.empty(sizes[r], dtype=tile.dtype, device=tile.device)
        if sizes[r]:
            dist.recv(buf, src=r, group=group)
        parts.append(buf)
    return parts

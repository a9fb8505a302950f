"""Multi-GPU sharding of a query batch.

Queries are independent — the only cross-lane step of the hot path, the slowest-joint reduction, is inside a
query — so a batch shards as contiguous query ranges, one rank (process) per GPU, limits replicated, with NO
collective on the data path. The only optional collective is a gather of the small switching-time records
(RCCL when the tensors live on GPUs, gloo on CPU tensors); dense trajectories stay on the GPU that sampled
them (a gather into one GPU is bounded by its 7 inbound xGMI links and by 288 GB of HBM).
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

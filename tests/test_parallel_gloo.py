"""world_size-2 gloo rehearsal of the multi-GPU path: shard the batch, plan each shard, gather the records.

No GPU exists here, so the per-shard computation is done by the CPU oracle standing in for the planner
(tests may use the oracle); what is under test is the sharding, the counter-based generator and the
gather (longtermplanner_amd/parallel.py) — the same code path bench.py and a multi-GPU caller use over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from longtermplanner_amd import generate_queries, limit_set
    from longtermplanner_amd.parallel import gather_records, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        D, lim = limit_set("panda")
        first, count = shard_range(n_total, rank, world)
        qg, q0, v0, a0 = generate_queries(count, lim, seed=4242, first_query=first)
        r = oracle.Oracle(D, 0.001, **lim).plan_batch(qg, q0, v0, a0, sample=False)
        local = {k: torch.from_numpy(np.ascontiguousarray(r[k])) for k in ("t_scaled", "t_required", "slowest", "traj_len", "status")}
        full = gather_records(local, n_total)
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), **{k: v.numpy() for k, v in full.items()})
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    from longtermplanner_amd import generate_queries, limit_set
    n_total, world = 1001, 2          # odd on purpose: shards of 501 and 500
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "gathered.npz"))
    D, lim = limit_set("panda")
    qg, q0, v0, a0 = generate_queries(n_total, lim, seed=4242)
    ref = oracle_mod.Oracle(D, 0.001, **lim).plan_batch(qg, q0, v0, a0, sample=False)
    for k in ("t_scaled", "t_required", "slowest", "traj_len", "status"):
        assert np.array_equal(got[k], ref[k]), k


def _traj_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from longtermplanner_amd.parallel import gather_trajectories_to_root
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        # ragged "tiles": rank r holds 1000 * (r + 1) + 7 valid elements of a larger buffer
        n = 1000 * (rank + 1) + 7
        tile = torch.full((5000,), -1.0, dtype=torch.float64)
        tile[:n] = torch.arange(n, dtype=torch.float64) + 10000.0 * rank
        parts = gather_trajectories_to_root(tile, n, root=0)
        if rank == 0:
            np.savez(os.path.join(out_dir, "traj.npz"), **{f"r{i}": p.numpy() for i, p in enumerate(parts)})
        else:
            assert parts is None
    finally:
        dist.destroy_process_group()


def test_two_rank_trajectory_gather_to_root(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_traj_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "traj.npz"))
    for r in range(world):
        n = 1000 * (r + 1) + 7
        assert np.array_equal(got[f"r{r}"], np.arange(n, dtype=np.float64) + 10000.0 * r)

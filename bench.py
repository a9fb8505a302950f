#!/usr/bin/env python3
"""bench.py — throughput of the batched planTrajectory hot path on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic queries that is already
resident in HBM: stages 1-3 (switching times, slowest-joint reduction, time scaling), the packed
offsets scan, and the dense q/v/a/j sampler for every plan of the batch. The sampled output of a
1M x 7-DoF batch (~385 GB at 1 ms) exceeds HBM, so the sampler runs over resident chunks that reuse
one output tile (SURVEY.md §7 hard part 5); every byte of every trajectory is still written.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one rank
per GPU, RCCL only for the barrier/timing reduction: query ranges shard with no data-path
collective). Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def chunk_bounds(offsets_host, capacity):
    """Greedy [first, end) plan ranges whose packed size fits `capacity` doubles."""
    n = offsets_host.size - 1
    bounds, first = [], 0
    while first < n:
        end = int(np.searchsorted(offsets_host, offsets_host[first] + np.uint64(capacity), side="right")) - 1
        if end <= first:
            raise RuntimeError("one trajectory does not fit the output tile; raise --tile-gib")
        bounds.append((first, min(end, n)))
        first = min(end, n)
    return bounds


def cpu_baseline(dof, lim, t_sample, seed, sample_switch_only):
    """The CPU oracle (a port: the reference itself cannot be built here) timed on the host cores."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    from longtermplanner_amd import generate_queries
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))   # a one-GPU box's CPU share
    per_thread = 49152 if not sample_switch_only else 1000000
    n = per_thread * cores
    qg, q0, v0, a0 = generate_queries(n, lim, seed=seed)
    orc = oracle.Oracle(dof, t_sample, **lim)
    orc.plan_batch(qg[:64], q0[:64], v0[:64], a0[:64], sample=not sample_switch_only, want_records=False)   # warm-up / page-in

    def work(t):
        return orc.plan_batch(qg, q0, v0, a0, sample=not sample_switch_only, first=t * per_thread, count=per_thread,
                              want_records=False)["n_ok"]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    n1 = min(per_thread, 32768 if not sample_switch_only else 262144)
    t1 = time.perf_counter()
    orc.plan_batch(qg, q0, v0, a0, sample=not sample_switch_only, first=0, count=n1, want_records=False)
    dt1 = time.perf_counter() - t1
    flat = None
    if not sample_switch_only:
        # BASELINE.md §3 second variant: trajectory arrays allocated once per thread and reused ("flat preallocated")
        def work_flat(t):
            return orc.plan_batch(qg, q0, v0, a0, sample="flat", first=t * per_thread, count=per_thread // 2, want_records=False)["n_ok"]
        t2 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work_flat, range(cores)))
        dtf = time.perf_counter() - t2
        t3 = time.perf_counter()
        orc.plan_batch(qg, q0, v0, a0, sample="flat", first=0, count=n1, want_records=False)
        dtf1 = time.perf_counter() - t3
        flat = {"value": cores * (per_thread // 2) / dtf, "one_thread": n1 / dtf1, "unit": "plans/s",
                "sample": f"{per_thread // 2} queries per thread on {cores} threads; {n1} on one thread"}
    return {"value": n / dt, "unit": "plans/s", "cores": cores, "kind": "port",
            "one_thread": {"value": n1 / dt1, "unit": "plans/s", "sample": f"first {n1} queries, {dt1:.1f} s"},
            "flat_preallocated": flat,
            "sample": f"first {n} queries of the same synthetic batch, {per_thread} per thread, "
                      f"{'switching times only' if sample_switch_only else 'full planTrajectory incl. per-plan allocation'}, "
                      f"{dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1_000_000, help="queries per GPU per step")
    ap.add_argument("--limits", default="panda", choices=["panda", "ref", "ref30"])
    ap.add_argument("--t-sample", type=float, default=0.001)
    ap.add_argument("--tile-gib", type=float, default=192.0, help="size of the reused trajectory output tile")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--layout", default="query_major", choices=["query_major", "joint_major"],
                    help="input arrays [n][dof] (the reference's per-query vectors) or SoA [dof][n]")
    ap.add_argument("--max-samples", type=int, default=0, help="store only the first N samples per row (0 = whole trajectories, the reference behaviour)")
    ap.add_argument("--sample-stride", type=int, default=1, help="store every N-th sample per row (1 = every sample, the reference behaviour)")
    ap.add_argument("--f32", action="store_true", help="store float32 rows (same binary64 results, rounded once); default float64 as the reference")
    ap.add_argument("--switch-only", action="store_true", help="config[1]: stages 1-3 only, no sampling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain-stores", action="store_true", help="sampler uses plain instead of non-temporal stores")
    ap.add_argument("--window-gib", type=float, default=0.0, help="DIAGNOSTIC: chunk capacity; chunks rotate through windows of the tile")
    ap.add_argument("--spread", type=int, default=0, help="sampler block->plan interleave factor (0 = library default 64, 1 = plan order)")
    ap.add_argument("--dry-sampler", action="store_true", help="DIAGNOSTIC ONLY: sampler stores without arithmetic (invalid as a result)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / timing reduction (gloo: rehearsal of N ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal for every rank (default: LOCAL_RANK)")
    ap.add_argument("--envelope", default="", metavar="WINDOW:COUNT",
                    help="VARIANT: instead of dense rows, reduce each plan on the device to per-joint [min q, max q] over COUNT windows of WINDOW samples (ltp_envelope_batch)")
    ap.add_argument("--receding", default="", metavar="ROUNDS:K",
                    help="VARIANT (SURVEY §8(f).1): per step ROUNDS receding-horizon cycles on the device — plan, sample the first "
                         "--max-samples samples, restart every query from stored sample K (ltp_replan_states_batch); value counts every replan")
    ap.add_argument("--sample-blocks", type=int, default=0, help="TUNING: size of the sampler's persistent grid (0 = library default)")
    ap.add_argument("--gather", action="store_true", help="also all_gather t_required over RCCL each step (optional path)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from longtermplanner_amd import LongTermPlanner, limit_set

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is not None:
        local_rank = args.device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":      # "nccl" is RCCL on ROCm
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where collective tensors live

    dof, lim = limit_set(args.limits)
    ltp = LongTermPlanner(dof, args.t_sample, device=local_rank, **lim)
    n = args.batch
    # this rank's shard of the global batch: query indices [rank*n, (rank+1)*n), generated on the device
    qg, q0, v0, a0 = ltp.generateQueries(n, seed=args.seed, first_query=rank * n, layout=args.layout)
    if args.max_samples:
        ltp.setMaxSamples(args.max_samples)
    if args.sample_stride > 1:
        ltp.setSampleStride(args.sample_stride)
    if args.sample_blocks:
        ltp._check(ltp._lib.ltp_debug_set_sample_blocks(ltp._h, args.sample_blocks))
    rec_spec = tuple(int(x) for x in args.receding.split(":")) if args.receding else None
    rec_direct = bool(rec_spec) and not args.max_samples      # restart states straight from the records (ltp_state_at_batch)
    env_spec = tuple(int(x) for x in args.envelope.split(":")) if args.envelope else None
    env_out = torch.empty((n, dof, env_spec[1], 2), dtype=torch.float64, device=dev) if env_spec else None
    tile = None
    if not args.switch_only and not env_spec and not rec_direct:
        # one big reused output tile; if this GPU cannot give 192 GiB right now, halve until it can
        gib = args.tile_gib
        while tile is None:
            try:
                tile = torch.empty(int(gib * (1 << 30)) // (4 if args.f32 else 8), dtype=torch.float32 if args.f32 else torch.float64, device=dev)
            except torch.OutOfMemoryError:
                if gib <= 8:
                    raise
                gib /= 2
                torch.cuda.empty_cache()
        args.tile_gib = gib
    offsets_pinned = torch.empty(n + 1, dtype=torch.int64, pin_memory=True)
    batch = None
    gather_buf = [torch.empty(n, dtype=torch.float64, device=cdev) for _ in range(world)] if (args.gather and world > 1) else None
    ev_pairs = []
    n_chunks = 0

    def step(timed):
        nonlocal batch, n_chunks
        if rec_spec:
            # every round: stages 1-3, first-N rows of all plans into the tile (they fit: N is small), new start states
            s0, s1, s2 = q0, v0, a0
            for _ in range(rec_spec[0]):
                batch = ltp.planSwitchTimesBatch(qg, s0, s1, s2, layout=args.layout, batch=batch)
                if rec_direct:
                    s0, s1, s2 = ltp.stateAt(batch, 0, n, rec_spec[1], layout=args.layout)
                    continue
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ltp.sampleBatch(batch, 0, n, tile, streaming=not args.plain_stores, spread=args.spread)
                if timed:
                    e1.record()
                    ev_pairs.append((e0, e1))
                s0, s1, s2 = ltp.replanStates(batch, 0, n, tile, rec_spec[1], layout=args.layout)
            n_chunks = 1
            return
        batch = ltp.planSwitchTimesBatch(qg, q0, v0, a0, layout=args.layout, batch=batch)
        if gather_buf is not None:
            dist.all_gather(gather_buf, batch.t_required.to(cdev))
        if args.switch_only:
            return
        if env_spec:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ltp.envelopeBatch(batch, 0, n, env_spec[0], env_spec[1], out=env_out)
            if timed:
                e1.record()
                ev_pairs.append((e0, e1))
            n_chunks = 1
            return
        offsets_pinned.copy_(batch.offsets, non_blocking=True)
        torch.cuda.current_stream().synchronize()       # chunk boundaries depend on this batch's trajectory lengths
        win = int(args.window_gib * (1 << 30)) // 8 if args.window_gib > 0 else tile.numel()
        nwin = max(1, tile.numel() // win)
        bounds = chunk_bounds(offsets_pinned.numpy().view(np.uint64), win)
        n_chunks = len(bounds)
        for ci, (first, end) in enumerate(bounds):
            view = tile[(ci % nwin) * win:(ci % nwin + 1) * win]
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ltp.sampleBatch(batch, first, end - first, view, streaming=not args.plain_stores, dry=args.dry_sampler, spread=args.spread)
            if timed:
                e1.record()
                ev_pairs.append((e0, e1))

    for _ in range(args.warmup):
        step(False)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # bookkeeping outside the timed region
    status = batch.status.cpu().numpy()
    traj_len = batch.traj_len.cpu().numpy().astype(np.int64)
    stored = -(-traj_len // args.sample_stride)
    stored = np.minimum(stored, args.max_samples) if args.max_samples else stored
    alg_bytes_per_step = int((16 if args.f32 else 32) * dof * stored.sum())   # 32*D*traj_len per plan (f64; stored samples)
    if env_spec:
        alg_bytes_per_step = 16 * dof * env_spec[1] * n                         # what the consumer writes: [min, max] per window
    roofline = None
    if ev_pairs:
        kern_ms = sum(a.elapsed_time(b) for a, b in ev_pairs)
        launches = len(ev_pairs)
        rounds = rec_spec[0] if rec_spec else 1    # receding variant: the lengths of the last round stand for all rounds
        achieved = alg_bytes_per_step * rounds * args.steps / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "sampler_write_bytes.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("limits") == args.limits and rec.get("batch") == n and rec.get("tile_gib") == args.tile_gib:
                    traffic = rec.get("write_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": "k_envelope" if env_spec else "k_sample", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_bytes_per_step // max(n_chunks, 1),
                    "avg_launch_ms": round(kern_ms / launches, 4), "launches_timed": launches}

    if rank == 0:
        out = {
            "metric": "7-DoF trajectory plans/sec (batch 1M)" if dof == 7 else f"{dof}-DoF trajectory plans/sec",
            "value": round(world * n * args.steps * (rec_spec[0] if rec_spec else 1) / elapsed, 1),
            "unit": "plans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if not args.f32 else "f64 (rows stored as f32)",
            "data": "synthetic" if not args.dry_sampler else "DIAGNOSTIC dry sampler: NOT a valid result",
            "config": {
                "workload": (f"{n} x {dof}-DoF queries per GPU per step, limits '{args.limits}', Tsample {args.t_sample} s, "
                             + (f"{rec_spec[0]} receding-horizon cycles per step on the device (plan, " + ("state at sample" if rec_direct else f"first {args.max_samples} samples, replan from stored sample") + f" {rec_spec[1]}); value counts replans; " if rec_spec else "")
                             + ("switching times only (stages 1-3)" if args.switch_only else
                                "no rows stored (ltp_state_at_batch)" if rec_direct else
                                f"on-device envelope consumer: [min q, max q] over {env_spec[1]} windows of {env_spec[0]} samples per joint, no dense rows" if env_spec else
                                (f"full q/v/a/j sampling" if not (args.max_samples or args.sample_stride > 1) else
                                 f"q/v/a/j rows: every {args.sample_stride}-th sample" + (f", first {args.max_samples} stored" if args.max_samples else ""))
                                + f" into a reused {args.tile_gib} GiB tile ({n_chunks} chunks per step)")),
                "batch_per_gpu": n, "dof": dof, "t_sample": args.t_sample, "limits": args.limits, "input_layout": args.layout,
                "sharding": "contiguous query ranges per rank, no data-path collective" + (", RCCL all_gather of t_required" if gather_buf else ""),
                "plans_ok_frac": round(float((status == 0).mean()), 5),
                "mean_traj_len": round(float(traj_len.mean()), 1),
                "bytes_per_plan": round(alg_bytes_per_step / n, 1),
            },
        }
        if roofline:
            out["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dof, lim, args.t_sample, args.seed, args.switch_only)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

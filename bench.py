#!/usr/bin/env python3
"""bench.py — throughput of the batched planTrajectory hot path on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic queries that is already
resident in HBM: stages 1-3 (switching times, slowest-joint reduction, time scaling), the packed
offsets scan, and the dense q/v/a/j sampler for every plan of the batch. The sampled output of a
1M x 7-DoF batch (~385 GB at 1 ms) exceeds HBM, so the sampler runs over resident chunks that reuse
one output tile (SURVEY.md §7 hard part 5); every byte of every trajectory is still written.

Contract: `python bench.py --gpus N --steps K --warmup W`, one rank (process) per GPU; RCCL only for the
barrier / timing reduction: query ranges shard with no data-path collective. Rank 0 prints ONE JSON line.
Launched under torch.distributed.run the ranks come from the environment (RANK / LOCAL_RANK / WORLD_SIZE);
launched plainly with --gpus N > 1, this process spawns the N rank processes itself BEFORE importing torch or
touching a GPU, forwards their output, and exits non-zero if any of them fails — it never reports an
n_gpus: 1 line under an N-GPU request.

The line — the LAST and only JSON line on stdout, held under 4 KB (HEADLINE_BUDGET) because the driver parses the tail
of stdout — carries the headline workload (BASELINE.json configs[2]: 1 M x 7-DoF, 1 ms, full sampling) with its
`roofline` and `cpu_baseline`. The other single-GPU configs run back to back in the same process (S-ref limits,
configs[1] 100 k switching times only, configs[4] 1 M x 30-DoF, the short-row / receding / consumer forms; with N > 1:
configs[3], 10 M queries sharded over the N GPUs, strong scaling): their full records go to --secondary-file
(gpurun_out/bench_secondary.jsonl, one JSON line each), one short text line each to stderr, and the headline
names the file and their count.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# binary64 vector peak: half the 157.3 TFLOP/s binary32 vector rate of the same guide = 256 CUs x 4 SIMDs x 16 lanes/clk x 2 flop x 2.4 GHz
# (tools/f64_issue_probe.hip measures ~2200 binary64 wave instructions per us and CU = 72 TFLOP/s of fused multiply-adds)
F64_VALU_PEAK_TFLOPS = 78.6


def workload_key(wl, n):
    """Names a workload for profiles/bench_counters.json (the committed rocprofv3 --pmc passes a line may quote)."""
    parts = [wl.limits, str(n), "f32" if wl.f32 else "f64"]
    if wl.switch_only:
        parts.append("switch_only" + ("+end_limit" if wl.end_limit else ""))
    if wl.max_samples:
        parts.append(f"first{wl.max_samples}")
    if wl.sample_stride > 1:
        parts.append(f"stride{wl.sample_stride}")
    if wl.envelope:
        parts.append(f"envelope{wl.envelope}" + ("+analytic" if getattr(wl, "envelope_analytic", False) else ""))
    if wl.receding:
        parts.append(f"receding{wl.receding}")
    if wl.in_flight > 1:
        parts.append(f"inflight{wl.in_flight}")
    if wl.walk is not None:
        parts.append("walk" if wl.walk else "nowalk")
    if wl.auto_waves is False:
        parts.append("builder_form")
    if not getattr(wl, "row_verdict", True):
        parts.append("no_row_verdict")
    if wl.table_pass != "auto":
        parts.append(f"tablepass_{wl.table_pass}")
    if wl.semantics != "cpp":
        parts.append(wl.semantics)
    if wl.pow_rule != "libm":
        parts.append(f"pow_{wl.pow_rule}")
    return ":".join(parts)


def committed_counters(key):
    """The entry of profiles/bench_counters.json for this workload, or None. These are PMC measurements of separate rocprofv3 passes
    (counters cannot be read inside this process); a line quotes them with their provenance and never as measured in this run."""
    path = os.path.join(ROOT, "profiles", "bench_counters.json")
    try:
        d = json.load(open(path))
        e = d["workloads"].get(key)
        if e is None:
            return None
        return dict(e, source=f"profiles/bench_counters.json ({d.get('collected', '?')}; committed rocprofv3 passes of this workload, NOT measured in this run)")
    except Exception:
        return None


def shard_range(n_total, rank, world):
    """Contiguous query range of a rank (remainder to the lowest ranks); = ltp_shard_range, parallel.shard_range."""
    base, rem = divmod(int(n_total), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def chunk_bounds(offsets_host, capacity):
    """Greedy [first, end) plan ranges whose packed size fits `capacity` elements."""
    import numpy as np
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
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # threads actually used: one GPU's share of the node's cores — online cores / 8 GPUs per node = 32 on the pool's 256-core hosts
    # (the node's other cores belong to its other GPUs) — and never more than this process may run on. All counts are reported.
    cores = max(1, min(affinity, max(1, (os.cpu_count() or 8) // 8), 32))
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
    return {"value": n / dt, "unit": "plans/s", "cores": cores, "cores_rule": "min(affinity, online cores / 8 GPUs per node, 32)",
            "host_cores_affinity": affinity, "host_cores_online": os.cpu_count(), "kind": "port",
            "one_thread": {"value": n1 / dt1, "unit": "plans/s", "sample": f"first {n1} queries, {dt1:.1f} s"},
            "flat_preallocated": flat,
            "sample": f"first {n} queries of the same synthetic batch, {per_thread} per thread, "
                      f"{'switching times only' if sample_switch_only else 'full planTrajectory incl. per-plan allocation'}, "
                      f"{dt:.1f} s wall"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1_000_000, help="queries per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total queries per step over ALL GPUs, sharded as contiguous ranges (strong scaling; overrides --batch). "
                         "BASELINE.json configs[3] is --gpus 8 --global-batch 10000000")
    ap.add_argument("--limits", default="panda", choices=["panda", "ref", "ref30"])
    ap.add_argument("--t-sample", type=float, default=0.001)
    ap.add_argument("--tile-gib", type=float, default=192.0, help="size of the reused trajectory output tile")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--layout", default="query_major", choices=["query_major", "joint_major"],
                    help="input arrays [n][dof] (the reference's per-query vectors) or SoA [dof][n]")
    ap.add_argument("--max-samples", type=int, default=0, help="store only the first N samples per row (0 = whole trajectories, the reference behaviour)")
    ap.add_argument("--sample-stride", type=int, default=1, help="store every N-th sample per row (1 = every sample, the reference behaviour)")
    ap.add_argument("--f32", action="store_true", help="store float32 rows (same binary64 results, rounded once); default float64 as the reference")
    ap.add_argument("--switch-only", action="store_true", help="config[1]: stages 1-3 only, no sampling (status = the pre-sampling verdict cc:14-39)")
    ap.add_argument("--end-limit", action="store_true",
                    help="with --switch-only / --receding without rows: also run planTrajectory's end-limit check (cc:59-61) without sampling "
                         "(ltp_end_limit_batch), so that status == 0 is exactly planTrajectory's bool")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the primary workload (profiling runs)")
    ap.add_argument("--secondary-file", default=os.path.join(ROOT, "gpurun_out", "bench_secondary.jsonl"),
                    help="where the secondary workloads' full records go, one JSON line each (stdout carries the headline line only)")
    ap.add_argument("--plain-stores", action="store_true", help="sampler uses plain instead of non-temporal stores")
    ap.add_argument("--window-gib", type=float, default=0.0, help="DIAGNOSTIC: chunk capacity; chunks rotate through windows of the tile")
    ap.add_argument("--spread", type=int, default=0, help="sampler block->plan interleave factor (0 = library default 64, 1 = plan order)")
    ap.add_argument("--dry-sampler", action="store_true", help="DIAGNOSTIC ONLY: sampler stores without arithmetic (invalid as a result)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / timing reduction (gloo: rehearsal of N ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal for every rank (default: LOCAL_RANK)")
    ap.add_argument("--envelope", default="", metavar="WINDOW:COUNT",
                    help="VARIANT: instead of dense rows, reduce each plan on the device to per-joint [min q, max q] over COUNT windows of WINDOW samples (ltp_envelope_batch)")
    ap.add_argument("--envelope-analytic", action="store_true",
                    help="with --envelope: ltp_set_envelope_mode(LTP_ENVELOPE_ANALYTIC) — the extreme samples from the roots of q'(m) per run instead of every sample")
    ap.add_argument("--receding", default="", metavar="ROUNDS:K",
                    help="VARIANT (SURVEY §8(f).1): per step ROUNDS receding-horizon cycles on the device — plan, sample the first "
                         "--max-samples samples, restart every query from stored sample K (ltp_replan_states_batch); value counts every replan")
    ap.add_argument("--table-pass", default="auto", choices=["auto", "on", "off"],
                    help="where the sampler's run tables are built: by the table pass (a kernel of its own) or inside the sampler kernel (ltp_set_table_pass)")
    ap.add_argument("--no-walk", action="store_true", help="A/B: forbid k_sample_walk_* (the automatic choice for caps <= 768 samples, float32 rows, every 3rd sample or sparser): capped rows of at most 8 KB float64 / 16 KB float32 per joint then take the table pass, the rest the fused k_sample")
    ap.add_argument("--walk", action="store_true", help="A/B: force k_sample_walk_* (also for rows it is not chosen for automatically: whole or long float64 rows at stride 1-2)")
    ap.add_argument("--no-row-verdict", action="store_true", help="capped rows WITHOUT the end-limit verdict (ltp_sample_batch flags bit 4): the walk kernels stop at the cap; status is then planTrajectory's verdict before its last check (cc:59-61)")
    ap.add_argument("--no-auto-waves", action="store_true", help="A/B: caps of at most 32 samples through the walk kernel's builder / streaming-wave form instead of k_sample_walk_auto_*")
    ap.add_argument("--in-flight", type=int, default=1, help="switching times only: steps alternate between this many planner handles, each on its own stream "
                    "(two batches in flight: the latency-bound queue-B kernel of one step runs under the next step's stages); 1 = one batch at a time")
    ap.add_argument("--table-gib", type=float, default=0.0, help="upper bound of the table-pass workspace (default: library default, 1/16 of device memory)")
    ap.add_argument("--sample-blocks", type=int, default=0, help="TUNING: size of the sampler's persistent grid (0 = library default)")
    ap.add_argument("--gather", action="store_true", help="also all_gather t_required over RCCL each step (optional path)")
    ap.add_argument("--semantics", default="cpp", choices=["cpp", "matlab"],
                    help="VARIANT (SURVEY §8(f).4): 'matlab' follows LTPlanner.m where the C++ translation diverges (ltp_set_semantics); rows then take the table pass")
    ap.add_argument("--pow-rule", default="libm", choices=["libm", "exact"],
                    help="how the reference's pow(x, 3 | 4 | 6 | 1/2) calls are formed (ltp_set_pow_rule): 'libm' = glibc's pow restated operation for operation "
                         "(the library's default: the bits of a gcc + glibc build of the reference), 'exact' = one rounding of the exact product (VARIANT: faster stage kernels)")
    ap.add_argument("--one-process", action="store_true",
                    help="--gpus N from ONE process: one planner handle, stream and host thread per device, device-resident shards "
                         "(ltp_plan_switch_times_multi / ltp_envelope_multi / ltp_state_at_multi), no torch.distributed. Workloads without dense "
                         "rows: --switch-only, --envelope, --receding R:K. With --device D all shards sit on device D (rehearsal)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 launched plainly: still create the torch.distributed process group (world size 1) and run every collective "
                         "of the N > 1 path — barrier, all_reduce(MAX / SUM), all_gather — through it (with --backend nccl: RCCL, device tensors). "
                         "Under torch.distributed.run (WORLD_SIZE in the environment) this happens by itself, also at world size 1")
    ap.add_argument("--no-rccl-check", action="store_true",
                    help="default --gpus 1 run: skip the RCCL self-check (a child process that runs a small batch through the nccl branch at "
                         "world size 1 before this process touches the GPU; its verdict is quoted in the line as 'rccl_world1')")
    ap.add_argument("--checksum", action="store_true",
                    help="add an order-independent checksum of all records of the batch (summed over ranks) to the line: equal for any sharding of one global batch")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes. This parent has not
    imported torch and never touches a GPU; it only forwards the children's output and their verdict."""
    import socket
    if args.backend == "nccl" and args.device is not None:
        print("bench.py: --backend nccl needs one GPU per rank (RCCL cannot run several ranks on one device); "
              "use --backend gloo --device D to rehearse N ranks on one GPU", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {procs.index(p)} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in live:      # exactly the processes started above
                    q.terminate()
    return rc


def rccl_self_check(args):
    """The default single-GPU run has no process group; the N > 1 path's use of torch.distributed over RCCL (init with device_id,
    barrier, all_reduce MAX / SUM of device tensors, all_gather of device tensors, --gather) is the only code of this file such a
    run never executes. This runs it once, at world size 1, in a CHILD process (this process has not touched the GPU yet), on a
    small batch, and returns the child's verdict for the line. It never fails the headline."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-dist", "--backend", "nccl", "--batch", "40000", "--steps", "2",
           "--warmup", "1", "--tile-gib", "16", "--no-secondary", "--no-cpu-baseline", "--gather", "--checksum", "--seed", str(args.seed)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    except Exception as e:
        return {"ok": False, "error": f"{type(e).__name__}: {e}"}
    wall = round(time.perf_counter() - t0, 1)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        return {"ok": False, "returncode": p.returncode, "wall_s": wall, "stderr_tail": p.stderr[-600:]}
    o = json.loads(lines[0])
    return {"ok": True, "wall_s": wall, "command": " ".join(cmd[1:]).replace(ROOT + os.sep, ""), "backend": o["config"]["backend"],
            "rank_devices": o["config"]["rank_devices"], "sharding": o["config"]["sharding"], "value": o["value"], "unit": o["unit"],
            "records_checksum": o["config"].get("records_checksum"), "roofline_frac": (o.get("roofline") or {}).get("frac")}


class Workload:
    """One measured configuration: limits, per-rank shard of the batch and the variant switches."""

    def __init__(self, args, **over):
        self.limits, self.t_sample, self.layout = args.limits, args.t_sample, args.layout
        self.batch, self.global_batch = args.batch, args.global_batch
        self.steps, self.warmup = args.steps, args.warmup
        self.switch_only, self.f32 = args.switch_only, args.f32
        self.max_samples, self.sample_stride = args.max_samples, args.sample_stride
        self.envelope, self.receding = args.envelope, args.receding
        self.envelope_analytic = args.envelope_analytic
        self.plain_stores, self.dry, self.spread, self.window_gib = args.plain_stores, args.dry_sampler, args.spread, args.window_gib
        self.sample_blocks, self.gather, self.checksum, self.seed = args.sample_blocks, args.gather, args.checksum, args.seed
        self.table_pass, self.table_gib = args.table_pass, args.table_gib
        self.walk = False if args.no_walk else (True if args.walk else None)
        self.auto_waves = False if args.no_auto_waves else None
        self.row_verdict = not args.no_row_verdict
        self.end_limit = args.end_limit
        self.in_flight = args.in_flight
        self.semantics = args.semantics
        self.pow_rule = args.pow_rule
        self.name = "primary"
        for k, v in over.items():
            setattr(self, k, v)


def run_workload(wl, ctx):
    """Runs one workload on this rank; returns the JSON object (rank 0) or None."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from longtermplanner_amd import LongTermPlanner, limit_set

    world, rank, dev, cdev, local_rank = ctx["world"], ctx["rank"], ctx["dev"], ctx["cdev"], ctx["local_rank"]
    grouped = ctx["dist"]       # a process group exists (always for world > 1; at world size 1 under torch.distributed.run / --force-dist)
    dof, lim = limit_set(wl.limits)
    ltp = LongTermPlanner(dof, wl.t_sample, device=local_rank, **lim)
    if wl.semantics != "cpp":
        ltp.setSemantics(wl.semantics)
    ltp.setPowRule(wl.pow_rule)
    ltp.setEnvelopeMode("analytic" if wl.envelope_analytic else "exhaustive")      # explicit: the library's default is analytic since round 6
    if wl.global_batch:
        first_query, n = shard_range(wl.global_batch, rank, world)
        total_queries = wl.global_batch
    else:
        n = wl.batch
        first_query, total_queries = rank * n, world * n
    # this rank's shard of the global batch: query indices [first_query, first_query + n), generated on the device
    qg, q0, v0, a0 = ltp.generateQueries(n, seed=wl.seed, first_query=first_query, layout=wl.layout)
    if wl.max_samples:
        ltp.setMaxSamples(wl.max_samples)
    if wl.sample_stride > 1:
        ltp.setSampleStride(wl.sample_stride)
    if wl.sample_blocks:
        ltp._check(ltp._lib.ltp_debug_set_sample_blocks(ltp._h, wl.sample_blocks))
    if wl.table_pass != "auto" or wl.table_gib:
        ltp.setTablePass({"auto": 0, "on": 1, "off": -1}[wl.table_pass], int(wl.table_gib * (1 << 30)) if wl.table_gib else None)
    rec_spec = tuple(int(x) for x in wl.receding.split(":")) if wl.receding else None
    rec_direct = bool(rec_spec) and not wl.max_samples      # restart states straight from the records (ltp_state_at_batch)
    env_spec = tuple(int(x) for x in wl.envelope.split(":")) if wl.envelope else None
    env_out = torch.empty((n, dof, env_spec[1], 2), dtype=torch.float64, device=dev) if env_spec else None
    tile = None
    tile_gib = None
    if not wl.switch_only and not env_spec and not rec_direct:
        tile_bytes, tile_gib = ctx["tile"]()
        tile = tile_bytes.view(torch.float32 if wl.f32 else torch.float64)
    offsets_pinned = torch.empty(n + 1, dtype=torch.int64, pin_memory=True)
    batch = None
    # --in-flight K (switching times only): K handles (each owns its device workspace), K streams, K record sets
    lanes = None
    if wl.in_flight > 1:
        if not wl.switch_only or wl.receding:
            raise SystemExit("bench.py: --in-flight applies to --switch-only")
        # the streams are made once per process and shared by the workloads: HIP deals streams to a few hardware queues, and
        # two streams that land on the same queue do not overlap (seen: the second in-flight workload of a run gained nothing)
        pool = ctx.setdefault("streams", [])
        while len(pool) < wl.in_flight:
            pool.append(torch.cuda.Stream())
        lanes = [{"ltp": ltp if i == 0 else LongTermPlanner(dof, wl.t_sample, device=local_rank, **lim), "stream": pool[i], "batch": None}
                 for i in range(wl.in_flight)]
        for lane in lanes[1:]:
            if wl.semantics != "cpp":
                lane["ltp"].setSemantics(wl.semantics)
            lane["ltp"].setPowRule(wl.pow_rule)
    step_no = 0
    gather_buf = [torch.empty(n, dtype=torch.float64, device=cdev) for _ in range(world)] if (wl.gather and grouped and not wl.global_batch) else None
    ev_pairs = []
    n_chunks = 0

    def step(timed):
        nonlocal batch, n_chunks, step_no
        if lanes:
            lane = lanes[step_no % len(lanes)]
            step_no += 1
            with torch.cuda.stream(lane["stream"]):
                lane["batch"] = lane["ltp"].planSwitchTimesBatch(qg, q0, v0, a0, layout=wl.layout, batch=lane["batch"], end_limit=wl.end_limit)
            batch = lane["batch"]
            return
        if rec_spec:
            # every round: stages 1-3 (+ end-limit verdict), first-N rows of all plans into the tile (they fit: N is small), new start states
            s0, s1, s2 = q0, v0, a0
            for _ in range(rec_spec[0]):
                batch = ltp.planSwitchTimesBatch(qg, s0, s1, s2, layout=wl.layout, batch=batch, end_limit=rec_direct and wl.end_limit)
                if rec_direct:
                    s0, s1, s2 = ltp.stateAt(batch, 0, n, rec_spec[1], layout=wl.layout)
                    continue
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ltp.sampleBatch(batch, 0, n, tile, streaming=not wl.plain_stores, spread=wl.spread, walk=wl.walk, auto_waves=wl.auto_waves, verdict=wl.row_verdict)
                if timed:
                    e1.record()
                    ev_pairs.append((e0, e1))
                s0, s1, s2 = ltp.replanStates(batch, 0, n, tile, rec_spec[1], layout=wl.layout)
            n_chunks = 1
            return
        # switching times only: with --end-limit the check of cc:59-61 runs without the sampler and status is planTrajectory's bool
        batch = ltp.planSwitchTimesBatch(qg, q0, v0, a0, layout=wl.layout, batch=batch, end_limit=wl.switch_only and wl.end_limit)
        if gather_buf is not None:
            dist.all_gather(gather_buf, batch.t_required.to(cdev))
        if wl.switch_only:
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
        win = int(wl.window_gib * (1 << 30)) // 8 if wl.window_gib > 0 else tile.numel()
        nwin = max(1, tile.numel() // win)
        bounds = chunk_bounds(offsets_pinned.numpy().view(np.uint64), win)
        n_chunks = len(bounds)
        for ci, (first, end) in enumerate(bounds):
            view = tile[(ci % nwin) * win:(ci % nwin + 1) * win]
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ltp.sampleBatch(batch, first, end - first, view, streaming=not wl.plain_stores, dry=wl.dry, spread=wl.spread, walk=wl.walk, auto_waves=wl.auto_waves, verdict=wl.row_verdict)
            if timed:
                e1.record()
                ev_pairs.append((e0, e1))

    for _ in range(wl.warmup):
        step(False)

    def sync_all():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(wl.steps):
        step(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # bookkeeping outside the timed region
    status = batch.status.cpu().numpy()
    traj_len = batch.traj_len.cpu().numpy().astype(np.int64)
    stored = -(-traj_len // wl.sample_stride)
    stored = np.minimum(stored, wl.max_samples) if wl.max_samples else stored
    alg_bytes_per_step = int((16 if wl.f32 else 32) * dof * stored.sum())   # 32*D*traj_len per plan (f64; stored samples)
    if env_spec:
        alg_bytes_per_step = 16 * dof * env_spec[1] * n                         # what the consumer writes: [min, max] per window
    checksum = None
    if wl.checksum:
        # order-independent: sums of the raw 64-bit patterns (mod 2^63) of every record array of this shard, summed over ranks
        # (a wrapping int64 sum is the true sum mod 2^64, hence also mod 2^48; the small residues add up without overflow)
        M = 1 << 48
        acc = 0
        for x in (batch.t_opt, batch.t_scaled, batch.dir, batch.v_drive, batch.t_required):
            acc += int(x.contiguous().view(torch.int64).sum().item()) % M
        for x in (batch.mod, batch.slowest, batch.traj_len, batch.status):
            acc += int(x.to(torch.int64).sum().item()) % M
        c = torch.tensor([acc % M], dtype=torch.int64, device=cdev)
        if grouped:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        checksum = int(c.item()) % M
    # (LTP_STATUS_MATLAB_COMPLEX = 256 is informational: the plan is delivered, planTrajectory returns true)
    counts = torch.tensor([float(((status & ~256) == 0).sum()), float(traj_len.sum()), float(alg_bytes_per_step)], dtype=torch.float64, device=cdev)
    if grouped:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    ok_total, len_total, bytes_total = (float(x) for x in counts.tolist())
    roofline = None
    key = workload_key(wl, n)
    pmc = committed_counters(key)
    if ev_pairs and env_spec:
        # the envelope consumer writes 16 bytes per window: its bound is the binary64 vector rate. Counted work: the cubic q(m) of every
        # sample it looks at, 3 fused multiply-adds = 6 flop (index conversion, min and max are issue slots too but not flops)
        kern_ms = sum(a.elapsed_time(b) for a, b in ev_pairs)
        evals = float(dof) * float(np.minimum(traj_len, env_spec[0] * env_spec[1]).sum())
        if getattr(wl, "envelope_analytic", False):
            evals = None
        tflops = (evals * 6.0 * wl.steps / (kern_ms * 1e-3) / 1e12) if evals else None
        roofline = {"kernel": ltp.lastSamplerKernel(), "bound": "valu_f64", "achieved": round(tflops, 2) if tflops else None, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / F64_VALU_PEAK_TFLOPS, 4) if tflops else None,
                    "counted": "6 flop (the 3 fused multiply-adds of q(m)) per sample inside the windows" if evals else
                               "analytic form: a few candidate samples per run and window instead of every sample; no flop count claimed",
                    "bytes_written_per_launch": alg_bytes_per_step, "avg_launch_ms": round(kern_ms / len(ev_pairs), 4), "launches_timed": len(ev_pairs),
                    "traffic": None, "measured_on": f"rank 0 of {world}" if world > 1 else "rank 0"}
        if pmc:
            roofline["counters_from_profile"] = pmc
            env_k = [v for k, v in (pmc.get("kernels") or {}).items() if "k_envelope" in k]
            if env_k:       # what the kernel is bound by: vector issue slots (index conversion, run lookup, min, max are slots, not flops)
                roofline["issue_slots_used_frac_from_profile"] = env_k[0].get("valu_issue_frac")
    elif ev_pairs:
        kern_ms = sum(a.elapsed_time(b) for a, b in ev_pairs)
        launches = len(ev_pairs)
        rounds = rec_spec[0] if rec_spec else 1    # receding variant: the lengths of the last round stand for all rounds
        achieved = alg_bytes_per_step * rounds * wl.steps / (kern_ms * 1e-3) / 1e9
        roofline = {"kernel": ltp.lastSamplerKernel(), "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    # HBM bytes per launch are a PMC measurement (rocprofv3 --pmc WRITE_SIZE, separate pass): not available inside
                    # this process. The committed measurement of the same workload, if any, is quoted with its provenance.
                    "traffic": None,
                    "algorithmic_bytes_per_launch": alg_bytes_per_step // max(n_chunks, 1),
                    "avg_launch_ms": round(kern_ms / launches, 4), "launches_timed": launches,
                    "measured_on": f"rank 0 of {world}" if world > 1 else "rank 0"}
        if rec_spec:
            roofline["rows"] = "written by the sampler every cycle; the restart states are recomputed from the records (k_replan_walk), not read back from the rows"
        if pmc and pmc.get("write_bytes_per_launch"):
            roofline["traffic_from_profile"] = {"write_bytes_per_launch": pmc["write_bytes_per_launch"], "source": pmc["source"]}
    elif wl.switch_only or rec_direct:
        # stages 1-3 move ~750 bytes and ~10^4 flop per plan: neither HBM nor the flop rate bounds them. What does: the
        # binary64 issue rate of the waves in flight and the latency of the deferred eigen-solves. The share of the vector
        # issue slots in use is a PMC figure: quoted from the committed pass, if there is one for this workload.
        roofline = {"kernel": "k_opt_fast + k_opt_slow + k_reduce_scale + k_scaling_slow" + (" + k_end_limit" if wl.end_limit else "") + (" + k_state_at" if rec_direct else ""),
                    "bound": "latency/valu_f64", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                    "valu_issue_frac": pmc.get("valu_issue_frac") if pmc else None,
                    "counters_from_profile": pmc}

    if rank != 0:
        return None
    replans = rec_spec[0] if rec_spec else 1
    out = {
        "metric": ("7-DoF trajectory plans/sec (batch 1M)" if dof == 7 else f"{dof}-DoF trajectory plans/sec") + (" [MATLAB semantics]" if wl.semantics == "matlab" else ""),
        "value": round(total_queries * wl.steps * replans / elapsed, 1),
        "unit": "plans/s",
        "n_gpus": world,
        "steps": wl.steps,
        "warmup": wl.warmup,
        "ms_per_step": round(elapsed / wl.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong" if wl.global_batch else "weak",
        "vs_baseline": None,
        "dtype": "f64" if not wl.f32 else "f64 (rows stored as f32)",
        "data": "synthetic" if not wl.dry else "DIAGNOSTIC dry sampler: NOT a valid result",
        "config": {
            "workload": ((f"{wl.global_batch} x {dof}-DoF queries per step sharded over {world} GPU(s) ({n} on rank 0)" if wl.global_batch else
                          f"{n} x {dof}-DoF queries per GPU per step") + f", limits '{wl.limits}', Tsample {wl.t_sample} s, "
                         + (f"{rec_spec[0]} receding-horizon cycles per step on the device (plan, " + ("state at sample" if rec_direct else f"first {wl.max_samples} samples written" + ("" if wl.row_verdict else " WITHOUT the end-limit verdict (opt-in, flags bit 4)") + ", replan from the state at stored sample") + f" {rec_spec[1]}" + ("" if rec_direct else ", recomputed from the records") + "); value counts replans; " if rec_spec else "")
                         + (("switching times only (stages 1-3" + (" + end-limit check" if wl.end_limit else "; status = pre-sampling verdict") + ", no rows)"
                             + (f"; {wl.in_flight} batches in flight (steps alternate between {wl.in_flight} handles and streams)" if wl.in_flight > 1 else "")) if wl.switch_only else
                            "no rows stored (ltp_state_at_batch)" if rec_direct else
                            f"on-device envelope consumer ({'analytic: the candidates of each run' if wl.envelope_analytic else 'every sample'}): [min q, max q] over {env_spec[1]} windows of {env_spec[0]} samples per joint, no dense rows" if env_spec else
                            ("full q/v/a/j sampling" if not (wl.max_samples or wl.sample_stride > 1) else
                             f"q/v/a/j rows: every {wl.sample_stride}-th sample" + (f", first {wl.max_samples} stored" if wl.max_samples else "")
                             + ("" if wl.row_verdict or not wl.max_samples else " (OPT-IN: the sampler does not form the end-limit verdict cc:59-61, ltp_sample_batch flags bit 4)"))
                            + f" into a reused {tile_gib} GiB tile ({n_chunks} chunks per step)")),
            "batch_per_gpu": n if not wl.global_batch else None, "global_batch": total_queries, "dof": dof, "t_sample": wl.t_sample,
            "workload_key": key, "limits": wl.limits, "input_layout": wl.layout, "table_pass": wl.table_pass, "batches_in_flight": wl.in_flight, "semantics": wl.semantics, "pow_rule": wl.pow_rule,
            "sharding": "contiguous query ranges per rank, no data-path collective" + (", all_gather of t_required through the process group every step" if gather_buf else ""),
            "backend": ctx["backend"], "rank_devices": ctx["rank_devices"],
            "plans_ok_frac": round(ok_total / total_queries, 5),
            "plans_ok_is": ("planTrajectory's bool" if (wl.end_limit or not (wl.switch_only or rec_direct)) else
                            "the pre-sampling verdict (cc:14-39); the end-limit check cc:59-61 was not run"),
            "mean_traj_len": round(len_total / total_queries, 1),
            "bytes_per_plan": round(bytes_total / total_queries, 1),
        },
    }
    if checksum is not None:
        out["config"]["records_checksum"] = checksum
    if roofline:
        out["roofline"] = roofline
    return out


def run_one_process(args):
    """--one-process: N devices driven from this process through the *_multi entry points (one handle, one stream and one
    host thread per device; shards are device-resident, nothing passes through the host). Returns the JSON object."""
    import numpy as np
    import torch
    from longtermplanner_amd import LongTermPlanner, limit_set

    k = args.gpus
    if not (args.switch_only or args.envelope or (args.receding and not args.max_samples)):
        raise SystemExit("bench.py: --one-process drives the calls that leave no dense rows behind: --switch-only, --envelope W:K or "
                         "--receding R:K (rows stay a per-rank matter: use the default one-process-per-GPU mode for them)")
    ndev = torch.cuda.device_count()
    devices = [args.device] * k if args.device is not None else list(range(k))
    if max(devices) >= ndev:
        raise SystemExit(f"bench.py: --one-process --gpus {k} needs {k} visible devices (found {ndev}); --device D puts all shards on device D")
    dof, lim = limit_set(args.limits)
    total = args.global_batch if args.global_batch else args.batch * k
    planners = [LongTermPlanner(dof, args.t_sample, device=d, **lim) for d in devices]
    streams, inputs, counts = [], [], []
    for g, d in enumerate(devices):
        first, cnt = shard_range(total, g, k)
        with torch.cuda.device(d):
            streams.append(torch.cuda.Stream(device=d))
            with torch.cuda.stream(streams[-1]):
                inputs.append(planners[g].generateQueries(cnt, seed=args.seed, first_query=first, layout=args.layout))
        counts.append(cnt)
    env_spec = tuple(int(x) for x in args.envelope.split(":")) if args.envelope else None
    rec_spec = tuple(int(x) for x in args.receding.split(":")) if args.receding else None
    batches = None
    envs = None
    # restart states: two sets per shard, allocated once and used alternately (a round reads one set and writes the other)
    state_sets = [[[torch.empty_like(inputs[g][1]) for _ in range(3)] for g in range(k)] for _ in range(2)] if rec_spec else None

    def step():
        nonlocal batches, envs
        if rec_spec:
            cur = inputs
            for r in range(rec_spec[0]):
                batches = LongTermPlanner.planSwitchTimesSharded(planners, cur, total, layout=args.layout, batches=batches,
                                                                 end_limit=args.end_limit, streams=streams)
                states = LongTermPlanner.stateAtSharded(planners, batches, total, rec_spec[1], streams=streams, outs=state_sets[r & 1])
                cur = [(inputs[g][0], *states[g]) for g in range(k)]
            return
        batches = LongTermPlanner.planSwitchTimesSharded(planners, inputs, total, layout=args.layout, batches=batches,
                                                         end_limit=args.switch_only and args.end_limit, streams=streams)
        if env_spec:
            envs = LongTermPlanner.envelopeSharded(planners, batches, total, env_spec[0], env_spec[1], outs=envs, streams=streams)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    LongTermPlanner.synchronizeSharded(planners, batches, streams)
    sync_all()
    elapsed = time.perf_counter() - t0

    ok = lens = 0
    M, acc = 1 << 48, 0
    for b in batches:
        ok += int(((b.status & ~256) == 0).sum().item())
        lens += int(b.traj_len.long().sum().item())
        if args.checksum:
            for x in (b.t_opt, b.t_scaled, b.dir, b.v_drive, b.t_required):
                acc += int(x.contiguous().view(torch.int64).sum().item()) % M
            for x in (b.mod, b.slowest, b.traj_len, b.status):
                acc += int(x.to(torch.int64).sum().item()) % M
    replans = rec_spec[0] if rec_spec else 1
    what = (f"{rec_spec[0]} receding-horizon cycles per step on the device (plan, state at sample {rec_spec[1]}: ltp_state_at_multi); value counts replans"
            if rec_spec else f"on-device envelope consumer: [min q, max q] over {env_spec[1]} windows of {env_spec[0]} samples per joint (ltp_envelope_multi)"
            if env_spec else "switching times only (ltp_plan_switch_times_multi" + (" + end-limit check" if args.end_limit else "; status = pre-sampling verdict") + ")")
    out = {
        "metric": "7-DoF trajectory plans/sec (batch 1M)" if dof == 7 else f"{dof}-DoF trajectory plans/sec",
        "value": round(total * args.steps * replans / elapsed, 1), "unit": "plans/s", "n_gpus": len(set(devices)), "n_shards": k,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{total} x {dof}-DoF queries per step in {k} device-resident shard(s) ({counts[0]} in shard 0), limits '{args.limits}', "
                        f"Tsample {args.t_sample} s, {what}, no rows",
            "batch_per_gpu": None if args.global_batch else args.batch, "global_batch": total, "dof": dof, "t_sample": args.t_sample,
            "limits": args.limits, "input_layout": args.layout, "devices": devices, "backend": "none (one process, no torch.distributed)",
            "sharding": "one process: one handle, stream and host thread per device, contiguous query ranges, device-resident shards, no collective",
            "plans_ok_frac": round(ok / total, 5),
            "plans_ok_is": "planTrajectory's bool" if (args.end_limit or env_spec) else "the pre-sampling verdict (cc:14-39); the end-limit check cc:59-61 was not run",
            "mean_traj_len": round(lens / total, 1),
            "bytes_per_plan": round(32.0 * dof * lens / total, 1),
        },
    }
    if args.checksum:
        out["config"]["records_checksum"] = acc % M
    return out

HEADLINE_BUDGET = 4096      # bytes: the driver keeps the tail of stdout only, so the line it parses must be short and LAST


def assemble_lines(out, secondary, rccl_world1, secondary_file):
    """(headline, side_lines, notes): the ONE JSON line stdout carries, the full secondary records for `secondary_file` (JSON lines), and
    one short plain-text note per secondary workload for stderr. Round 5's default line carried 18 secondary workloads inline and grew to
    29 KB; the driver's capture kept its tail only and parsed nothing. The headline is held under HEADLINE_BUDGET bytes whatever the run adds."""
    head = json.loads(json.dumps(out))      # deep copy: the caller's record is not trimmed
    side, notes = [], []
    for s in secondary or []:
        side.append(dict(s, kind="secondary"))
        r = s.get("roofline") or {}
        key = (s.get("config") or {}).get("workload_key", "?")
        notes.append("bench.py secondary: %s | %s %s | %s ms/step | %s %s" % (
            key, s.get("value"), s.get("unit", ""), s.get("ms_per_step"), r.get("bound", "-"),
            ("frac %s (%s)" % (r.get("frac"), r.get("kernel"))) if r.get("frac") is not None else (s.get("error") or "")))
    if rccl_world1 is not None:
        side.append(dict(rccl_world1, kind="rccl_world1"))
        head["rccl_world1"] = {k: rccl_world1[k] for k in ("ok", "value", "unit", "backend", "rank_devices", "roofline_frac", "error") if k in rccl_world1}
    if side:
        head["secondary"] = {"count": len(secondary or []), "errors": len([s for s in secondary or [] if "error" in s]), "file": secondary_file,
                             "note": "one JSON line per secondary workload in `file` (and one short text line each on stderr); this line is the headline only"}
    line = json.dumps(head)
    if len(line) > HEADLINE_BUDGET:       # drop prose first, never the numbers
        for path in (("roofline", "traffic_from_profile"), ("cpu_baseline", "flat_preallocated"), ("cpu_baseline", "cores_rule"), ("rccl_world1", "backend"),
                     ("secondary", "note"), ("config", "sharding"), ("config", "plans_ok_is"), ("cpu_baseline", "measured_on")):
            d = head.get(path[0])
            if isinstance(d, dict) and path[1] in d:
                del d[path[1]]
                line = json.dumps(head)
                if len(line) <= HEADLINE_BUDGET:
                    break
    return line, [json.dumps(x) for x in side], notes


def emit_result(out, secondary, rccl_world1, secondary_file):
    line, side, notes = assemble_lines(out, secondary, rccl_world1, secondary_file)
    if side:
        try:
            d = os.path.dirname(secondary_file)
            if d:
                os.makedirs(d, exist_ok=True)
            with open(secondary_file, "w") as f:
                f.write("\n".join(side) + "\n")
        except OSError as e:
            print(f"bench.py: could not write {secondary_file}: {e}", file=sys.stderr)
    for n in notes:
        print(n, file=sys.stderr)
    sys.stderr.flush()
    print(line, flush=True)


def main():
    args = parse_args()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.in_flight < 1 or (args.in_flight > 1 and (not args.switch_only or args.receding)):
        print("bench.py: --in-flight K (K >= 1) interleaves whole switching-times batches: it needs --switch-only", file=sys.stderr)
        return 2
    if args.one_process:
        out = run_one_process(args)
        print(json.dumps(out), flush=True)
        return 0
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args)            # no torch, no GPU in this process
    grouped = env_world is not None or args.force_dist     # a launcher's rank (also a lone one) always makes its process group
    if env_world is None and args.force_dist:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(RANK="0", LOCAL_RANK=str(args.device or 0), WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        # never report a line for a different number of GPUs than was asked for
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE {world} rank(s)", file=sys.stderr)
        return 2

    variant = (args.no_walk or args.walk or args.no_auto_waves or args.no_row_verdict or args.switch_only or args.end_limit or args.f32 or args.max_samples or args.sample_stride > 1 or args.envelope or args.receding or args.dry_sampler
               or args.in_flight > 1 or args.semantics != "cpp" or args.pow_rule != "libm" or args.table_pass != "auto" or args.limits != "panda" or args.batch != 1_000_000 or args.global_batch or args.window_gib or args.layout != "query_major")
    rccl_world1 = None
    if not grouped and not variant and not args.no_secondary and not args.no_rccl_check and args.backend == "nccl":
        rccl_world1 = rccl_self_check(args)     # a child process, BEFORE this one imports torch or touches the GPU

    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is not None:
        local_rank = args.device
    if local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants HIP device {local_rank}, but only {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        return 3
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":      # "nccl" is RCCL on ROCm
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where collective tensors live

    # one big reused output tile for every workload of this process; if this GPU cannot give 192 GiB right now
    # (e.g. several rehearsal ranks share it), halve until it can
    tile_state = {}

    def get_tile():
        if "t" not in tile_state:
            gib = args.tile_gib
            share = world if args.device is not None else 1       # rehearsal: N ranks on one device share its memory
            gib = gib / share
            while True:
                try:
                    tile_state["t"] = torch.empty(int(gib * (1 << 30)) // 4096 * 4096, dtype=torch.uint8, device=dev)
                    break
                except torch.OutOfMemoryError:
                    if gib <= 1:
                        raise
                    gib /= 2
                    torch.cuda.empty_cache()
            tile_state["gib"] = gib
        return tile_state["t"], tile_state["gib"]

    # which HIP device every rank drives, and what carried the barrier / reductions: stated in every line
    rank_devices = [local_rank]
    if grouped:
        gathered = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([local_rank], dtype=torch.int64, device=cdev))
        rank_devices = [int(x.item()) for x in gathered]
    backend = "none (single rank)" if not grouped else ("nccl (RCCL over xGMI)" if args.backend == "nccl" else "gloo (CPU tensors: rehearsal)")
    if grouped and world == 1:
        backend += "; world size 1: every collective of the N > 1 path ran through the process group"
    ctx = {"dist": grouped, "world": world, "rank": rank, "dev": dev, "cdev": cdev, "local_rank": local_rank, "tile": get_tile, "backend": backend,
           "rank_devices": rank_devices}
    primary = Workload(args)
    out = run_workload(primary, ctx)

    secondary = []
    if not args.no_secondary and not variant:
        few = max(1, min(args.steps, 2))
        if world == 1:
            plans = [
                ("S-ref: the reference's own limits (README.md:129-131), 1 M x 7-DoF, full sampling",
                 dict(limits="ref", steps=few, warmup=1)),
                ("configs[1]: 100 k x 7-DoF, switching times only",
                 dict(batch=100_000, switch_only=True, steps=max(args.steps, 20), warmup=2)),
                ("configs[1] + planTrajectory's end-limit verdict without sampling (ltp_end_limit_batch)",
                 dict(batch=100_000, switch_only=True, end_limit=True, steps=max(args.steps, 20), warmup=2)),
                ("configs[1] as throughput: two batches in flight (two handles, two streams; a step is still one whole batch)",
                 dict(batch=100_000, switch_only=True, in_flight=2, steps=max(args.steps, 40), warmup=4)),
                ("configs[1] with the reference's limits",
                 dict(limits="ref", batch=100_000, switch_only=True, steps=max(args.steps, 20), warmup=2)),
                ("configs[1] under the opt-in pow rule LTP_POW_EXACT (correctly rounded powers instead of glibc's pow restated: within 1 ulp of any libm)",
                 dict(batch=100_000, switch_only=True, pow_rule="exact", steps=max(args.steps, 20), warmup=2)),
                ("switching times only, 1 M x 7-DoF",
                 dict(switch_only=True, steps=max(args.steps, 10), warmup=2)),
                ("switching times only, 1 M x 7-DoF, two batches in flight",
                 dict(switch_only=True, in_flight=2, steps=max(args.steps, 20), warmup=4)),
                ("switching times only, 1 M x 7-DoF, under the opt-in pow rule LTP_POW_EXACT",
                 dict(switch_only=True, pow_rule="exact", steps=max(args.steps, 10), warmup=2)),
                ("configs[4]: 1 M x 30-DoF (S-ref30), full sampling through the reused tile",
                 dict(limits="ref30", steps=few, warmup=1)),
                ("first 256 samples of every row (SURVEY §8(f).2): k_sample_walk, run tables kept in the compute unit",
                 dict(max_samples=256, steps=max(args.steps, 5), warmup=1)),
                ("first 32 samples of every row: k_sample_walk_auto (caps of at most 32 samples: every wave builds AND writes its own batches; issue-bound, DESIGN.md §4 'Short rows')",
                 dict(max_samples=32, steps=max(args.steps, 5), warmup=1)),
                ("receding horizon, 10 cycles per step, 64-sample rows WRITTEN every cycle (from the second cycle on a third of the random restarts are rejected: batches are made of the live plans)",
                 dict(receding="10:100", max_samples=64, steps=few, warmup=1)),
                ("receding horizon (SURVEY §8(f).1): 10 cycles per step, 128-sample rows WRITTEN every cycle, restart states = stored sample 100 RECOMPUTED from the records (the rows are not read back)",
                 dict(receding="10:100", max_samples=128, steps=few, warmup=1)),
                ("every 4th sample of every row (SURVEY §8(f).2, strided rows): k_sample_walk, long-row form",
                 dict(sample_stride=4, steps=max(args.steps, 5), warmup=1)),
                ("float32 rows (SURVEY §8(f).2): the same binary64 results rounded once, k_sample_walk in sample pairs",
                 dict(f32=True, steps=few, warmup=1)),
                ("on-device envelope consumer (SURVEY §8(f).2): [min q, max q] over 32 windows of 64 samples per joint, every sample evaluated (bit-identical to the reduced rows)",
                 dict(envelope="64:32", steps=max(args.steps, 5), warmup=1)),
                ("the same envelopes, analytic form (ltp_set_envelope_mode: per run and window the end samples and the samples at the roots of q'(m) instead of every sample; k_envelope_walk: lane = (plan, joint), no run tables)",
                 dict(envelope="64:32", envelope_analytic=True, steps=max(args.steps, 5), warmup=1)),
            ]
        else:
            plans = [(f"configs[3]: 10 M x 7-DoF queries sharded over {world} GPUs, full sampling",
                      dict(global_batch=10_000_000, steps=few, warmup=1))]
        for name, over in plans:
            try:
                o = run_workload(Workload(args, name=name, **over), ctx)
            except Exception as e:      # a secondary workload must not take the headline down
                o = {"error": f"{type(e).__name__}: {e}"} if rank == 0 else None
                if grouped:
                    raise
            if o is not None:
                keep = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "roofline") if k in o}
                if "config" in o:
                    keep["config"] = o["config"]
                if "error" in o:
                    keep["error"] = o["error"]
                keep["name"] = name
                secondary.append(keep)

    if rank == 0:
        if not args.no_cpu_baseline:
            # rank 0 only, outside every timed region; with N > 1 the other ranks wait in the final barrier meanwhile
            from longtermplanner_amd import limit_set
            dof, lim = limit_set(args.limits)
            out["cpu_baseline"] = cpu_baseline(dof, lim, args.t_sample, args.seed, args.switch_only)
            if world > 1:
                out["cpu_baseline"]["measured_on"] = f"rank 0 of {world}, after the timed steps, while the other ranks wait in the final barrier"
        emit_result(out, secondary, rccl_world1, args.secondary_file)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

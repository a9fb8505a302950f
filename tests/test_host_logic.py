"""Host-side logic that needs no GPU: synthetic batches, sharding, chunking, packed layout."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_is_counter_based_and_valid(oracle_mod):
    from longtermplanner_amd import generate_queries, limit_set
    for name in ("panda", "ref", "ref30"):
        D, lim = limit_set(name)
        full = generate_queries(3000, lim, seed=5)
        # any shard reproduces its slice of the full batch bit for bit
        part = generate_queries(1000, lim, seed=5, first_query=1500)
        for f, p in zip(full, part):
            assert np.array_equal(f[1500:2500], p)
        other = generate_queries(3000, lim, seed=6)
        assert not np.array_equal(full[0], other[0])
        # every query passes the reference's checkInputs (cc:68-77), as tests/randomConfiguration.m intends
        orc = oracle_mod.Oracle(D, 0.001, **lim)
        qg, q0, v0, a0 = full
        assert all(orc.check_inputs(q0[i], v0[i], a0[i]) for i in range(0, 3000, 7))
        for arr, lo, hi in ((q0, lim["q_min"], lim["q_max"]), (qg, lim["q_min"], lim["q_max"])):
            assert np.all(arr >= np.array(lo)) and np.all(arr <= np.array(hi))


def test_shard_ranges_partition_the_batch():
    from longtermplanner_amd.parallel import shard_counts, shard_range
    for n in (0, 1, 7, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                first, count = shard_range(n, r, world)
                cover.append((first, count))
            assert cover[0][0] == 0 and sum(c for _, c in cover) == n
            for (f0, c0), (f1, _) in zip(cover, cover[1:]):
                assert f0 + c0 == f1
            assert max(shard_counts(n, world)) - min(shard_counts(n, world)) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_chunk_bounds_cover_every_plan_once():
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 50, size=1000).astype(np.uint64) * 16
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    for cap in (800, 5000, int(offsets[-1])):
        bounds = bench.chunk_bounds(offsets, cap)
        assert bounds[0][0] == 0 and bounds[-1][1] == 1000
        for (f0, e0), (f1, _) in zip(bounds, bounds[1:]):
            assert e0 == f1
        assert all(offsets[e] - offsets[f] <= cap for f, e in bounds)
    with pytest.raises(RuntimeError):
        bench.chunk_bounds(offsets, 8)


def test_unpack_trajectory_layout():
    from longtermplanner_amd import unpack_trajectory
    dof, length, stride = 3, 20, 32   # ltp_row_stride(20)
    packed = np.arange(1000, dtype=np.float64)
    q, v, a, j = unpack_trajectory(packed, 100, dof, length)
    assert q.shape == (dof, length)
    assert q[0, 0] == 100 and q[1, 0] == 100 + stride and v[0, 0] == 100 + dof * stride
    assert j[2, 19] == 100 + (3 * dof + 2) * stride + 19


def test_shard_range_is_the_same_rule_everywhere():
    """bench.py, parallel.py and the C ABI (ltp_shard_range, used by ltp_plan_batch_multi) must cut a batch identically."""
    import ctypes as C
    import bench
    from longtermplanner_amd import _abi
    from longtermplanner_amd.parallel import shard_range
    lib = _abi.lib()
    f, c = C.c_longlong(), C.c_longlong()
    for n in (0, 1, 5, 101, 10_000_000):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lib.ltp_shard_range(n, r, world, C.byref(f), C.byref(c))
                assert (f.value, c.value) == shard_range(n, r, world) == bench.shard_range(n, r, world)


def _run_bench(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)


def test_bench_refuses_a_mismatched_launch():
    # `--gpus 8` must never print a line for fewer ranks. A launcher that set WORLD_SIZE=1: exit code, no JSON.
    p = _run_bench(["--gpus", "8", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "{" not in p.stdout and "WORLD_SIZE 1" in p.stderr


def test_bench_spawns_ranks_itself_and_propagates_their_failure():
    # plain `python bench.py --gpus 2`: two rank processes are started (before torch / the GPU is touched in the parent). Here
    # there is no GPU, so each rank stops with "device not visible"; the parent must report that, not an n_gpus: 1 line.
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box runs this for real in tests/test_gpu_multi.py")
    p = _run_bench(["--gpus", "2", "--steps", "1", "--batch", "1000", "--no-cpu-baseline"])
    assert p.returncode != 0 and "{" not in p.stdout
    assert "rank 1 wants HIP device 1" in p.stderr and "exited with code" in p.stderr
    # nccl with every rank pinned to one device is refused up front
    p = _run_bench(["--gpus", "2", "--device", "0", "--steps", "1"])
    assert p.returncode != 0 and "gloo" in p.stderr


def test_bench_in_flight_needs_the_switching_times_workload():
    # batches in flight interleave whole switching-times calls on several handles; with rows the tile has one user
    p = _run_bench(["--in-flight", "2", "--steps", "1"])
    assert p.returncode != 0 and "{" not in p.stdout and "--switch-only" in p.stderr
    p = _run_bench(["--in-flight", "0", "--switch-only"])
    assert p.returncode != 0 and "{" not in p.stdout


def test_dense_comparator_of_the_soak_tool(oracle_mod):
    """oracle/ltp_oracle.c: ltpo_compare_dense (tools/dense_soak.py) — rows packed from the oracle's own trajectories compare
    equal; a perturbed sample, a wrong length and a wrong verdict are each reported."""
    import ctypes as C
    from longtermplanner_amd.synthetic import generate_queries, limit_set
    D, lim = limit_set("ref")
    Ts = 0.004
    orc = oracle_mod.Oracle(D, Ts, **lim)
    n = 12
    qs = [np.ascontiguousarray(x) for x in generate_queries(n, lim, seed=5)]
    qs[1][3, 0] = 9.0                                        # rejected by checkInputs
    lens, status, blocks, off = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), [], [0]
    for p in range(n):
        r = orc.plan_trajectory(*[x[p] for x in qs])
        if r["status"] == 0:
            status[p] = 1
            off.append(off[-1])
            continue
        L = r["length"]
        lens[p] = L
        status[p] = 8 if r["status"] == 2 else 0
        stride = (L + 31) // 32 * 32
        blk = np.zeros((4, D, stride))
        for x, key in enumerate("qvaj"):
            blk[x, :, :L] = r[key]
        blocks.append(blk.reshape(-1))
        off.append(off[-1] + blk.size)
    packed = np.concatenate(blocks)
    off = np.asarray(off, dtype=np.uint64)
    lib = oracle_mod.lib()
    lib.ltpo_compare_dense.restype = C.c_longlong
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)

    def run():
        maxd, flag = np.zeros((n, 4)), np.zeros(n, dtype=np.int32)
        cnt = lib.ltpo_compare_dense(orc._ref, C.c_long(0), C.c_long(n), *[x.ctypes.data_as(dp) for x in qs], packed.ctypes.data_as(dp),
                                     off.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_ulonglong(0), lens.ctypes.data_as(ip),
                                     status.ctypes.data_as(ip), maxd.ctypes.data_as(dp), flag.ctypes.data_as(ip))
        assert np.all(flag & 16)                            # every plan visited
        return cnt, maxd, flag & ~16
    cnt, maxd, flag = run()
    assert cnt == 4 * D * int(lens.sum()) and not maxd.any() and not flag.any()
    packed[int(off[5]) + 7] += 1e-6                          # q of joint 0, sample 7 of plan 5
    packed[int(off[8]) + 3 * D * ((int(lens[8]) + 31) // 32 * 32) + 2] += 1e-13      # a jerk sample of plan 8, below any tolerance
    status[6] ^= 8
    lens[7] += 1
    status[3] = 0                                            # the device "planned" a query the oracle rejects
    cnt, maxd, flag = run()
    assert abs(maxd[5, 0] - 1e-6) < 1e-12 and not maxd[5, 1:].any() and flag[5] == 0 and flag[6] == 4 and flag[7] == 2 and flag[3] == 1 and flag[8] == 8
    assert np.count_nonzero(maxd) == 2 and np.count_nonzero(flag) == 4


def test_bench_workload_keys_and_committed_counters():
    """bench.py names every workload (config.workload_key) and quotes the committed PMC passes of profiles/bench_counters.json under that
    name, with their provenance — never as measured in the run. Pure host logic: the keys of the lines the default run prints exist in
    the committed file where a pass was made, the entries have the fields the roofline blocks read, and unknown workloads get nothing."""
    import json
    import types
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args([])
    wl = bench.Workload(args)
    assert bench.workload_key(wl, 1_000_000) == "panda:1000000:f64"
    cases = [(dict(batch=100_000, switch_only=True), 100_000, "panda:100000:f64:switch_only"),
             (dict(batch=100_000, switch_only=True, end_limit=True), 100_000, "panda:100000:f64:switch_only+end_limit"),
             (dict(batch=100_000, switch_only=True, pow_rule="exact"), 100_000, "panda:100000:f64:switch_only:pow_exact"),
             (dict(switch_only=True, in_flight=2), 1_000_000, "panda:1000000:f64:switch_only:inflight2"),
             (dict(max_samples=256), 1_000_000, "panda:1000000:f64:first256"),
             (dict(sample_stride=4), 1_000_000, "panda:1000000:f64:stride4"),
             (dict(f32=True), 1_000_000, "panda:1000000:f32"),
             (dict(envelope="64:32"), 1_000_000, "panda:1000000:f64:envelope64:32"),
             (dict(envelope="64:32", envelope_analytic=True), 1_000_000, "panda:1000000:f64:envelope64:32+analytic"),
             (dict(receding="10:100", max_samples=128), 1_000_000, "panda:1000000:f64:first128:receding10:100"),
             (dict(limits="ref", semantics="matlab"), 1_000_000, "ref:1000000:f64:matlab")]
    for over, n, want in cases:
        assert bench.workload_key(bench.Workload(args, **over), n) == want, (over, bench.workload_key(bench.Workload(args, **over), n))
    d = json.load(open(os.path.join(ROOT, "profiles", "bench_counters.json")))
    assert "rocprofv3" in d["how"] and d["collected"]
    for key in ("panda:1000000:f64", "panda:1000000:f64:first256", "panda:1000000:f64:stride4", "panda:1000000:f32"):
        e = bench.committed_counters(key)
        assert e["write_bytes_per_launch"] > 1e10 and "k_sample" in e["sampler_kernel"] and "NOT measured in this run" in e["source"], key
    # the headline's HBM write bytes: within 0.2 % of the algorithmic 192.9 GB per launch
    assert abs(bench.committed_counters("panda:1000000:f64")["write_bytes_per_launch"] / 192906497168 - 1.0) < 2e-3
    for key in ("panda:100000:f64:switch_only", "panda:1000000:f64:switch_only", "panda:100000:f64:switch_only:pow_exact", "panda:1000000:f64:envelope64:32"):
        e = bench.committed_counters(key)
        assert 0.0 < e["valu_issue_frac"] < 1.0 and e["kernel_time_us"] > 0, key
        for name, k in e["kernels"].items():
            assert k["avg_us"] > 0 and k["valu_wave_insts"] > 0 and 0.0 <= k["valu_issue_frac"] <= 1.0, (key, name)
    assert bench.committed_counters("panda:123:f64:first7") is None

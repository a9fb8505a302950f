"""SURVEY.md §8(e) on the hardware a single-GPU box has: one batch cut into k virtual shards on device 0 must give the bits
of the unsharded run — through the one-process C ABI overload (ltp_plan_batch_multi, one handle + host thread per shard),
through the C++ class (planTrajectoryBatchSharded, tests/cpp/dropin_tests.cc) and through bench.py's one-process-per-rank
path (self-spawned ranks, gloo for the barrier, every rank on device 0)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("limits,n,k", [("panda", 1001, 3), ("ref", 777, 2), ("ref30", 130, 4), ("ref", 3, 5)])
def test_virtual_shards_are_bit_identical_to_the_unsharded_batch(limits, n, k):
    import longtermplanner_amd as amd
    D, lim = amd.limit_set(limits)
    one = amd.LongTermPlanner(D, 0.002, device=0, **lim)
    shards = [amd.LongTermPlanner(D, 0.002, device=0, **lim) for _ in range(k)]
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=31)
    if n > 100:
        q0[17, 0] = 99.0                                       # a rejected query inside a shard
    want = one.planBatchHost(qg, q0, v0, a0, sample=True)
    got = amd.LongTermPlanner.planBatchSharded(shards, qg, q0, v0, a0, sample=True)
    for key in want:
        assert want[key].tobytes() == got[key].tobytes(), key
    assert want["packed"].size > 0
    # switching times only: same records, status includes the end-limit verdict, nothing sampled
    got0 = amd.LongTermPlanner.planBatchSharded(shards, qg, q0, v0, a0, sample=False)
    for key in got0:
        assert want[key].tobytes() == got0[key].tobytes(), key
    # shards must be configured alike
    shards[-1].setSampleTime(0.004)
    with pytest.raises(amd.LtpError):
        amd.LongTermPlanner.planBatchSharded(shards, qg, q0, v0, a0)


def _bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "1", "--warmup", "1",
                        "--tile-gib", "8", *args], capture_output=True, text=True, timeout=600, env=env)
    return p


def test_bench_spawns_its_ranks_and_shards_one_global_batch():
    """`python bench.py --gpus 2` run plainly starts two rank processes (never an n_gpus: 1 line), and one global batch cut
    into 1, 2 and 3 contiguous shards has the same record checksum, ok-count and bytes."""
    lines = {}
    for world in (1, 2, 3):
        p = _bench("--gpus", str(world), "--backend", "gloo", "--device", "0", "--global-batch", "30001", "--limits", "ref", "--checksum")
        assert p.returncode == 0, p.stderr[-2000:]
        out = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(out) == 1, p.stdout
        lines[world] = json.loads(out[0])
        assert lines[world]["n_gpus"] == world and lines[world]["scaling"] == "strong"
        assert lines[world]["config"]["global_batch"] == 30001
    for world in (2, 3):
        for key in ("records_checksum", "plans_ok_frac", "mean_traj_len", "bytes_per_plan"):
            assert lines[world]["config"][key] == lines[1]["config"][key], (world, key)
    # weak scaling (the default): N ranks x --batch queries
    p = _bench("--gpus", "2", "--backend", "gloo", "--device", "0", "--batch", "20000")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 40000


def test_two_batches_in_flight_give_the_same_records():
    """`bench.py --switch-only --in-flight 2` alternates its steps between two handles on two streams (the queue-B kernel of
    one step runs under the next step's stages): the records are those of one batch at a time."""
    sums = {}
    for k in (1, 2, 3):
        p = _bench("--switch-only", "--batch", "30001", "--limits", "ref", "--checksum", "--in-flight", str(k), "--steps", "5", "--end-limit")
        assert p.returncode == 0, p.stderr[-2000:]
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        assert line["config"]["batches_in_flight"] == k and line["steps"] == 5
        sums[k] = (line["config"]["records_checksum"], line["config"]["plans_ok_frac"], line["config"]["mean_traj_len"])
    assert sums[2] == sums[1] and sums[3] == sums[1]
    p = _bench("--in-flight", "2")                        # only the switching-times workload can be interleaved like this
    assert p.returncode != 0


def test_bench_never_reports_fewer_gpus_than_requested():
    # a launcher that started a different number of ranks: refuse (exit code, no JSON line)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0 and "{" not in p.stdout
    # more ranks than devices, nccl: every rank needs its own GPU -> non-zero exit, no line
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _bench("--gpus", "2", "--batch", "1000")
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]

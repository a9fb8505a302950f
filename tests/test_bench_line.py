"""bench.py's output contract, on canned numbers (no GPU): the LAST stdout line is the headline alone and short.

Round 5's default run put 18 secondary workloads inside the single result line (29 388 bytes, profiles/r05_bench_driver_like.json); the
driver keeps the tail of stdout only, so BENCH_r05.json's `parsed` was null and the round's headline went unrecorded. The canned input here
IS that record: its secondary list and rccl block must leave the line, the numbers the driver checks must stay."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_like.json")))
    secondary = rec.pop("secondary")
    rccl = rec.pop("rccl_world1")
    assert len(secondary) >= 16 and len(json.dumps(secondary)) > 20000
    return rec, secondary, rccl


def test_headline_is_short_and_carries_what_the_driver_reads(tmp_path):
    out, secondary, rccl = _canned()
    f = str(tmp_path / "sub" / "bench_secondary.jsonl")
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit_result(out, secondary, rccl, f)
    lines = so.getvalue().splitlines()
    assert len(lines) == 1, "stdout carries exactly one line"
    last = lines[-1]
    assert len(last) < bench.HEADLINE_BUDGET == 4096, len(last)
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["config"]["workload"].startswith("1000000 x 7-DoF") and d["config"]["workload_key"] == "panda:1000000:f64"
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "GB/s" and "traffic" in r
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["kind"] == "port" and c["cores"] >= 1 and c["sample"]
    assert d["secondary"] == {"count": len(secondary), "errors": 0, "file": f, "note": d["secondary"]["note"]}
    assert d["rccl_world1"]["ok"] is True and d["rccl_world1"]["rank_devices"] == [0]
    # the full records are in the file, one line each, nothing lost
    side = [json.loads(x) for x in open(f).read().splitlines()]
    assert [x["kind"] for x in side] == ["secondary"] * len(secondary) + ["rccl_world1"]
    for a, b in zip(side, secondary):
        assert {k: v for k, v in a.items() if k != "kind"} == b
    # stderr: one short text line per secondary workload, none of which a JSON-line parser would pick up
    notes = se.getvalue().splitlines()
    assert len(notes) == len(secondary) and all(not n.startswith("{") and len(n) < 250 for n in notes)
    assert len(se.getvalue()) < 4096


def test_headline_stays_under_budget_when_a_run_adds_prose():
    out, secondary, rccl = _canned()
    out["config"]["sharding"] = "x" * 3000
    out["cpu_baseline"]["measured_on"] = "y" * 500
    line, side, notes = bench.assemble_lines(out, secondary, rccl, "f.jsonl")
    d = json.loads(line)
    assert len(line) <= bench.HEADLINE_BUDGET and d["value"] == out["value"] and d["roofline"]["frac"] and d["cpu_baseline"]["value"]


def test_plain_variant_line_is_unchanged():
    out, _, _ = _canned()
    line, side, notes = bench.assemble_lines(out, [], None, "f.jsonl")
    assert side == [] and notes == [] and json.loads(line) == out


def test_multi_gpu_line_keeps_the_same_contract(tmp_path):
    """The N > 1 path (rank 0 of a launcher's ranks): one secondary workload (configs[3], 10 M queries sharded), no rccl_world1 child —
    the same single short line, the same file."""
    out, secondary, _ = _canned()
    out.update(n_gpus=8, value=8 * out["value"], scaling="weak")
    out["config"].update(global_batch=8_000_000, backend="nccl (RCCL over xGMI)", rank_devices=list(range(8)))
    out["cpu_baseline"]["measured_on"] = "rank 0 of 8, after the timed steps, while the other ranks wait in the final barrier"
    sec = [dict(secondary[0], name="configs[3]: 10 M x 7-DoF queries sharded over 8 GPUs, full sampling")]
    f = str(tmp_path / "bench_secondary.jsonl")
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit_result(out, sec, None, f)
    lines = so.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.HEADLINE_BUDGET
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["secondary"]["count"] == 1 and "rccl_world1" not in d and d["roofline"]["frac"] and d["cpu_baseline"]["cores"]
    assert [json.loads(x)["kind"] for x in open(f).read().splitlines()] == ["secondary"]

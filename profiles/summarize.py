#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries kept under profiles/.

usage: python profiles/summarize.py <round tag> <trace dir> [<pmc write dir> [<pmc fetch dir>]]
  trace dir : output of  rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
  pmc dirs  : output of  rocprofv3 --kernel-trace --pmc WRITE_SIZE|FETCH_SIZE --output-format csv -- python3 bench.py ...
WRITE_SIZE / FETCH_SIZE are reported by rocprofv3 in KiB. Per /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) WRITE_SIZE is exact for 16-B-per-lane streaming stores; FETCH_SIZE under-reports wide coalesced
reads by 2x on gfx950 (the sampler's reads are a few narrow record loads, so its FETCH figure is only
indicative).
"""
import collections
import csv
import glob
import json
import os
import sys


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    if not g:
        raise SystemExit(f"no file matches {pattern}")
    return max(g, key=os.path.getmtime)   # gpurun merges new runs into the same directory: take the latest


def short(name):
    return name.split("(")[0]


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    out = {"round": tag, "kernels": []}
    rows = list(csv.DictReader(open(one(os.path.join(trace, "**", "*_kernel_stats.csv")))))
    for r in rows:
        out["kernels"].append({"name": short(r["Name"]), "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                               "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"]),
                               "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3})
    for i, key in ((3, "WRITE_SIZE"), (4, "FETCH_SIZE")):
        if len(sys.argv) > i:
            f = one(os.path.join(sys.argv[i], "**", "*_counter_collection.csv"))
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == key:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            out[key + "_KiB_per_launch"] = {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in agg.items()}
    path = os.path.join(here, f"{tag}_rocprof_summary.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(one(os.path.join(trace, "**", "*_kernel_stats.csv"))).read())


if __name__ == "__main__":
    main()

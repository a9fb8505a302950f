#!/bin/bash
# After `gpurun -- bash tools/refresh_profiles.sh TAG` has merged its output into gpurun_out/: condense it into profiles/.
TAG=${1:-r06}
cd "$(dirname "$0")/.."
python profiles/summarize.py $TAG gpurun_out/prof_stats gpurun_out/prof_write gpurun_out/prof_fetch || exit 1
cp gpurun_out/bench_default.json profiles/${TAG}_bench.json
cp gpurun_out/bench_default_secondary.jsonl profiles/${TAG}_bench_secondary.jsonl
cp gpurun_out/bench_variants.jsonl profiles/${TAG}_bench_variants.jsonl
cp gpurun_out/bench_dry.jsonl profiles/${TAG}_dry_sampler_ceilings.jsonl
cp gpurun_out/bench_2ranks_gloo.json profiles/${TAG}_bench_2ranks_one_gpu_gloo.json
cp gpurun_out/bench_4ranks_gloo_global.json profiles/${TAG}_bench_4ranks_one_gpu_gloo_global_batch.json
cp gpurun_out/bench_config4_4ranks_gloo.json profiles/${TAG}_bench_config4_10M_4ranks_one_gpu_gloo.json
cp "$(find gpurun_out/prof_f32_stats -name '*_kernel_stats.csv' | xargs ls -t | head -1)" profiles/${TAG}_f32_rows_kernel_stats.csv
cp "$(find gpurun_out/prof_tab_stats -name '*_kernel_stats.csv' | xargs ls -t | head -1)" profiles/${TAG}_table_pass_first256_kernel_stats.csv
cp "$(find gpurun_out/prof_switch_100000 -name '*_kernel_stats.csv' | xargs ls -t | head -1)" profiles/${TAG}_switch_only_100k_kernel_stats.csv
cp "$(find gpurun_out/prof_switch_1000000 -name '*_kernel_stats.csv' | xargs ls -t | head -1)" profiles/${TAG}_switch_only_1M_kernel_stats.csv
python - <<PY
import csv, glob, collections, json
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary <workload>",
       "note": "per-launch means over all launches of the run (warm-up included); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)", "workloads": {}}
for d, name in (("prof_sq_first256", "--max-samples 256 (1 M plans, first 256 samples, k_sample_walk: tables kept in the compute unit)"),
                ("prof_sq_first256_tab", "--max-samples 256 --no-walk (the same through the table pass: k_build_tables + k_sample_tab)"),
                ("prof_sq_first64", "--max-samples 64 (k_sample_walk)"), ("prof_sq_first64_tab", "--max-samples 64 --no-walk (table pass)"), ("prof_sq_switch100k", "--switch-only --batch 100000 (config 2)"),
                ("prof_sq_f32", "--f32 (1 M plans, float32 rows)"), ("prof_sq_envelope", "--envelope 64:32 (1 M plans, envelope consumer)")):
    f = max(glob.glob(f"gpurun_out/{d}/**/*_counter_collection.csv", recursive=True), key=lambda x: __import__("os").path.getmtime(x))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = {}
    for k, c in agg.items():
        if "ltp::" not in k:
            continue
        e = {n: round(sum(v) / len(v), 1) for n, v in c.items()}
        e["launches"] = len(next(iter(c.values())))
        if e.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac"] = round(e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"], 3)
        w[k] = e
    out["workloads"][name] = w
json.dump(out, open("profiles/${TAG}_sq_counters.json", "w"), indent=1)
PY
python - <<PY
import json, csv
b = json.loads(open("profiles/${TAG}_bench.json").read())
print("headline", b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"])
for s in [json.loads(x) for x in open("profiles/${TAG}_bench_secondary.jsonl") if '"kind": "secondary"' in x]:
    print("  ", s["name"][:72], s.get("value"), s.get("ms_per_step"), (s.get("roofline") or {}).get("achieved"), s.get("error"))
for f in ["profiles/${TAG}_bench_variants.jsonl", "profiles/${TAG}_dry_sampler_ceilings.jsonl", "profiles/${TAG}_bench_2ranks_one_gpu_gloo.json",
          "profiles/${TAG}_bench_4ranks_one_gpu_gloo_global_batch.json", "profiles/${TAG}_bench_config4_10M_4ranks_one_gpu_gloo.json"]:
    print(f)
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        d = json.loads(ln); r = d.get("roofline") or {}; c = d["config"]
        print("  %-80s | %9.3f M | %8.3f ms | %s | tp=%s" % (c["workload"][48:128], d["value"] / 1e6, d["ms_per_step"], r.get("achieved"), c.get("table_pass")))
for r in list(csv.DictReader(open("profiles/${TAG}_table_pass_first256_kernel_stats.csv")))[:3]:
    print("  ", r["Name"][:30], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY

# profiles/bench_counters.json: the PMC figures bench.py's roofline blocks quote (keyed by config.workload_key), with how they were made
python - <<PY
import csv, glob, json, os, collections
G = "gpurun_out"
def latest(pattern):
    g = glob.glob(pattern, recursive=True)
    return max(g, key=os.path.getmtime) if g else None
def counters(d, want):
    f = latest(f"{G}/{d}/**/*_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    if f:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in acc.items() if want(k)}
def stats(d, want):
    f = latest(f"{G}/{d}/**/*_kernel_stats.csv")
    return {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f)) if want(r["Name"])} if f else {}
out = {"collected": "round ${TAG#r0}, tools/refresh_profiles.sh ${TAG} + tools/collect_profiles.sh ${TAG}",
       "how": "separate rocprofv3 passes of `python3 bench.py --no-secondary --no-cpu-baseline <workload>`: --pmc WRITE_SIZE (KiB, exact for 16-byte-per-lane stores per "
              "MI355X_MICROARCH.md) for the row writers; --pmc SQ_INSTS_VALU ... and --kernel-trace --stats for the stage / envelope kernels. valu_issue_frac = "
              "SQ_INSTS_VALU (wave instructions) x 4 issue cycles / (1024 SIMDs x 2.4 GHz x kernel time): the share of the chip's vector issue slots the kernel used",
       "workloads": {}}
def writer(key, d, kernel_part):
    c = counters(d, lambda k: kernel_part in k)
    for k, v in c.items():
        if "WRITE_SIZE" in v:
            out["workloads"][key] = {"sampler_kernel": k, "write_bytes_per_launch": int(v["WRITE_SIZE"] * 1024)}
writer("panda:1000000:f64", "prof_write", "k_sample")
writer("panda:1000000:f64:first256", "prof_write_first256", "k_sample")
writer("panda:1000000:f64:stride4", "prof_write_stride4", "k_sample")
writer("panda:1000000:f32", "prof_write_f32", "k_sample")
def stages(key, dsq, dstats, names=("k_opt_fast", "k_opt_slow", "k_reduce_scale", "k_scaling_slow", "k_envelope", "k_build_tables")):
    c = counters(dsq, lambda k: any(n in k for n in names))
    t = stats(dstats, lambda k: any(n in k for n in names))
    e, ins, tot = {}, 0.0, 0.0
    for k, v in c.items():
        if k in t and "SQ_INSTS_VALU" in v:
            frac = v["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9 * t[k] * 1e-6)
            e[k] = {"avg_us": round(t[k], 2), "valu_wave_insts": int(v["SQ_INSTS_VALU"]), "valu_issue_frac": round(frac, 3),
                    "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") else None}
            ins += v["SQ_INSTS_VALU"]; tot += t[k]
    if e:
        out["workloads"][key] = {"kernels": e, "valu_issue_frac": round(ins * 4.0 / (1024 * 2.4e9 * tot * 1e-6), 3), "kernel_time_us": round(tot, 2)}
stages("panda:100000:f64:switch_only", "prof_sq_switch100k", "prof_switch_100000")
stages("panda:1000000:f64:switch_only", "prof_sq_switch1M", "prof_switch_1000000")
stages("panda:100000:f64:switch_only:pow_exact", "prof_sq_switch100k_exact", "prof_switch_100000_exact")
stages("panda:1000000:f64:envelope64:32", "prof_sq_envelope", "prof_env_stats")
json.dump(out, open("profiles/bench_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY

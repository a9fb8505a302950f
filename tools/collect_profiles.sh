#!/bin/bash
# After `gpurun -- bash tools/refresh_profiles.sh TAG` has merged its output into gpurun_out/: condense it into profiles/.
TAG=${1:-r04}
cd "$(dirname "$0")/.."
python profiles/summarize.py $TAG gpurun_out/prof_stats gpurun_out/prof_write gpurun_out/prof_fetch || exit 1
cp gpurun_out/bench_default.json profiles/${TAG}_bench.json
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
for s in b.get("secondary", []):
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

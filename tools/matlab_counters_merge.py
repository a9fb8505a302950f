"""Adds the MATLAB-semantics 100 k switching-times entry to profiles/bench_counters.json from the passes of tools/matlab_counters.sh
(same arithmetic as tools/collect_profiles.sh's stages())."""
import collections, csv, glob, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
names = ("k_opt_fast", "k_opt_slow", "k_reduce_scale", "k_scaling_slow")
f = max(glob.glob(f"{G}/prof_sq_switch100k_matlab/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
c = {k: {n: sum(v.values()) / len(v) for n, v in cs.items()} for k, cs in acc.items() if any(x in k for x in names)}
f = max(glob.glob(f"{G}/prof_switch_100000_matlab/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
t = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f)) if any(x in r["Name"] for x in names)}
e, ins, tot = {}, 0.0, 0.0
for k, v in c.items():
    if k in t and "SQ_INSTS_VALU" in v:
        e[k] = {"avg_us": round(t[k], 2), "valu_wave_insts": int(v["SQ_INSTS_VALU"]), "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9 * t[k] * 1e-6), 3),
                "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") else None}
        ins += v["SQ_INSTS_VALU"]; tot += t[k]
p = os.path.join(R, "profiles", "bench_counters.json")
d = json.load(open(p))
d["workloads"]["panda:100000:f64:switch_only:matlab"] = {"kernels": e, "valu_issue_frac": round(ins * 4.0 / (1024 * 2.4e9 * tot * 1e-6), 3), "kernel_time_us": round(tot, 2),
                                                        "collected": "round 5, tools/matlab_counters.sh + tools/matlab_counters_merge.py (the build with the register-resident roots())"}
json.dump(d, open(p, "w"), indent=1)
print(json.dumps(d["workloads"]["panda:100000:f64:switch_only:matlab"], indent=1))

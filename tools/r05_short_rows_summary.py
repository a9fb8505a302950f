"""Summarises gpurun_out/r05_short_rows/<tag>/{stats,SQ,SQ2,TCC,TCC2,TCP}: per workload the sampler kernel's average time and the
per-launch mean of every counter (summed over the dimension instances rocprofv3 reports), as one JSON."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
out = {}
for tag in sorted(os.listdir(root)):
    d = os.path.join(root, tag)
    if not os.path.isdir(d):
        continue
    rec = {}
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Name"].split("(")[0].replace("void ", "")
            if "k_sample" in name:
                rec["kernel"] = name
                rec["avg_us"] = float(row["AverageNs"]) / 1e3
                rec["calls"] = int(row["Calls"])
    counters = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_sample" not in row["Kernel_Name"]:
                continue
            counters[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    rec["counters_per_launch"] = {c: sum(v.values()) / len(v) for c, v in sorted(counters.items())}
    for f in glob.glob(os.path.join(d, "stats.log")):
        for line in open(f):
            if line.startswith("{"):
                o = json.loads(line)
                rec["bench"] = {"value": o["value"], "ms_per_step": o["ms_per_step"], "roofline": o.get("roofline"), "bytes_per_plan": o["config"]["bytes_per_plan"]}
    out[tag] = rec
print(json.dumps(out, indent=1))

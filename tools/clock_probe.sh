#!/bin/bash
# One box: the bench line of a workload (roofline fraction from HIP events) and, in a pass of its own, the chip's effective clock during
# its dominant kernel — GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / kernel time (MI355X_MICROARCH.md, "DVFS give-back") — to see
# whether a box-to-box spread of one kernel is a spread of clocks. usage: bash tools/clock_probe.sh "<bench args>" <kernel substring>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline --no-secondary --no-rccl-check --steps 10 --warmup 3 $1 > $O/clock_line.json 2>/dev/null || exit 1
rm -rf $O/clock_pmc
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clock_pmc -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-rccl-check --steps 10 --warmup 3 $1 > $O/clock_pmc.log 2>&1 || { tail -3 $O/clock_pmc.log; exit 1; }
python3 - "$2" <<PY
import csv, glob, json, sys, collections
O = "$O"
d = json.loads(open(O + "/clock_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
f = glob.glob(O + "/clock_pmc/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(float); dur = {}
for row in csv.DictReader(open(f)):
    if sys.argv[1] in row["Kernel_Name"] and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
        k = row["Dispatch_Id"]
        tot[k] += float(row["Counter_Value"])
        dur[k] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
clk = sorted(tot[k] / 8.0 / dur[k] for k in tot if dur[k] > 0)
agents = open(glob.glob(O + "/clock_pmc/**/*agent_info.csv", recursive=True)[0]).read()
print(json.dumps({"workload": d["config"]["workload_key"], "frac": r.get("frac"), "avg_launch_ms": r.get("avg_launch_ms"), "kernel": r.get("kernel"),
                  "dispatches": len(clk), "effective_clock_GHz_median": round(clk[len(clk) // 2], 3) if clk else None,
                  "effective_clock_GHz_min_max": [round(clk[0], 3), round(clk[-1], 3)] if clk else None,
                  "profiled_launch_ms_median": round(sorted(dur.values())[len(dur) // 2] / 1e6, 3) if dur else None}))
PY

#!/bin/bash
# One box: roofline fraction of a workload and, in a pass of its own, the vector L1's address-translation counters of its dominant kernel
# (UTCL1 requests / hits / misses, cycles with translations pending) — is a box-to-box spread of one kernel a spread of TLB behaviour
# (the physical placement of the output tile differs from box to box)? usage: bash tools/tlb_probe.sh "<bench args>" <kernel substring>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline --no-secondary --no-rccl-check --steps 10 --warmup 3 $1 > $O/tlb_line.json 2>/dev/null || exit 1
rm -rf $O/tlb_pmc
timeout -k 10 90 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT --output-format csv -d $O/tlb_pmc -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-rccl-check --steps 5 --warmup 2 $1 > $O/tlb_pmc.log 2>&1 || { tail -3 $O/tlb_pmc.log; exit 1; }
python3 - "$2" <<PY
import csv, glob, json, sys, collections
O = "$O"
d = json.loads(open(O + "/tlb_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
f = glob.glob(O + "/tlb_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f)):
    if sys.argv[1] in row["Kernel_Name"]:
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
out = {c: round(sum(v.values()) / len(v)) for c, v in acc.items()}
print(json.dumps({"workload": d["config"]["workload_key"], "frac": r.get("frac"), "avg_launch_ms": r.get("avg_launch_ms"), "per_launch": out}))
PY

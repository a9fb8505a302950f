#!/bin/bash
# One TCC/EA pass (4 counters: more "exceeds the capabilities of the hardware" in one pass) per workload: the write-request mix and the
# EA write stalls of the row writers. usage: bash tools/r05_tcc_pass.sh "bench args" ...   -> gpurun_out/r05_tcc/<n>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_tcc
mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1)); rm -rf $O/$i; mkdir -p $O/$i; echo "$v" > $O/$i/args.txt
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL --output-format csv -d $O/$i/tcc -- python3 $R/bench.py $v --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-check > $O/$i/tcc.log 2>&1 || { echo "$v failed"; tail -3 $O/$i/tcc.log; }
  echo "done $v"
done
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$O/*/")):
    args = open(os.path.join(d, "args.txt")).read().strip()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "tcc", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sample" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    e = {}
    for (k, c), v in acc.items():
        e.setdefault(k, {})[c] = sum(v.values()) / len(v)
    for line in open(os.path.join(d, "tcc.log")):
        if line.startswith("{"):
            o = json.loads(line); e["bench"] = {"roofline": o.get("roofline"), "bytes_per_plan": o["config"]["bytes_per_plan"]}
    out[args] = e
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
PY

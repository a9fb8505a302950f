#!/bin/bash
# SQ counters of the single-call kernel (k_plan_small) under tools/latency_probe.py: where do its ~20 us go?
# usage (on the GPU box): bash tools/small_pmc.sh  -> gpurun_out/small_pmc/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/small_pmc
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  tag=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/$tag -- python3 $R/tools/latency_probe.py > $O/$tag.log 2>&1 || { echo "set $SET failed"; tail -3 $O/$tag.log; }
  # config 2 of BASELINE.json (1000 plans, switching times only): the stage kernels
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/c2_$tag -- python3 $R/bench.py --switch-only --batch 1000 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/c2_$tag.log 2>&1 || { echo "config 2 set $SET failed"; tail -3 $O/c2_$tag.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_plan_small" in k or "k_opt" in k or "k_scaling" in k or "k_reduce" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} mean {sum(v)/len(v):12.1f}  n={len(v)}")
PY

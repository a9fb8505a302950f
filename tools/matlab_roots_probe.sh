#!/bin/bash
# gpurun -- bash tools/matlab_roots_probe.sh : durations and instruction counts of tools/matlab_roots_probe.py's launches
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/probe_trace -- python3 $R/tools/matlab_roots_probe.py > $O/probe.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/probe_pmc -- python3 $R/tools/matlab_roots_probe.py > $O/probe_pmc.log 2>&1 || exit 1
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" "SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAIT_ANY SQ_IFETCH"; do
n=$(echo $set | cut -d" " -f1)
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/probe_pmc_$n -- python3 $R/tools/matlab_roots_probe.py > $O/probe_pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, glob
t = sorted(glob.glob("$O/probe_trace/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(t)) if "k_roots_matlab" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
pm = None
for c in sorted(glob.glob("$O/probe_pmc*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for r in csv.DictReader(open(c)):
        if "k_roots_matlab" not in r["Kernel_Name"]: continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    one = [per[k] for k in sorted(per)]
    if pm is None: pm = one
    else:
        for a, b in zip(pm, one): a.update(b)
names = ["warm-up"] + [f"degree {d}: {w}" for d in (4, 5, 6) for w in ("worst x 64 lanes", "worst x 1 lane", "all of the batch")]
with open("$O/matlab_roots_probe.txt", "w") as f:
    f.write(open("$O/probe.log").read())
    for i, n in enumerate(names):
        p = pm[i] if i < len(pm) else {}
        f.write(f"{n:32s} {dur[i]:9.2f} us  " + "  ".join(f"{k[3:] if k.startswith('SQ_') else k} {v:.0f}" for k, v in sorted(p.items())) + "\n")
print(open("$O/matlab_roots_probe.txt").read())
PY

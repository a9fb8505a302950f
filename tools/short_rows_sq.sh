cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for cap in 32 16; do
rm -rf $O/sq_$cap
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --output-format csv -d $O/sq_$cap -- python3 $R/bench.py --max-samples $cap --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-check > $O/sq_$cap.log 2>&1 || tail -3 $O/sq_$cap.log
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/sq_$cap/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); dur={}
for r in csv.DictReader(open(f)):
    if "k_sample_walk" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"]); dur[r["Dispatch_Id"]]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
print("cap $cap", {c: round(sum(v.values())/len(v)) for c,v in acc.items()}, "us", round(sum(dur.values())/len(dur)/1e3,1))
PY
done

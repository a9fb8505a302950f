"""Durations (us) of every dispatch of the kernels whose name contains argv[2], from a rocprofv3 --kernel-trace directory."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        print(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")

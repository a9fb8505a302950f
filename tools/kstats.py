"""Condensed view of a rocprofv3 --kernel-trace --stats directory: kernel name (without arguments), calls, average microseconds."""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
for row in csv.DictReader(open(f)):
    name = row["Name"].split("(")[0].replace("void ", "")
    print(f'{name:45s} {int(row["Calls"]):6d} calls  {float(row["AverageNs"]) / 1e3:10.2f} us avg  {float(row["Percentage"]):6.2f} %')

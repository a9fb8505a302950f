#!/usr/bin/env python3
"""Per-kernel registers / scratch / LDS / occupancy of a csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per
kernel. usage: python tools/kres.py ltp_stage_kernels [name filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "longtermplanner_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src + ".hip"], cwd=d, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:60s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):4d} scratch {v.get('ScratchSize', -1):5d} vspill {v.get('VGPRs Spill', -1):4d} occ {v.get('Occupancy', -1):2d} LDS {v.get('LDS Size', -1):6d}")

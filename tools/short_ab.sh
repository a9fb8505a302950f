#!/bin/bash
# A/B of two builds of the library on the short-row bench lines, in ONE run (boxes differ by up to 15 % on these):
# every tools/ab/*.so (base.so, ...) against the library in the tree. usage: bash tools/short_ab.sh ["bench args" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp longtermplanner_amd/libltp_hip.so tools/ab/new.so
if [ $# -eq 0 ]; then set -- "--max-samples 64" "--receding 10:100 --max-samples 128" "--max-samples 256" "--max-samples 16"; fi
for rep in 1 2; do
for lib in $(cd tools/ab && ls *.so | sed s/.so// | grep -v "^new$") new; do
cp tools/ab/$lib.so longtermplanner_amd/libltp_hip.so
for v in "$@"; do
python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$lib', f'{d[\"value\"]/1e6:9.1f} M/s {d[\"ms_per_step\"]:9.3f} ms {r.get(\"achieved\")} GB/s', d['config']['workload'][70:140])"
done
done
done
cp tools/ab/new.so longtermplanner_amd/libltp_hip.so

#!/bin/bash
# A/B of the table pass vs the fused table build on the workloads the pass is meant for (runs on the GPU box).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-table_pass_ab.jsonl}
cd $R
: > $O
if [ -n "$SHORT" ]; then set -- "$1" "--max-samples 256" "--max-samples 64" "--receding 10:100 --max-samples 128" "--max-samples 256 --f32"
else set -- "$1" "--max-samples 256" "--max-samples 256 --f32" "--max-samples 64" "--max-samples 1024" "--receding 10:100 --max-samples 128" "--envelope 64:32" "--f32" "--sample-stride 4" ""; fi
shift
for v in "$@"; do
  for m in off on; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --table-pass $m --table-gib 16 $v >> $O 2>> $O.err || exit 1
    echo "done: $v / $m"
  done
done
python - <<PY
import json
for ln in open("$O"):
    d = json.loads(ln)
    r = d.get("roofline") or {}
    print(f'{d["config"]["table_pass"]:4s} {d["value"]/1e6:9.2f} M/s  {d["ms_per_step"]:8.3f} ms  {r.get("achieved", 0):7.1f} GB/s  {d["config"]["workload"][60:150]}')
PY

#!/bin/bash
# same box: the default capped-row lines of two builds (tools/ab/base.so, the tree's library), then the tree's library with --no-row-verdict
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/short_ab.sh "--max-samples 16" "--max-samples 32" "--max-samples 64" "--receding 10:100 --max-samples 64" "--semantics matlab --max-samples 32" "--semantics matlab --max-samples 64"
for v in "--max-samples 16" "--max-samples 32" "--max-samples 64" "--max-samples 128" "--receding 10:100 --max-samples 64" "--receding 10:100 --max-samples 32"; do
python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 --no-row-verdict $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('no-verdict', f'{d[\"value\"]/1e6:9.1f} M/s {d[\"ms_per_step\"]:9.3f} ms {r.get(\"achieved\")} GB/s', d['config']['workload'][70:200])"
done

#!/bin/bash
# kernel times of the short-row workloads (rocprofv3 --kernel-trace --stats). usage: bash tools/short_rows_prof.sh ["bench args" ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/short_rows_prof
mkdir -p $O
if [ $# -eq 0 ]; then set -- "--max-samples 64" "--max-samples 64 --no-walk" "--max-samples 256" "--receding 10:100 --max-samples 128"; fi
i=0
for v in "$@"; do
  i=$((i+1))
  rm -rf $O/v$i
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $v > $O/v$i.log 2>&1 || exit 1
  echo "== $v"; python3 $R/tools/kstats.py $O/v$i > $O/v$i.txt; head -8 $O/v$i.txt
done

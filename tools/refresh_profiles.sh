#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/refresh_profiles.sh): headline bench, workload variants and the rocprofv3
# passes whose summaries are kept under profiles/. Everything lands in gpurun_out/; profiles/summarize.py condenses it.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O && rm -rf $O/prof_stats $O/prof_write $O/prof_fetch
cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "default done"; cat $O/bench_default.json
: > $O/bench_variants.jsonl
for a in "--limits ref" "--limits ref30 --batch 200000" "--switch-only --batch 100000" "--switch-only" "--switch-only --limits ref" \
         "--max-samples 256" "--sample-stride 4" "--f32" "--f32 --limits ref" "--f32 --max-samples 256" "--envelope 64:32" \
         "--envelope 64:32 --limits ref" "--receding 10:100" "--receding 10:100 --max-samples 128" "--tile-gib 64" "--layout joint_major"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $a >> $O/bench_variants.jsonl 2>> $O/bench_variants.err || exit 1
  echo "variant $a done"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_stats.log 2>&1 || exit 1
echo "stats pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_write.log 2>&1 || exit 1
echo "WRITE_SIZE pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_fetch.log 2>&1 || exit 1
echo "FETCH_SIZE pass done"
# keep only the small CSVs
find $O/prof_stats $O/prof_write $O/prof_fetch -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" -delete

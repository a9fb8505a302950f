#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/refresh_profiles.sh [tag]): headline bench, workload variants and the rocprofv3
# passes whose summaries are kept under profiles/. Everything lands in gpurun_out/; profiles/summarize.py condenses it.
set -o pipefail
TAG=${1:-r06}
PART=${2:-all}      # all | bench (the bench lines) | prof (the rocprofv3 passes, first half) | prof2 (second half): separate gpurun calls, one would run out of time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
if [ "$PART" != bench ]; then rm -rf $O/prof_stats $O/prof_write $O/prof_fetch $O/prof_tab_stats $O/prof_switch_100000 $O/prof_switch_1000000 $O/prof_sq_first256 $O/prof_sq_first256_tab $O/prof_sq_first64 $O/prof_sq_first64_tab $O/prof_sq_switch100k $O/prof_sq_f32 $O/prof_sq_envelope $O/prof_f32_stats $O/prof_write_first256 $O/prof_write_stride4 $O/prof_write_f32 $O/prof_sq_switch1M $O/prof_env_stats $O/prof_sq_switch100k_exact $O/prof_switch_100000_exact; fi
cd $R
if [ "$PART" != prof ] && [ "$PART" != prof2 ]; then
timeout -k 10 400 python bench.py --secondary-file $O/bench_default_secondary.jsonl > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "default done"; cut -c1-300 $O/bench_default.json
: > $O/bench_variants.jsonl
for a in "--limits ref" "--limits ref30 --batch 200000" "--switch-only --batch 100000" "--switch-only --batch 100000 --end-limit" "--switch-only" \
         "--switch-only --end-limit" "--switch-only --limits ref" \
         "--switch-only --batch 100000 --in-flight 2 --steps 40 --warmup 4" "--switch-only --in-flight 2 --steps 20 --warmup 4" \
         "--max-samples 256" "--max-samples 256 --no-walk" "--max-samples 256 --table-pass off" "--max-samples 128" "--max-samples 128 --no-walk" \
         "--max-samples 64" "--max-samples 64 --no-walk" "--max-samples 32" "--max-samples 32 --no-walk" \
         "--max-samples 16" "--max-samples 16 --no-walk" "--max-samples 16 --no-auto-waves" "--max-samples 32 --no-auto-waves" "--max-samples 4" \
         "--receding 10:100 --max-samples 32" "--receding 10:100 --max-samples 16" \
         "--max-samples 16 --no-row-verdict" "--max-samples 32 --no-row-verdict" "--receding 10:100 --max-samples 32 --no-row-verdict" "--semantics matlab --max-samples 32" \
         "--f32 --max-samples 128" "--f32 --max-samples 128 --no-walk" \
         "--receding 10:100 --max-samples 64" "--receding 10:100 --max-samples 64 --no-walk" "--sample-stride 4" "--sample-stride 4 --no-walk" "--sample-stride 3" \
         "--sample-stride 2" "--max-samples 512" "--max-samples 512 --no-walk" "--f32" "--f32 --no-walk" "--f32 --sample-stride 4" "--f32 --sample-stride 4 --no-walk" "--f32 --limits ref" \
         "--f32 --max-samples 256" "--f32 --max-samples 256 --table-pass off" "--f32 --max-samples 1024" "--f32 --max-samples 1024 --no-walk" "--envelope 64:32" \
         "--envelope 64:32 --table-pass off" "--envelope 64:32 --limits ref" "--receding 10:100" "--receding 10:100 --end-limit" \
         "--receding 10:100 --max-samples 128" "--receding 10:100 --max-samples 128 --no-walk" "--receding 10:100 --max-samples 128 --table-pass off" "--tile-gib 64" "--layout joint_major" "--switch-only --layout joint_major" "--switch-only --batch 100000 --layout joint_major" \
         "--semantics matlab --steps 3" "--semantics matlab --limits ref --steps 2" "--semantics matlab --switch-only --batch 100000 --steps 30 --warmup 3" \
         "--semantics matlab --switch-only --batch 100000 --limits ref --steps 30 --warmup 3" "--semantics matlab --envelope 64:32" \
         "--switch-only --batch 100000 --pow-rule exact" "--switch-only --pow-rule exact" "--switch-only --limits ref --pow-rule exact" "--pow-rule exact" \
         "--max-samples 64 --pow-rule exact" "--receding 10:100 --max-samples 128 --pow-rule exact" "--envelope 64:32 --envelope-analytic" "--envelope 64:32 --limits ref --envelope-analytic" \
         "--envelope 256:8" "--envelope 256:8 --envelope-analytic" "--switch-only --batch 100000 --limits ref --pow-rule exact" "--switch-only --limits ref30 --batch 200000" "--semantics matlab --envelope 64:32 --envelope-analytic" \
         "--gpus 1 --force-dist --gather --checksum --steps 5" \
         "--gpus 8 --one-process --device 0 --global-batch 10000000 --switch-only --end-limit --checksum --steps 5" \
         "--gpus 8 --one-process --device 0 --global-batch 10000000 --envelope 64:32 --steps 2" \
         "--gpus 8 --one-process --device 0 --global-batch 10000000 --receding 10:100 --steps 2" \
         "--gpus 1 --global-batch 10000000 --switch-only --end-limit --checksum --steps 5"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $a >> $O/bench_variants.jsonl 2>> $O/bench_variants.err || exit 1
  echo "variant $a done"
done
# ceilings of the store pattern alone (DIAGNOSTIC: stores without arithmetic; not results)
: > $O/bench_dry.jsonl
for a in "" "--f32" "--max-samples 256 --table-pass off" "--limits ref"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --dry-sampler $a >> $O/bench_dry.jsonl 2>> $O/bench_variants.err || exit 1
  echo "dry $a done"
done
# N ranks rehearsed on this one GPU (gloo for the barrier; every rank on device 0): the launch path the 8-GPU node uses
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --device 0 --steps 3 > $O/bench_2ranks_gloo.json 2>> $O/bench_variants.err || exit 1
timeout -k 10 300 python bench.py --gpus 4 --backend gloo --device 0 --no-secondary --global-batch 1000000 --steps 3 > $O/bench_4ranks_gloo_global.json 2>> $O/bench_variants.err || exit 1
# BASELINE.json configs[3] at its real size, as far as one GPU goes: 10 M queries, four ranks (a box allows six GPU processes) sharing the device
timeout -k 10 400 python bench.py --gpus 4 --backend gloo --device 0 --no-cpu-baseline --no-secondary --global-batch 10000000 --checksum --tile-gib 96 --steps 2 > $O/bench_config4_4ranks_gloo.json 2>> $O/bench_variants.err || exit 1
echo "rank rehearsal done"
fi
if [ "$PART" = bench ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
if [ "$PART" != prof2 ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_stats.log 2>&1 || exit 1
echo "stats pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/prof_write.log 2>&1 || exit 1
echo "WRITE_SIZE pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/prof_fetch.log 2>&1 || exit 1
echo "FETCH_SIZE pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tab_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --max-samples 256 > $O/prof_tab_stats.log 2>&1 || exit 1
echo "table-pass stats pass done"
for b in 100000 1000000; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_switch_$b -- python3 $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --switch-only --batch $b > $O/prof_switch_$b.log 2>&1 || exit 1
done
echo "switching-times stats passes done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_first256 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-samples 256 > $O/prof_sq_first256.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_first256_tab -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-samples 256 --no-walk > $O/prof_sq_first256_tab.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_first64 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-samples 64 > $O/prof_sq_first64.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_first64_tab -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-samples 64 --no-walk > $O/prof_sq_first64_tab.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_switch100k -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --switch-only --batch 100000 > $O/prof_sq_switch100k.log 2>&1 || exit 1
fi
if [ "$PART" = prof ]; then find $O/prof_* -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.log" -delete; exit 0; fi
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_f32 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --f32 > $O/prof_sq_f32.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_envelope -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --envelope 64:32 > $O/prof_sq_envelope.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --f32 > $O/prof_f32_stats.log 2>&1 || exit 1
# round 5: what profiles/bench_counters.json (quoted by bench.py's roofline blocks) is made from — WRITE_SIZE of the walk-kernel lines, the
# stage kernels at 1 M and under the opt-in pow rule, the envelope kernel's time
for v in "first256:--max-samples 256" "stride4:--sample-stride 4" "f32:--f32"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write_${v%%:*} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ${v#*:} > $O/prof_write_${v%%:*}.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_switch1M -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --switch-only > $O/prof_sq_switch1M.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_switch100k_exact -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --switch-only --batch 100000 --pow-rule exact > $O/prof_sq_switch100k_exact.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_switch_100000_exact -- python3 $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --switch-only --batch 100000 --pow-rule exact > $O/prof_switch_100000_exact.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_env_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --envelope 64:32 > $O/prof_env_stats.log 2>&1 || exit 1
echo "SQ counter passes done"
# keep only the small CSVs
find $O/prof_write_first256 $O/prof_write_stride4 $O/prof_write_f32 $O/prof_sq_switch1M $O/prof_env_stats $O/prof_sq_switch100k_exact $O/prof_switch_100000_exact $O/prof_stats $O/prof_write $O/prof_fetch $O/prof_tab_stats $O/prof_switch_100000 $O/prof_switch_1000000 $O/prof_sq_first256 $O/prof_sq_first256_tab $O/prof_sq_first64 $O/prof_sq_first64_tab $O/prof_sq_switch100k $O/prof_sq_f32 $O/prof_sq_envelope $O/prof_f32_stats -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" -delete 2>/dev/null || true

cd $GRAFT_REPO_ROOT
: > gpurun_out/matlab_variants.jsonl
for a in "--semantics matlab --max-samples 32" "--semantics matlab --steps 3" "--semantics matlab --limits ref --steps 2" "--semantics matlab --switch-only --batch 100000 --steps 30 --warmup 3" \
         "--semantics matlab --switch-only --batch 100000 --limits ref --steps 30 --warmup 3" "--semantics matlab --envelope 64:32" "--semantics matlab --envelope 64:32 --envelope-analytic" \
         "--semantics matlab --switch-only --batch 100000 --steps 30 --warmup 3 --pow-rule exact" "--semantics matlab --switch-only --steps 10 --warmup 2"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $a >> gpurun_out/matlab_variants.jsonl 2>> gpurun_out/matlab_variants.err || exit 1
  echo "variant $a done"
done

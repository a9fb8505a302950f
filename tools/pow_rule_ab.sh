#!/bin/bash
# A/B of the pow rule (ltp_set_pow_rule): switching times only at 100 k / 1 M, both limit sets, and the headline. One box.
set -e
out=gpurun_out/r05_pow_rule_ab.jsonl; : > $out
for rule in exact libm; do
  for lim in panda ref; do
    python bench.py --switch-only --batch 100000 --limits $lim --steps 40 --warmup 5 --pow-rule $rule --no-cpu-baseline --no-secondary >> $out
    python bench.py --switch-only --batch 1000000 --limits $lim --steps 20 --warmup 3 --pow-rule $rule --no-cpu-baseline --no-secondary >> $out
  done
  python bench.py --steps 10 --warmup 2 --pow-rule $rule --no-cpu-baseline --no-secondary --no-rccl-check >> $out
  python bench.py --switch-only --batch 100000 --semantics matlab --steps 40 --warmup 5 --pow-rule $rule --no-cpu-baseline --no-secondary >> $out
done
python - <<'P'
import json
for l in open('gpurun_out/r05_pow_rule_ab.jsonl'):
    o=json.loads(l); c=o['config']; print(c['pow_rule'], c['limits'], c['semantics'], c['workload'][:60], o['ms_per_step'], o['value'])
P

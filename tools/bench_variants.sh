#!/bin/bash
# The single-GPU bench variants the reviews track (short rows, float32 rows, consumers), one JSON line each into $1 (a .jsonl).
out=${1:-gpurun_out/bench_variants.jsonl}
: > "$out"
run() { python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 "$@" >> "$out" 2>> "$out.err" || echo "{\"error\": \"$*\"}" >> "$out"; }
run --f32
run --max-samples 256
run --max-samples 256 --f32
run --max-samples 64
run --sample-stride 4
run --envelope 64:32
run --receding 10:100 --max-samples 128
run --receding 10:100
run --limits ref --f32 --steps 2 --warmup 1
run --semantics matlab --steps 2 --warmup 1
run --semantics matlab --switch-only --batch 100000 --steps 30 --warmup 3
run --semantics matlab --limits ref --switch-only --batch 100000 --steps 30 --warmup 3
run --switch-only --batch 100000 --steps 50 --warmup 5
run --switch-only --batch 100000 --limits ref --steps 50 --warmup 5
python - "$out" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "error" in d:
        print("ERROR", d["error"]); continue
    r = d.get("roofline") or {}
    print(f'{d["value"]/1e6:9.1f} M/s {d["ms_per_step"]:9.3f} ms  {r.get("achieved")} GB/s {r.get("kernel","")}  | {d["config"]["workload"][70:190]}')
PY

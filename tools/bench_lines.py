"""Runs bench.py once per argument string and prints one condensed line each (value, ms per step, sampler GB/s).
usage: python tools/bench_lines.py "--max-samples 64" "--semantics matlab --switch-only --batch 100000" ..."""
import json, os, shlex, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for a in sys.argv[1:]:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + shlex.split(a), capture_output=True, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print(f"{a:70s} FAILED rc={p.returncode} {p.stderr[-300:]}")
        continue
    d = json.loads(lines[-1]); r = d.get("roofline") or {}
    print(f"{a:70s} {d['value'] / 1e6:10.2f} M/s {d['ms_per_step']:9.3f} ms  {r.get('achieved')} GB/s  {r.get('kernel', '')}", flush=True)

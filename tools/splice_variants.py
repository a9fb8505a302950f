"""Replaces (or appends) lines of profiles/<tag>_bench_variants.jsonl by freshly measured ones: python tools/splice_variants.py r05 gpurun_out/matlab_variants.jsonl
A line is identified by (metric, workload text, pow rule, envelope mode)."""
import json, sys
tag, src = sys.argv[1], sys.argv[2]
new = [json.loads(l) for l in open(src) if l.strip().startswith("{")]
key = lambda d: (d["metric"], d["config"]["workload"], d["config"].get("pow_rule"), d["config"].get("envelope_mode"))
newmap = {key(d): d for d in new}
path = f"profiles/{tag}_bench_variants.jsonl"
out, used = [], set()
for l in open(path).read().split("\n"):
    if not l.strip().startswith("{"):
        out.append(l)
        continue
    k = key(json.loads(l))
    if k in newmap:
        out.append(json.dumps(newmap[k])); used.add(k)
    else:
        out.append(l)
while out and not out[-1].strip():
    out.pop()
out += [json.dumps(d) for d in new if key(d) not in used]
open(path, "w").write("\n".join(out) + "\n")
print("replaced", len(used), "appended", len(new) - len(used))

cd $GRAFT_REPO_ROOT
for ms in 64 256; do for g in 0.03 0.06 0.125 0.25 0.5 2; do
python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 --max-samples $ms --table-gib $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('cap $ms table-gib $g', f'{d[\"value\"]/1e6:9.1f} M/s {d[\"ms_per_step\"]:9.3f} ms {r.get(\"achieved\")} GB/s')"
done; done

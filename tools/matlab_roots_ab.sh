#!/bin/bash
# A/B of two builds of the library on the MATLAB-semantics bench lines, in ONE run on one box: every tools/exp/*.so against the
# library in the tree (round 5: the LAPACK restatement with its matrix in dynamic LDS vs in registers; queued lanes dealt 64 to a
# block vs spread over the blocks that run at once), MATLAB and C++ semantics.
# usage: bash tools/matlab_roots_ab.sh        (writes gpurun_out/matlab_roots_ab.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
cp longtermplanner_amd/libltp_hip.so /tmp/new.so
out=gpurun_out/matlab_roots_ab.txt
: > $out
for rep in 1 2; do
for lib in $(cd tools/exp && ls *.so | sed 's/\.so$//') new; do
if [ $lib = new ]; then cp /tmp/new.so longtermplanner_amd/libltp_hip.so; else cp tools/exp/$lib.so longtermplanner_amd/libltp_hip.so; fi
for v in "--semantics matlab --switch-only --batch 100000 --steps 40 --warmup 5" "--semantics matlab --switch-only --batch 100000 --steps 40 --warmup 5 --pow-rule exact" \
         "--semantics matlab --switch-only --batch 1000000 --steps 10 --warmup 2" "--semantics matlab --switch-only --batch 100000 --steps 40 --warmup 5 --limits ref" \
         "--switch-only --batch 100000 --steps 40 --warmup 5" "--switch-only --batch 100000 --steps 40 --warmup 5 --limits ref" "--switch-only --batch 1000000 --steps 10 --warmup 2" \
         "--switch-only --batch 10000 --steps 40 --warmup 5"; do
python bench.py --no-cpu-baseline --no-secondary --no-rccl-check --checksum $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$v', f'{d[\"value\"]/1e6:9.1f} M/s {d[\"ms_per_step\"]:9.4f} ms', 'checksum', d.get('config',{}).get('records_checksum'))" >> $out
done
done
done
cp /tmp/new.so longtermplanner_amd/libltp_hip.so
cat $out

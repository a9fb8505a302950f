#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of one bench line under a list of environment settings, on one box
# (the LTP_EXP_* shapes of the stage kernels are read by a library built with -DLTP_EXP_KNOBS only: make -C longtermplanner_amd/csrc clean all EXTRA=-DLTP_EXP_KNOBS)
# usage: bash tools/exp_env_ab.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...   -> gpurun_out/env_ab.txt (appended)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
args="$1"; shift
i=0
for e in "base" "$@"; do
  i=$((i+1)); rm -rf $O/eab_$i
  if [ "$e" = base ]; then envs="LTP_NOTHING=1"; else envs="$e"; fi
  ( export $envs; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eab_$i -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-rccl-check $args > $O/eab_$i.log 2>&1 ) || { tail -3 $O/eab_$i.log; exit 1; }
  echo "== [$e]  $args  $(python3 -c "import json,sys; d=json.loads([l for l in open('$O/eab_$i.log') if l.startswith('{')][-1]); print(d['ms_per_step'], 'ms/step')")" >> $O/env_ab.txt
  python3 $R/tools/kstats.py $O/eab_$i | head -4 >> $O/env_ab.txt
done

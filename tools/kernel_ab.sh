#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of one bench line under every tools/exp/*.so and the library in the tree, on one box
# usage: bash tools/kernel_ab.sh "<bench args>"   -> gpurun_out/kernel_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cp $R/longtermplanner_amd/libltp_hip.so /tmp/new.so
cd /tmp && export TMPDIR=/tmp
: > $O/kernel_ab.txt
for lib in $(cd $R/tools/exp && ls *.so | sed 's/\.so$//') new; do
if [ $lib = new ]; then cp /tmp/new.so $R/longtermplanner_amd/libltp_hip.so; else cp $R/tools/exp/$lib.so $R/longtermplanner_amd/libltp_hip.so; fi
rm -rf $O/kab_$lib
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kab_$lib -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-rccl-check $1 > $O/kab_$lib.log 2>&1 || exit 1
echo "== $lib  $1" >> $O/kernel_ab.txt
python3 $R/tools/kstats.py $O/kab_$lib | head -6 >> $O/kernel_ab.txt
done
cp /tmp/new.so $R/longtermplanner_amd/libltp_hip.so
cat $O/kernel_ab.txt

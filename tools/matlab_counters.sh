#!/bin/bash
# gpurun -- bash tools/matlab_counters.sh : the SQ counter and kernel-stats passes of the MATLAB-semantics 100 k switching-times line, then
# (here or back in the container) python tools/matlab_counters_merge.py adds them to profiles/bench_counters.json under the line's
# config.workload_key ("panda:100000:f64:switch_only:matlab"), as tools/collect_profiles.sh does for the C++-semantics lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
rm -rf $O/prof_sq_switch100k_matlab $O/prof_switch_100000_matlab
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/prof_sq_switch100k_matlab -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-check --switch-only --batch 100000 --semantics matlab > $O/prof_sq_switch100k_matlab.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_switch_100000_matlab -- python3 $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-check --switch-only --batch 100000 --semantics matlab > $O/prof_switch_100000_matlab.log 2>&1 || exit 1
echo "passes done"

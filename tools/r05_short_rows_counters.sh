#!/bin/bash
# what bounds the short rows of k_sample_walk? Per cap (first-16/32/64/256 rows, 1 M panda plans) and for the dry
# sampler (stores without arithmetic) at the same geometry: kernel time (--stats), SQ counters, TCC/EA write-request counters, TCP/TLB
# counters — each in a pass of its own (counters never together with other trace domains). usage: bash tools/r05_short_rows_counters.sh [caps...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_short_rows
mkdir -p $O
CAPS=${@:-"16 32 64 256"}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR"
SQ2="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
TCC="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_WRITE TCC_REQ TCC_TAG_STALL"
TCC2="TCC_EA0_WRREQ_LEVEL TCC_BUSY TCC_CYCLE TCC_NORMAL_WRITEBACK TCC_WRITEBACK TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL TCC_STREAMING_REQ"
TCP="TCP_TCC_WRITE_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_WRITE TCP_TCC_WRITE_REQ_LATENCY TCP_UTCL1_REQUEST"
for cap in $CAPS; do
  for dry in "" "--dry-sampler"; do
    tag=cap${cap}${dry:+_dry}
    args="--max-samples $cap $dry --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
    rm -rf $O/$tag; mkdir -p $O/$tag
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/stats -- python3 $R/bench.py $args > $O/$tag/stats.log 2>&1 || { echo "$tag stats failed"; tail -3 $O/$tag/stats.log; exit 1; }
    for set in SQ SQ2 TCC TCC2 TCP; do
      timeout -k 10 150 rocprofv3 --kernel-trace --pmc ${!set} --output-format csv -d $O/$tag/$set -- python3 $R/bench.py $args > $O/$tag/$set.log 2>&1 || { echo "$tag $set failed"; tail -3 $O/$tag/$set.log; }
    done
    echo "done $tag"
  done
done
python3 $R/tools/r05_short_rows_summary.py $O > $O/summary.json && cat $O/summary.json | head -c 3000

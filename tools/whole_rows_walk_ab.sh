#!/bin/bash
# A/B: whole / long / sparse rows through k_sample_walk_* (--walk) against the fused sampler k_sample (--no-walk), same box.
#   tools/whole_rows_walk_ab.sh [out.txt] [more]
out=${1:-gpurun_out/r04_whole_rows_walk_ab.txt}
: > $out
line() {
  echo "== bench.py $*" >> $out
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'GB/s', r['achieved'], r.get('kernel'), 'launch ms', r.get('avg_launch_ms'))" >> $out || exit 1
}
ab() { for extra in "--no-walk" "--walk"; do line "$@" $extra || exit 1; done; }
if [ "$2" = "more" ]; then
  ab --sample-stride 3
  ab --limits ref --batch 400000 --sample-stride 4
  ab --limits ref --batch 400000 --f32
  ab --max-samples 2048
  ab --max-samples 1024
  ab --limits ref30 --batch 100000 --sample-stride 4
else
  ab
  ab --sample-stride 2
  ab --sample-stride 4
  ab --f32
  ab --f32 --sample-stride 4
  ab --limits ref --batch 400000
  ab --max-samples 512
fi
cat $out

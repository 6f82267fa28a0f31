#!/bin/bash
# Kernel timeline of the forward at small batch sizes (hipGraph replay): tools/profile_small_batch.sh <tag> "1 8"
set -u
tag=$1; sizes=${2:-"1 8"}
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/prof_$tag; mkdir -p $out
for b in $sizes; do
  cmd="python $root/bench.py --pairs $b --graph --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
  (cd /tmp && rocprofv3 --kernel-trace -d $root/$out/t$b -o $tag -- $cmd) > $out/trace_$b.log 2>&1
  db=$(ls $out/t$b/*/*_results.db $out/t$b/*_results.db 2>/dev/null | head -1)
  python tools/rocpd_timeline.py $db list > $out/${tag}_timeline_pairs$b.txt 2>&1
  rm -rf $out/t$b
  head -40 $out/${tag}_timeline_pairs$b.txt
done
exit 0

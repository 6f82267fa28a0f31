#!/bin/bash
# Everything a round commits under profiles/ from ONE build, in one call on the GPU box (run from the repo root under gpurun):
#   tools/refresh_profiles.sh r2_i
# kernel-trace summary + PMC passes of the forward (tools/profile_round.sh), the traffic record moved to where bench.py looks for
# it, the full default bench line, the kernel-trace summary of the training step and its bench line.  Outputs: gpurun_out/prof_<tag>/
# (the rocprofv3 databases are deleted once summarised: gpurun_out/ travels back and is size-limited).
set -u
tag=$1
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/prof_$tag
bash tools/profile_round.sh $tag pmc > /dev/null 2>&1
cp $out/${tag}_pmc_traffic.json profiles/r3_pmc_traffic.json
rm -rf $out/trace $out/pmc_*
python bench.py --steps 20 --warmup 3 > $out/${tag}_bench_forward_pairs64.json 2> $out/bench_forward.err
cat $out/${tag}_bench_forward_pairs64.json
cmd="python $root/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline --no-roofline"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/train -o ${tag}_train -- $cmd) > $out/train.log 2>&1
db=$(ls $out/train/*/*_results.db $out/train/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db > $out/${tag}_train_step_kernel_stats_dpt_pairs32.md
rm -rf $out/train
python bench.py --mode train --steps 10 --warmup 3 > $out/${tag}_bench_train_dpt_pairs32.json 2> $out/bench_train.err
cat $out/${tag}_bench_train_dpt_pairs32.json

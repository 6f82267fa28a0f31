#!/bin/bash
# Profiles of the headline bench on the GPU box (run from the repo root under gpurun):
#   tools/profile_round.sh r2_a            kernel-trace summary only
#   tools/profile_round.sh r2_a pmc        + FETCH_SIZE / WRITE_SIZE / MFMA-busy passes and the traffic record bench.py reads
# Outputs land in gpurun_out/prof_<tag>/ (copy what is to be kept into profiles/).
set -u
tag=$1; mode=${2:-trace}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag; mkdir -p $out
root=$(pwd)
cmd="python $root/bench.py --steps 5 --warmup 2 --single-stream --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/trace -o $tag -- $cmd) > $out/trace.log 2>&1
db=$(ls $out/trace/*/*_results.db $out/trace/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db > $out/${tag}_forward_kernel_stats_pairs128.md
python tools/rocpd_dispatches.py $db > $out/${tag}_forward_dispatches_pairs128.txt
head -30 $out/${tag}_forward_kernel_stats_pairs128.md
if [ "$mode" = "pmc" ]; then
  cmd2="python $root/bench.py --steps 2 --warmup 1 --single-stream --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $root/$out/pmc_$c -o $tag -- $cmd2) > $out/pmc_$c.log 2>&1
    dbc=$(ls $out/pmc_$c/*/*_results.db $out/pmc_$c/*_results.db 2>/dev/null | head -1)
    python tools/rocpd_pmc.py $dbc > $out/${tag}_pmc_${c}_bench_pairs128.txt
  done
  python tools/pmc_traffic.py $out $tag > $out/${tag}_pmc_traffic.json
  cat $out/${tag}_pmc_traffic.json
fi

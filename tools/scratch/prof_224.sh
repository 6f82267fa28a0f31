set -u
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/prof_r6_n; mkdir -p $out
base="--steps 3 --warmup 2 --single-stream --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/t224 -o t224 -- python $root/bench.py --img 224 --pairs 256 $base) > $out/t224.log 2>&1
db=$(ls $out/t224/*/*_results.db $out/t224/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db > $out/r6_n_forward_kernel_stats_224_pairs256.md
python tools/rocpd_dispatches.py $db > $out/r6_n_forward_dispatches_224_pairs256.txt
rm -rf $out/t224

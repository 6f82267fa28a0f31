set -u
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/cfg3; mkdir -p $out
cmd="python $root/bench.py --encoder dinov2 --img 518 --pairs 32 --steps 4 --warmup 2 --single-stream --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/t -o c3 -- $cmd) > $out/trace.log 2>&1
db=$(ls $out/t/*/*_results.db $out/t/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db > $out/cfg3_stats.md
python tools/rocpd_dispatches.py $db > $out/cfg3_dispatches.txt
rm -rf $out/t
head -34 $out/cfg3_stats.md | cut -c1-150

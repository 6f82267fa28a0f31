set -u
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/tt; mkdir -p $out
cmd="python $root/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --single-stream"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/train -o tt -- $cmd) > $out/train.log 2>&1
db=$(ls $out/train/*/*_results.db $out/train/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db > $out/train_stats.md
rm -rf $out/train
python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TRAIN', d['value'], d['ms_per_step'])"

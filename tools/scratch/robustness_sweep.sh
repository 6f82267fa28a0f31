# non-default configurations at bench-like sizes: each must run and print a finite line (round 6: the fp32-class training step had
# never been run at bench sizes and failed there)
base="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
run() { echo "== $*"; python bench.py $* $base 2>&1 | tail -1 | python -c "
import sys, json
t = sys.stdin.read().strip()
try:
    l = json.loads(t.splitlines()[-1]); print('   ok', l['value'], l['unit'], 'ms/step', l['ms_per_step'], 'peak GB', l.get('peak_hbm_gb'))
except Exception as e:
    print('   FAILED:', t[-600:])
"; }
run --mode train --encoder dinov2 --pairs 16
run --mode train --img 224 --pairs 128
run --mode train --head linear --pairs 64
run --mode train --precision fp32 --pairs 2
run --precision fp32 --pairs 4
run --img 1024 --pairs 8 --attention fp8
run --img 1024 --pairs 8 --mode train
run --graph --pairs 4
run --encoder dinov2 --pairs 64 --head linear
run --img 384 --pairs 128
run --img 224 --pairs 512 --head linear

# batch-size sweep of the 224x224 and DINOv2-518 forwards (tile-count quantisation of the GEMM launches: 256 CUs x 256-row tiles)
base="--steps 8 --warmup 3 --no-cpu-baseline --no-reference-policy --no-extra-legs --no-roofline"
for p in 224 240 250 256 272 288 292 320 384; do
  python bench.py --img 224 --pairs $p $base 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224', $p, l['value'], l['ms_per_step'])"
done
for p in 24 32 40 47 48 56 64 94; do
  python bench.py --encoder dinov2 --pairs $p $base 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dinov2', $p, l['value'], l['ms_per_step'])"
done

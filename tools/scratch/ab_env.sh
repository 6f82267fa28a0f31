#!/bin/bash
# usage: ab_env.sh VAR "bench args" v1 v2 ...   — one short bench line per value of the environment variable VAR
var=$1; shift; args=$1; shift
for v in "$@"; do
  env $var=$v timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-legs --no-reference-policy 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print('$var=$v', d['config'].get('pairs_per_gpu'), d['value'], d['ms_per_step'])
    except Exception as e: print(l[:300])"
done

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run from the repo root under gpurun): tools/calib_fetch.sh
set -u
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/calib_fetch; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $root/$out/$c -o x -- python $root/tools/calib_fetch.py) > $out/$c.log 2>&1
  db=$(ls $out/$c/*/*_results.db $out/$c/*_results.db 2>/dev/null | head -1)
  python tools/rocpd_pmc_by_grid.py $db > $out/calib_$c.txt
  cat $out/calib_$c.txt
  rm -rf $out/$c
done
grep "^gemm" $out/FETCH_SIZE.log

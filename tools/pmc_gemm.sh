#!/bin/bash
# SQ counters of one dense GEMM shape per tile variant (run from the repo root under gpurun): tools/pmc_gemm.sh M N K "2 6"
set -u
M=$1; N=$2; K=$3; variants=${4:-"2 6"}
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/pmc_gemm; mkdir -p $out
for v in $variants; do
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
    tag=v${v}_$(echo $c | tr ' ' '_')
    (cd /tmp && UC_GEMM_VARIANT=$v rocprofv3 --kernel-trace --pmc $c -d $root/$out/$tag -o x -- python $root/tools/loop_gemm.py $M $N $K 0.5) > $out/$tag.log 2>&1
    db=$(ls $out/$tag/*/*_results.db $out/$tag/*_results.db 2>/dev/null | head -1)
    echo "== variant $v: $c"; python tools/rocpd_pmc.py $db | grep -v "^gemm\|^ *$" 
  done
done

#!/bin/bash
# SQ counters of the bf16 attention forward (B=64, H=16, N=1024) per workgroup size: tools/pmc_attn.sh "4 8"   (UC_ATTN_NW values)
set -u
forms=${1:-"4 8"}
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/pmc_attn; mkdir -p $out
for v in $forms; do
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=nw${v}_$(echo $c | tr ' ' '_')
    (cd /tmp && UC_ATTN_NW=$v rocprofv3 --kernel-trace --pmc $c -d $root/$out/$tag -o x -- python $root/tools/scratch/probe_attn_anatomy.py run) > $out/$tag.log 2>&1
    db=$(ls $out/$tag/*/*_results.db $out/$tag/*_results.db 2>/dev/null | head -1)
    echo "== NW $v: $c"; python tools/rocpd_pmc.py $db | grep -A3 "attn_bf16" | grep -v "^attn"
  done
done

#!/bin/bash
# SQ counters of the bf16 attention backward kernels (B=64, H=16, N=1024; the stand-alone probe binary): tools/pmc_attn_bwd.sh
set -u
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/pmc_attn_bwd; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Wno-inline-asm tools/probes/attn_bwd64_probe.hip -o /tmp/b64 || exit 1
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  tag=$(echo $c | tr ' ' '_')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $root/$out/$tag -o x -- /tmp/b64 64 16 1024) > $out/$tag.log 2>&1
  db=$(ls $out/$tag/*/*_results.db $out/$tag/*_results.db 2>/dev/null | head -1)
  echo "== $c"; python tools/rocpd_pmc.py $db | grep -A2 "attn_bwd_d" 
done
rm -rf $out

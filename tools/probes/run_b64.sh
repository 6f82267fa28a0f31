#!/bin/bash
# builds and runs the stand-alone backward probe on the GPU box:  gpurun -- bash tools/probes/run_b64.sh ["flag set 1" "flag set 2" ...]
mkdir -p gpurun_out
: > gpurun_out/b64_probe.txt
if [ $# -eq 0 ]; then set -- ""; fi
for fl in "$@"; do
  echo "== flags: $fl" | tee -a gpurun_out/b64_probe.txt
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Wno-inline-asm $fl tools/probes/attn_bwd64_probe.hip -o /tmp/b64 || continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Wno-inline-asm -DB64_TIMING $fl tools/probes/attn_bwd64_probe.hip -o /tmp/b64t || continue
  for args in "128 16 1024" "16 16 4096"; do timeout 120 /tmp/b64 $args; done 2>&1 | grep -v "dkv32" | tee -a gpurun_out/b64_probe.txt
  timeout 120 /tmp/b64t 128 16 1024 2>&1 | grep "wg    9\|wg 9" | tee -a gpurun_out/b64_probe.txt
done

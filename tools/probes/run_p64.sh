#!/bin/bash
# builds and runs the stand-alone attention probe on the GPU box:  gpurun -- bash tools/probes/run_p64.sh
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/attn_p64_probe.hip -o /tmp/p64
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DP64_TIMING tools/probes/attn_p64_probe.hip -o /tmp/p64t

if [ -n "$P64_QUICK" ]; then for args in "128 16 1024"; do timeout 120 /tmp/p64 $args; done; timeout 120 /tmp/p64t 128 16 1024; P64_PRESCALED=1 timeout 120 /tmp/p64t 128 16 1024; exit 0; fi
for args in "2 4 512" "1 16 1024" "2 3 1024 1024 1" "2 3 1024 1024 2" "1 2 256 100" "2 16 1370" "3 5 300 1370 1" "64 16 1024" "128 16 1024" "128 16 1024 1024 0 256" "128 12 1024" "16 16 4096" "64 16 1370"; do
  timeout 120 /tmp/p64 $args || echo "FAILED: $args"
done 2>&1 | tee gpurun_out/p64_probe.txt
for args in "128 16 1024 1024 0 256" "128 16 1024 1024 0 512" "16 16 4096 4096 0 512"; do timeout 120 /tmp/p64t $args; done 2>&1 | tee gpurun_out/p64_probe_timing.txt

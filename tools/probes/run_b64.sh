#!/bin/bash
# builds and runs the stand-alone backward probe on the GPU box:  gpurun -- bash tools/probes/run_b64.sh
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude $B64_FLAGS tools/probes/attn_bwd64_probe.hip -o /tmp/b64
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DB64_TIMING $B64_FLAGS tools/probes/attn_bwd64_probe.hip -o /tmp/b64t
for args in "128 16 1024" "16 16 4096"; do timeout 120 /tmp/b64 $args; done 2>&1 | tee gpurun_out/b64_probe.txt
timeout 120 /tmp/b64t 128 16 1024 2>&1 | tee -a gpurun_out/b64_probe.txt

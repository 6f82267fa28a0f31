#!/bin/bash
# samples rocm-smi power / sclk while a command runs:  power_watch.sh <label> <cmd...>
label=$1; shift
( for i in $(seq 1 12); do sleep 0.4; rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'sclk\|Average Graphics Package Power\|Current Socket' | tr '\n' ' '; echo; done ) > gpurun_out/power_$label.txt &
W=$!
"$@" > gpurun_out/power_${label}_cmd.txt 2>&1
wait $W
echo "== $label"; cat gpurun_out/power_$label.txt | sed 's/  */ /g' | cut -c1-200

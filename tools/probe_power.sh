#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while a command runs: is the chip at its power cap under the GEMM load?
"$@" > /tmp/_pp.log 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -E "Power|sclk|fclk|mclk" | tr '\n' ' '; echo
  sleep 1
done
wait $pid
tail -3 /tmp/_pp.log

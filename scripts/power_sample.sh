#!/bin/bash
# Board power and clocks while the dominant kernel runs back to back (evidence for the clock give-back discussion in DESIGN.md).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
REPS=${REPS:-40000} python scripts/bench_conv.py encAB_res_256 > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Graphics Package Power|sclk" | tr -s ' \t' ' ' | sort -t: -k1,1 | head -20
  echo ---
  sleep 1.5
done
wait $BP
tail -1 gpurun_out/power_bench.log

#!/bin/bash
# as prof_bench.sh with the two RefineNet heads on ONE stream (FP_HEADS_SERIAL=1): every kernel's duration is its duration alone on the chip
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export FP_HEADS_SERIAL=1
OUT=gpurun_out/prof_bench_serial
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
tail -c 300 $OUT/bench.log

#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench command -> gpurun_out/prof_bench/ (copy *_kernel_stats.csv to profiles/)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
ls $OUT
tail -c 600 $OUT/bench.log

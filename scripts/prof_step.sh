#!/bin/bash
# kernel trace of the bench step, per-(kernel, grid) medians: bash scripts/prof_step.sh [tag] [ENV=VAL ...]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-step}; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/run.log 2>&1
python3 scripts/ktrace.py $OUT 0.25 > $OUT/summary.txt 2>&1
head -45 $OUT/summary.txt | cut -c1-170

#!/bin/bash
# per-kernel durations of a shard's local step (WORLD=8: 32 hypotheses, 4: 63, 2: 126); usage: bash scripts/prof_shard.sh 8 [tag]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
W=${1:-8}
TAG=${2:-shard$W}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
WORLD=$W STEPS=10 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 scripts/shard_steps.py > $OUT/run.log 2>&1
python3 scripts/ktrace.py $OUT 0.6 > $OUT/summary.txt 2>&1
tail -3 $OUT/run.log | head -1
head -60 $OUT/summary.txt

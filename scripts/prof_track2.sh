#!/bin/bash
# kernel trace of tracking frames, per-(kernel, grid) medians + idle: bash scripts/prof_track2.sh one|multi [tag]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
MODE=${1:-one}
TAG=${2:-track_$MODE}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
MODE=$MODE FRAMES=60 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 scripts/track_steps.py > $OUT/run.log 2>&1
python3 scripts/ktrace.py $OUT 0.5 > $OUT/summary.txt 2>&1
tail -2 $OUT/run.log | head -1
head -70 $OUT/summary.txt | cut -c1-170

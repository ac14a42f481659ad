#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "FP_S2W_ROWS=2" "FP_S2W_ROWS=4" "FP_S2W_OFF=1"; do
  OUT=gpurun_out/prof_s2w_$cfg; rm -rf $OUT; mkdir -p $OUT
  export $cfg
  REPS=30 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 scripts/bench_conv.py down_64_128 > $OUT/run.log 2>&1
  unset FP_S2W_ROWS FP_S2W_OFF
  echo "== $cfg"; python3 scripts/kstats.py $OUT conv | head -4
done

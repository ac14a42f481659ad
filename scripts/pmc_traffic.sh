#!/bin/bash
# HBM / fabric traffic of every kernel of one bench step: FETCH_SIZE and WRITE_SIZE in SEPARATE counter passes
# (MI355X_MICROARCH.md, HBM / rocprofv3 section); no tracing domains beside --kernel-trace.  Run on the GPU box:
#   gpurun -- 'bash scripts/pmc_traffic.sh'   then   python scripts/pmc_aggregate.py   (here)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/write.log 2>&1
ls $OUT/fetch $OUT/write

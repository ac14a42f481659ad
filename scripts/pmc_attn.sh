#!/bin/bash
# SQ counter passes over the attention core alone (scripts/bench_attn.py); separate runs, no tracing domains beside --kernel-trace.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn2
rm -rf $OUT && mkdir -p $OUT
REPS=5 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a -o a --output-format csv -- python3 scripts/bench_attn.py > $OUT/a.log 2>&1
REPS=5 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/b -o b --output-format csv -- python3 scripts/bench_attn.py > $OUT/b.log 2>&1
ls $OUT/a $OUT/b

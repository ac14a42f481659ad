#!/bin/bash
# SQ counter passes over the three 3x3 stride-1 layer shapes (separate runs: 8 SQ slots per pass; no tracing domains).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_sq
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
REPS=5 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a -o a --output-format csv -- python3 scripts/bench_conv.py encA_res_128 encAB_res_256 encAB_res_512 > $OUT/a.log 2>&1
REPS=5 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/b -o b --output-format csv -- python3 scripts/bench_conv.py encA_res_128 encAB_res_256 encAB_res_512 > $OUT/b.log 2>&1
ls -R $OUT | head -30

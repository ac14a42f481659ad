#!/bin/bash
# SQ counter passes over the strided layers' kernels (stem.hip, conv_s2.hip, conv_igemm2 for 64->128) and one 3x3 stride-1 layer
# for comparison: MFMA-pipe busy cycles, LDS bank-conflict cycles against LDS-array cycles, wave cycles.  Separate runs, no tracing
# domains beside --kernel-trace.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_strided
rm -rf $OUT && mkdir -p $OUT
REPS=5 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/a -o a --output-format csv -- python3 scripts/bench_conv.py stem_7x7 down_256_512 down_64_128 encAB_res_256 > $OUT/a.log 2>&1
REPS=5 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT/b -o b --output-format csv -- python3 scripts/bench_conv.py stem_7x7 down_256_512 down_64_128 encAB_res_256 > $OUT/b.log 2>&1
ls $OUT/a $OUT/b

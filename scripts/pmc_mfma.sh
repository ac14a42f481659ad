#!/bin/bash
# MFMA utilisation of every kernel of a bench step: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in one counter pass (kernel trace only
# beside it).  Run on the GPU box:  gpurun -- 'bash scripts/pmc_mfma.sh'  then  python scripts/pmc_mfma.py  (here) -> profiles/<ROUND>_mfma_util.json
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mfma
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT -o m --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/run.log 2>&1
ls $OUT

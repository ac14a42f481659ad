#!/bin/bash
# A/B of the 3x3 stride-1 kernel forms per layer shape (diagnostic build on the box's scratch copy; never shipped):
# FP_HALO_FORM=0 band by LDS-DMA, 8 waves, 1 workgroup per CU; FORM=1 NPW=4 band through registers, 8 waves; FORM=1 NPW=2 4 waves x 2 workgroups per CU
cd "$GRAFT_REPO_ROOT"
make -C foundationpose_amd/csrc -B -j12 EXTRA=-DHALO_STAMP > gpurun_out/diag_build.log 2>&1 || exit 1
for cfg in "0 4" "1 4" "1 2"; do
  set -- $cfg
  echo "== FP_HALO_FORM=$1 FP_HALO_NPW=$2"
  FP_HALO_FORM=$1 FP_HALO_NPW=$2 REPS=50 timeout -k 10 120 python3 scripts/bench_conv.py encA_res_128 encAB_res_256 encAB_res_512 || exit 1
done

#!/bin/bash
# timing experiments of the Winograd kernel's main loop (diagnostic builds): FLAGS = ';'-separated -D sets, ENVS = ';'-separated env sets
cd "$GRAFT_REPO_ROOT"
IFS=';' read -ra SETS <<< "${FLAGS:--DWN_EXP=0}"
IFS=';' read -ra ES <<< "${ENVS:-FP_WINO_WAVES=8}"
for e in "${SETS[@]}"; do
  touch foundationpose_amd/csrc/conv_wino.hip
  make -C foundationpose_amd/csrc EXTRA="-DHALO_STAMP $e" -j16 > gpurun_out/exp_wino_make.log 2>&1 || { tail -5 gpurun_out/exp_wino_make.log; exit 1; }
  for v in "${ES[@]}"; do
    echo "== $e  $v"
    env $v N_HYP=${N_HYP:-252} timeout -k 10 120 python3 scripts/wino_stamps.py 2>&1 | grep "^C "
  done
done

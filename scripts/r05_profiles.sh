#!/bin/bash
# Every profile of the round in one box session (run on the GPU box: gpurun -- 'bash scripts/r05_profiles.sh'); summaries land under
# gpurun_out/, scripts/r05_collect.sh copies them into profiles/ here.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash scripts/prof_bench.sh > gpurun_out/r05_prof_bench.log 2>&1
python3 scripts/kernel_busy.py gpurun_out/prof_bench gpurun_out/r05_kernel_busy.json > gpurun_out/r05_kernel_busy.log 2>&1
bash scripts/pmc_traffic.sh > gpurun_out/r05_pmc_traffic.log 2>&1
bash scripts/pmc_mfma.sh > gpurun_out/r05_pmc_mfma.log 2>&1
bash scripts/prof_step.sh step > gpurun_out/r05_prof_step.log 2>&1
bash scripts/prof_shard.sh 8 shard32 > gpurun_out/r05_prof_shard32.log 2>&1
GRAPH=1 bash scripts/prof_track2.sh one track_one_graph > gpurun_out/r05_prof_track_one_graph.log 2>&1
GRAPH=1 bash scripts/prof_track2.sh multi track_multi_64_graph > gpurun_out/r05_prof_track_multi.log 2>&1
ls gpurun_out

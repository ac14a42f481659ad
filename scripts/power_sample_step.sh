#!/bin/bash
# Board power and shader clock while (a) the whole bench step, (b) the Linear 512->1024 layer alone, (c) the attention core alone
# run back to back - rocm-smi samples taken 10 s into each loop.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
sample() {
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Graphics Package Power|sclk" | tr -s ' \t' ' ' | sort -u | head -4
    sleep 1.0
  done
}
echo "== bench step (252 hypotheses, 5 refine + score), 600 steps"
python bench.py --steps 600 --warmup 2 --no-cpu-baseline > gpurun_out/power_step.log 2>&1 &
BP=$!
sleep 17; sample; wait $BP
python scripts/show_bench.py < gpurun_out/power_step.log | head -1
echo "== Linear 512->1024 alone"
REPS=120000 python scripts/bench_conv.py linear_512_1024 > gpurun_out/power_lin.log 2>&1 &
BP=$!
sleep 12; sample; wait $BP
tail -1 gpurun_out/power_lin.log
echo "== attention alone"
REPS=90000 python scripts/bench_attn.py > gpurun_out/power_attn.log 2>&1 &
BP=$!
sleep 12; sample; wait $BP
tail -1 gpurun_out/power_attn.log

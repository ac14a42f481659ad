#!/bin/bash
# A/B of an environment knob inside ONE box session: alternates bench steps with and without it.
#   usage: scripts/ab_env.sh "FP_PIPES=2" [rounds] [steps]
cd "$(dirname "$0")/.." || exit 1
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']
print('$1', 'ms/step %.2f (events off %.2f)' % (d['ms_per_step'], d['ms_per_step_events_off']), ' '.join('%s %.2f' % (c, k[c]['busy_ms_per_step']) for c in ('heads_wall','attention','linear','conv3x3_halo','render') if c in k))"; }
for i in $(seq 1 ${2:-2}); do
  python bench.py --no-cpu-baseline --no-extras --steps ${3:-20} | show "default   " || exit 1
  env $1 python bench.py --no-cpu-baseline --no-extras --steps ${3:-20} | show "$1" || exit 1
done

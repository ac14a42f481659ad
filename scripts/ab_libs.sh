#!/bin/bash
# A/B of two builds of the library inside ONE box session (boxes differ by several per cent): alternates bench steps with
# foundationpose_amd/lib/libfp_old.so and the current library.   usage: scripts/ab_libs.sh [rounds] [steps]
cd "$(dirname "$0")/.." || exit 1
L=foundationpose_amd/lib
cp $L/libfoundationpose_amd.so $L/libfp_new.so || exit 1
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']
print('$1', 'ms/step %.2f (events off %.2f)' % (d['ms_per_step'], d['ms_per_step_events_off']), ' '.join('%s %.2f' % (c, k[c]['ms_per_step']) for c in ('heads_wall','attention','linear','conv3x3_halo') if c in k))"; }
for i in $(seq 1 ${1:-2}); do
  cp $L/libfp_old.so $L/libfoundationpose_amd.so && python bench.py --no-cpu-baseline --no-extras --steps ${2:-20} | show old || exit 1
  cp $L/libfp_new.so $L/libfoundationpose_amd.so && python bench.py --no-cpu-baseline --no-extras --steps ${2:-20} | show new || exit 1
done

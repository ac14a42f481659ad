#!/bin/bash
# per-kernel times of the fused render at several batch sizes -> gpurun_out/prof_render/
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_render
rm -rf $OUT && mkdir -p $OUT
for N in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r$N -- python3 scripts/bench_render.py $N > $OUT/r$N.log 2>&1
  tail -1 $OUT/r$N.log
  python3 - <<PY
import csv, glob
f = glob.glob('$OUT/**/r${N}_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
  if any(k in r['Name'] for k in ('render_kernel', 'classify_faces', 'xform_vertices')):
    print('   %-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done

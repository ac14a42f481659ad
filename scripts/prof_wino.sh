#!/bin/bash
# per-dispatch durations of the direct and the Winograd 3x3 kernels at the network's layer shapes (rocprofv3 kernel trace)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_wino
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 scripts/bench_wino.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob('gpurun_out/prof_wino/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
  n = r['Kernel_Name']
  if 'conv3x3' in n or 'halo' in n or 'wino' in n:
    d[(n[:70], r['Grid_Size_X'], r.get('VGPR_Count', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
  v = sorted(v[2:]) if len(v) > 4 else sorted(v)
  print('%-72s grid %8s n %3d  min %7.1f  median %7.1f us' % (k[0], k[1], len(v), v[0], v[len(v) // 2]))
PY

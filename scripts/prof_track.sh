#!/bin/bash
# per-kernel durations inside a tracking frame (B = 1 and B = 64)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_track
mkdir -p $OUT
FRAMES=100 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 scripts/bench_track.py > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_track/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the first 110 frames after set-up are track_one eager (10 warm-up + 100): take dispatches by grid size is fragile - report by kernel name
d = collections.defaultdict(list)
for r in rows: d[(r['Kernel_Name'][:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
  print(f'{k[0]:70s} grid {k[1]:>8s} n {len(v):6d} total {sum(v)/1e3:8.1f} ms  median {sorted(v)[len(v)//2]:7.1f} us')
PY
tail -20 $OUT/run.log

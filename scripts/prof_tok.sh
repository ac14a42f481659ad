#!/bin/bash
# per-kernel durations of the token GEMM epilogues alone (rocprofv3 kernel trace) + in-kernel stamps (diagnostic build)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_tokalone
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 scripts/tok_stamps.py > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_tokalone/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
  if 'tok_gemm' in r['Kernel_Name']: d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()): print(k, 'n', len(v), 'min %.1f us  median %.1f us' % (min(v), sorted(v)[len(v) // 2]))
PY
make -C foundationpose_amd/csrc -B EXTRA=-DHALO_STAMP -j16 > $OUT/make.log 2>&1 && python3 scripts/tok_stamps.py

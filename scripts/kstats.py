"""Print per-kernel averages from a rocprofv3 --kernel-trace --stats directory: python scripts/kstats.py DIR [substring ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True)[0]
pats = sys.argv[2:]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
  if not pats or any(p in r['Name'] for p in pats):
    print('%-86s calls %6s  avg %9.1f us  total %8.2f ms  %5.1f %%' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6,
                                                                        100 * float(r['TotalDurationNs']) / tot))

"""Per-(kernel, grid) figures from a rocprofv3 --kernel-trace directory: python scripts/ktrace.py DIR [last_fraction]
Only the last `last_fraction` (default 0.6) of the dispatches, by count, are used (set-up and warm-up are left out).
Prints count, median and summed duration per (kernel name, grid), the union of all dispatch intervals (GPU busy) and the idle gaps."""
import csv, glob, sys, collections
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', ''))) for r in rows)
iv = iv[int(len(iv) * (1 - frac)):]
g = collections.defaultdict(list)
for s, e, n, grid in iv:
  g[(n[:72], grid)].append((e - s) / 1e3)
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, _, _ in iv:
  if cur_e is None or s > cur_e:
    if cur_e is not None:
      busy += cur_e - cur_s
      gaps.append((s - cur_e) / 1e3)
    cur_s, cur_e = s, e
  else:
    cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = (iv[-1][1] - iv[0][0]) / 1e6
tot = sum(sum(v) for v in g.values())
print(f'window {span:.2f} ms, {len(iv)} dispatches, busy (union) {busy / 1e6:.2f} ms, idle {span - busy / 1e6:.2f} ms in {len(gaps)} gaps '
      f'(median gap {sorted(gaps)[len(gaps) // 2] if gaps else 0:.1f} us), sum of spans {tot / 1e3:.2f} ms')
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
  print(f'{k[0]:72s} grid {k[1]:>8s} n {len(v):5d} median {sorted(v)[len(v) // 2]:8.1f} us  total {sum(v) / 1e3:8.2f} ms {100 * sum(v) / tot:5.1f} %')

"""GPU idle time per bench step from a rocprofv3 kernel trace: union of all dispatch intervals against wall time, and the kernels in
front of which the idle gaps sit.   python scripts/gpu_idle.py gpurun_out/prof_bench"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:56]) for r in csv.DictReader(open(f)))
rk = [s for s, e, n in iv if 'render_kernel<1' in n or 'render_kernelILi1' in n]       # six fused renders per step
n_steps = len(rk) // 6
lo, hi = 6 * min(10, n_steps // 3), 6 * (n_steps - 2)
t0, t1, steps = rk[lo], rk[hi], (hi - lo) / 6
busy, a, b, gaps = 0, None, None, []
for s, e, n in iv:
  if s < t0 or s >= t1:
    continue
  if a is None or s > b:
    if a is not None:
      busy += b - a
      gaps.append((s - b, n))
    a, b = s, e
  else:
    b = max(b, e)
busy += b - a
print(f'{steps:.0f} steps: wall {(t1 - t0) / steps / 1e6:.3f} ms/step, some kernel executing {busy / steps / 1e6:.3f} ms/step, idle {(t1 - t0 - busy) / steps / 1e6:.3f} ms/step in {len(gaps) / steps:.0f} gaps')
cnt, tot = collections.Counter(), collections.Counter()
for g, n in gaps:
  cnt[n] += 1
  tot[n] += g
for k, v in tot.most_common(12):
  print(f'  in front of {k:56s} {cnt[k] / steps:5.1f} per step  {v / steps / 1e3:6.1f} us/step  mean {v / cnt[k] / 1e3:5.1f} us')

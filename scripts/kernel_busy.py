"""Per kernel of a rocprofv3 --kernel-trace directory: calls, mean span of a dispatch, BUSY time (union of the dispatch intervals: the
time during which at least one dispatch of the kernel was executing) and the mean number of its dispatches in flight.  Kernels whose
launches overlap on two streams (the two half-batch trunks, the two RefineNet heads) have spans about twice their busy time per launch;
FLOPs / busy time is the rate the chip sustains on the kernel.
  python scripts/kernel_busy.py gpurun_out/prof_bench [out.json]"""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
iv = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
  iv[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
def union_ns(v):
  v = sorted(v)
  busy, a, b = 0, None, None
  for s, e in v:
    if a is None or s > b:
      if a is not None:
        busy += b - a
      a, b = s, e
    else:
      b = max(b, e)
  return busy + (b - a if a is not None else 0)


# kernel classes of bench.py (all template variants of a class together: their launches overlap each other too)
CLASSES = {'conv3x3_halo': ('conv3x3_halo_dma_kernel', 'conv3x3_s1_band_kernel'), 'conv3x3_s2': ('conv3x3_s2_kernel', 'conv_igemm2_kernelILi128ELi3'), 'conv7x7': 'stem7x7_kernel',
           'linear': ('tok_gemm_kernel', 'tok_qkv_kernel', 'head_mlp_kernel', 'head_mlp128_kernel'), 'attention': '16attention_kernel', 'render': ('render_kernel', 'xform_vertices', 'classify_faces')}
steps = sum(len(v) for k, v in iv.items() if 'render_kernel<1' in k or 'render_kernelILi1' in k) / 6.0
cls_rows = []
for cname, pats in CLASSES.items():
  pats = (pats,) if isinstance(pats, str) else pats
  v = [x for k, vv in iv.items() if any(p in k for p in pats) for x in vv]
  if v and steps:
    cls_rows.append(dict(kernel_class=cname, calls_per_step=len(v) / steps, span_ms_per_step=sum(e - s for s, e in v) / 1e6 / steps,
                         busy_ms_per_step=union_ns(v) / 1e6 / steps))
    print('class %-14s %6.1f launches/step  spans %7.2f ms/step  busy %7.2f ms/step' % (cname, cls_rows[-1]['calls_per_step'], cls_rows[-1]['span_ms_per_step'],
                                                                                       cls_rows[-1]['busy_ms_per_step']))
rows = []
for name, v in iv.items():
  v.sort()
  busy, a, b = 0, None, None
  for s, e in v:
    if a is None or s > b:
      if a is not None:
        busy += b - a
      a, b = s, e
    else:
      b = max(b, e)
  busy += b - a
  span = sum(e - s for s, e in v)
  rows.append(dict(kernel=name.split('(')[0][:100], calls=len(v), mean_span_us=span / len(v) / 1e3, busy_ms=busy / 1e6, busy_us_per_call=busy / len(v) / 1e3,
                   in_flight=span / busy))
rows.sort(key=lambda r: -r['busy_ms'])
for r in rows[:30]:
  print('%-90s calls %5d  span %8.1f us  busy/call %8.1f us  busy %8.2f ms  in flight %.2f' % (r['kernel'][:90], r['calls'], r['mean_span_us'], r['busy_us_per_call'],
                                                                                                  r['busy_ms'], r['in_flight']))
if len(sys.argv) > 2:
  json.dump({'steps_in_trace': steps, 'classes': cls_rows, 'kernels': rows}, open(sys.argv[2], 'w'), indent=1)

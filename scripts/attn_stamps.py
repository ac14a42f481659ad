"""Diagnostic (needs `make -B EXTRA=-DHALO_STAMP`): where a wave of the (persistent) attention kernel spends its cycles.
Stamps: [0] entry, [31] exit, and of the workgroup's second item [1] start, per key block {barrier passed, S done, softmax done,
PV done}, [2] stores issued."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ['REPS'] = '3'
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_attn.py')).read())
buf = np.zeros((1024, 7, 32), dtype=np.uint64)
lib().fp_dbg_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p))
b = buf.astype(np.int64)
ok = b[:, :, 31] > 0
n_items = 2 * 4 * B
per_wg = n_items / min(n_items, 256)
for name, sel in (('waves 0-3', slice(0, 4)), ('waves 4-6', slice(4, 7))):
  w = b[:, sel][ok[:, sel]]
  life = w[:, 31] - w[:, 0]
  print(f'{name}: workgroup life {life.mean():.0f} cycles = {life.mean() / per_wg:.0f} per item ({per_wg:.2f} items); second item: {np.mean(w[:, 2] - w[:, 1]):.0f}')
  for kb in range(7):
    s0 = w[:, 3 + 4 * kb]
    prev = w[:, 1] if kb == 0 else w[:, 6 + 4 * (kb - 1)]
    print(f'   block {kb}: wait+barrier {np.mean(s0 - prev):6.0f}  stage+S {np.mean(w[:, 4 + 4 * kb] - s0):6.0f}  softmax {np.mean(w[:, 5 + 4 * kb] - w[:, 4 + 4 * kb]):6.0f}  PV {np.mean(w[:, 6 + 4 * kb] - w[:, 5 + 4 * kb]):6.0f}')
  print(f'   epilogue (normalise + stores issued) {np.mean(w[:, 2] - w[:, 30]):.0f}')

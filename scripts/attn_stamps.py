"""Diagnostic (needs `make -B EXTRA=-DHALO_STAMP`): where a wave of the attention kernel spends its cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ['REPS'] = '3'
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_attn.py')).read())
buf = np.zeros((1024, 7, 32), dtype=np.uint64)
lib().fp_dbg_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p))
b = buf.astype(np.int64)
ok = b[:, :, 31] > 0
for name, sel in (('early waves 0-3', slice(0, 4)), ('late waves 4-6', slice(4, 7))):
  w = b[:, sel][ok[:, sel]]
  life = w[:, 31] - w[:, 0]
  print(f'{name}: life {life.mean():.0f} cycles; Q load {np.mean(w[:, 1] - w[:, 0]):.0f}; prologue DMA issue {np.mean(w[:, 2] - w[:, 1]):.0f}')
  for kb in range(7):
    s0 = w[:, 3 + 4 * kb]
    prev = w[:, 2] if kb == 0 else w[:, 6 + 4 * (kb - 1)]
    print(f'   block {kb}: wait+barrier {np.mean(s0 - prev):6.0f}  stage+(late PV)+S {np.mean(w[:, 4 + 4 * kb] - s0):6.0f}  softmax {np.mean(w[:, 5 + 4 * kb] - w[:, 4 + 4 * kb]):6.0f}  PV {np.mean(w[:, 6 + 4 * kb] - w[:, 5 + 4 * kb]):6.0f}')
  print(f'   after loop -> exit stamp {np.mean(w[:, 31] - w[:, 30]):.0f}')

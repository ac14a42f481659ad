"""Diagnostic (needs `make -B EXTRA=-DHALO_STAMP`): per-wave cycle counts of conv_igemm2 (prologue / K loop / epilogue) for the
Linear layer shapes; mean over the first workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr

ctx = _lib.Context.get('cuda:0')
L = lib()
for name, (M, K, N, use_res) in {'linear 512->512 +res': (100800, 512, 512, True), 'linear 512->512': (100800, 512, 512, False),
                                  'linear 512->1024': (100800, 512, 1024, False)}.items():
  g = torch.Generator(device='cuda').manual_seed(0)
  x = torch.randn((M, 1, 1, K), device='cuda', generator=g).half()
  w = (torch.randn((N, K), device='cuda', generator=g) * 0.05).half()
  b = torch.randn((N,), device='cuda', generator=g) * 0.1
  res = torch.randn((M, 1, 1, N), device='cuda', generator=g).half() if use_res else None
  out = torch.empty((M, 1, 1, N), device='cuda', dtype=torch.float16)
  run = lambda: check(L.fp_conv2d_f16(ctx.handle, ptr(x), M, 1, 1, K, ptr(w), ptr(b), N, 1, 1, 1, 0, ptr(res), 0, ptr(out), 0, stream_ptr()))
  for _ in range(3): run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): run()
  e1.record(); torch.cuda.synchronize()
  buf = np.zeros((4096, 4, 4), dtype=np.uint64)
  L.fp_dbg_igemm_stamps(buf.ctypes.data_as(ctypes.c_void_p))
  v = buf[:400, [0, 2, 3]].reshape(-1, 4).astype(np.float64)
  v = v[v[:, 3] > 0]
  m = v.mean(0)
  print(f'{name:22s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   prologue {m[0]:7.0f}  K loop {m[1]:7.0f}  epilogue {m[2]:7.0f} cycles')
  # per-CU timelines from wave 1 of every workgroup: how many workgroups share a CU at a time, gaps between them
  w1 = buf[:, 1]
  w1 = w1[w1[:, 3] > 0]
  t0, t1, hw = w1[:, 0].astype(np.int64), w1[:, 1].astype(np.int64), w1[:, 2]
  cu = ((hw >> 32) & 0xf) * 4096 + (hw & 0xffffffff & 0xff00) // 256 * 1 + (((hw >> 13) & 7) * 64) + (((hw >> 12) & 1) * 32)   # xcc, cu_id, se_id, sh_id
  span = t1.max() - t0.min()
  life = (t1 - t0)
  print(f'   {len(w1)} workgroups on {len(np.unique(cu))} CUs; workgroup life mean {life.mean():.0f} (min {life.min()}, max {life.max()})')
  spans = np.array([t1[cu == c].max() - t0[cu == c].min() for c in np.unique(cu)], dtype=np.float64)
  us = e0.elapsed_time(e1) / 20 * 1e3
  print(f'   per-CU span (first entry -> last exit) median {np.median(spans):.0f} ticks over {us:.1f} us -> {np.median(spans) / us / 1e3:.2f} GHz if a tick is a shader cycle')
  conc, gaps = [], []
  for c in np.unique(cu)[:64]:
    sel = cu == c
    ev = sorted([(a, 1) for a in t0[sel]] + [(b, -1) for b in t1[sel]])
    cur, last, busy2, busy1, idle = 0, ev[0][0], 0, 0, 0
    for t, d in ev:
      dt = t - last
      if cur >= 2: busy2 += dt
      elif cur == 1: busy1 += dt
      else: idle += dt
      cur += d; last = t
    conc.append((busy2, busy1, idle, sel.sum()))
  conc = np.array(conc, dtype=np.float64)
  tot = conc[:, :3].sum(1)
  print(f'   per CU (first 64): {conc[:, 3].mean():.1f} workgroups; time with >=2 resident {100 * (conc[:, 0] / tot).mean():.0f} %, 1 resident {100 * (conc[:, 1] / tot).mean():.0f} %, none {100 * (conc[:, 2] / tot).mean():.0f} %')

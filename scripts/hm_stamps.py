"""Diagnostic (needs `make -B EXTRA=-DHM_STAMP`): per-wave cycle counts of head_mlp128_kernel's phases at M = 252 x 400 tokens
(out-projection K loop through the ring / LayerNorm1 epilogue / linear1 K loop / ff hand-over / linear2 K loop / LayerNorm2 sums)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr

ctx = _lib.Context.get('cuda:0')
L = lib()
n_hyp = int(os.environ.get('NHYP', '252'))
M = n_hyp * 400
g = torch.Generator(device='cuda').manual_seed(0)
att = torch.randn((M, 512), device='cuda', generator=g).half()
tok = torch.randn((M, 512), device='cuda', generator=g).half()
mk = lambda s: (torch.randn((512, 512), generator=torch.Generator().manual_seed(s)) * 0.05).numpy()
w = [mk(1), mk(2), mk(3)]
b = np.zeros(512, np.float32); gam = np.ones(512, np.float32)
out = torch.empty((M // 16, 512), device='cuda', dtype=torch.float32)
run = lambda: check(L.fp_head_mlp_f16(ctx.handle, ptr(att), ptr(tok), M, ptr(w[0]), ptr(b), ptr(gam), ptr(b), ptr(w[1]), ptr(b), ptr(w[2]), ptr(b), ptr(out), stream_ptr()))
for _ in range(3): run()
torch.cuda.synchronize()
if hasattr(L, 'fp_dbg_hm_stamps'):
  buf = np.zeros((1024, 8, 8), dtype=np.uint64)
  L.fp_dbg_hm_stamps(buf.ctypes.data_as(ctypes.c_void_p))
  n = min(1024, (M + 127) // 128)
  v = buf[:n].astype(np.float64)
  names = ['K1 (ring)', 'LN1 epi', 'K2', 'ff hand-over', 'K3', 'LN2 sums']
  d = np.stack([v[..., i + 1] - v[..., i] for i in range(6)], -1).reshape(-1, 6)
  print(f'M = {M}: ' + '  '.join(f'{nm} {x:7.0f}' for nm, x in zip(names, d.mean(0))) + f'  total {d.sum(1).mean():.0f} cycles per wave (mean over {n} workgroups)')
  first = v[:256]
  print('  first round: start spread %.0f cycles, end - start of a workgroup %.0f' % (first[..., 0].max() - first[..., 0].min(), (first[..., 6] - first[..., 0]).mean()))

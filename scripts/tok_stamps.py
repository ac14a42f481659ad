"""Diagnostic (needs `make -B EXTRA=-DHALO_STAMP`): per-wave cycle counts of tok_gemm_kernel (tile load / K loop / epilogue)
for the four epilogues at M = 252 x 400 tokens; wall time per launch from HIP events."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr

ctx = _lib.Context.get('cuda:0')
L = lib()
stamped = hasattr(L, 'fp_dbg_tok_stamps')
M = 252 * 400
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((M, 512), device='cuda', generator=g).half()
res = torch.randn((M, 512), device='cuda', generator=g).half()
w = (torch.randn((512, 512), generator=torch.Generator().manual_seed(1)) * 0.05).numpy()
b = np.zeros(512, np.float32); gam = np.ones(512, np.float32)
outs = {0: torch.empty((M, 512), device='cuda', dtype=torch.float16), 1: torch.empty((252, 4, 128, 416), device='cuda', dtype=torch.float16),
        2: torch.empty((M, 512), device='cuda', dtype=torch.float16), 3: torch.empty((M // 16, 512), device='cuda', dtype=torch.float32)}
for epi, name in ((0, 'rows'), (1, 'V^T image'), (2, 'residual + LN rows'), (3, 'residual + LN group sums')):
  run = lambda: check(L.fp_token_linear_f16(ctx.handle, ptr(x), M, ptr(w), ptr(b), epi, 0, ptr(res) if epi >= 2 else None, ptr(gam) if epi == 2 else None,
                                            ptr(b) if epi == 2 else None, 400, ptr(outs[epi]), stream_ptr()))
  for _ in range(3): run()
  msg = f'{name:26s}'
  if stamped:
    buf = np.zeros((2048, 8, 4), dtype=np.uint64)
    L.fp_dbg_tok_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    v = buf[:700].astype(np.float64)
    d = np.stack([v[..., 1] - v[..., 0], v[..., 2] - v[..., 1], v[..., 3] - v[..., 2]], -1).reshape(-1, 3)
    msg += f'  tile load {d[:, 0].mean():7.0f}  K loop {d[:, 1].mean():7.0f}  epilogue {d[:, 2].mean():7.0f} cycles (mean per wave; {d.sum(1).mean():.0f} total)'
  print(msg, flush=True)

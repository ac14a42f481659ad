"""Diagnostic (needs `make -B EXTRA=-DHALO_STAMP`): in-kernel cycle sums of conv3x3_wino_kernel per wave: [vmcnt wait + barrier B_g],
[first k-step of a group], [second k-step], whole loop, prologue, epilogue; mean over the first workgroups of one launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
ctx = _lib.Context.get('cuda:0')
L = lib()
nh = int(os.environ.get('N_HYP', '252'))
for C, HW, N in ((128, 40, 2 * nh), (256, 40, nh), (512, 20, nh)):
  g = torch.Generator(device='cuda').manual_seed(0)
  x = torch.randn((N, HW, HW, C), device='cuda', generator=g).half().relu()
  w32 = (torch.randn((C, C, 3, 3), generator=torch.Generator().manual_seed(1)) * (2.0 / (9 * C)) ** 0.5).contiguous()
  b = torch.randn((C,), device='cuda', generator=g) * 0.1
  res = torch.randn((N, HW, HW, C), device='cuda', generator=g).half()
  out = torch.empty((N, HW, HW, C), device='cuda', dtype=torch.float16)
  run = lambda: check(L.fp_conv3x3_wino_f16(ctx.handle, ptr(x), N, HW, C, C, w32.data_ptr(), ptr(b), ptr(res), 1, ptr(out), stream_ptr()))
  for _ in range(3): run()
  L.fp_dbg_wino_stamps(None, 1)
  run()
  buf = np.zeros((4096, 8, 8), dtype=np.uint64)
  L.fp_dbg_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0)
  v = buf[:1024].astype(np.float64).reshape(-1, 8)
  v = v[v[:, 3] > 0]
  groups = 3 * C // 32
  m = v.mean(0)
  life = (v[:, 7] - v[:, 6]) * 10.0
  print(f'C {C} HW {HW}: per group: wait+barrier {m[0] / groups:6.0f}  k-step 0 {m[1] / groups:6.0f}  k-step 1 {m[2] / groups:6.0f} (8 waves: 8 MFMAs = 256 pipe cycles per wave and k-step, two waves per SIMD; 4 waves: 16 = 512) | '
        f'prologue {m[4]:6.0f} loop {m[3]:7.0f} epilogue {m[5]:6.0f} cycles | lifetime {life.mean() / 1e3:5.1f} us -> {(m[3] + m[4] + m[5]) / life.mean():.2f} GHz', flush=True)

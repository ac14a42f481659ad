"""3x3 stride-1 layer shapes at 252 hypotheses: the direct kernels (conv_halo / conv_s1b) and the Winograd form (conv_wino), N_REP back-to-back
launches each; run under `rocprofv3 --kernel-trace` (scripts/prof_wino.sh reads the per-dispatch durations: the Winograd building block
packs its weights on the host per call, so host timing would not mean anything)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
ctx = _lib.Context.get('cuda:0')
reps = int(os.environ.get('N_REP', '12'))
nh = int(os.environ.get('N_HYP', '252'))
for C, HW, N in ((128, 40, 2 * nh), (256, 40, nh), (512, 20, nh)):
  g = torch.Generator(device='cuda').manual_seed(0)
  x = torch.randn((N, HW, HW, C), device='cuda', generator=g).half().relu()
  w32 = (torch.randn((C, C, 3, 3), generator=torch.Generator().manual_seed(1)) * (2.0 / (9 * C)) ** 0.5).contiguous()
  w = w32.permute(0, 2, 3, 1).reshape(C, 9 * C).half().cuda().contiguous()
  b = torch.randn((C,), device='cuda', generator=g) * 0.1
  res = torch.randn((N, HW, HW, C), device='cuda', generator=g).half()
  out = torch.empty((N, HW, HW, C), device='cuda', dtype=torch.float16)
  for _ in range(reps):
    check(lib().fp_conv2d_f16(ctx.handle, ptr(x), N, HW, HW, C, ptr(w), ptr(b), C, 3, 3, 1, 1, ptr(res), 1, ptr(out), 0, stream_ptr()))
  torch.cuda.synchronize()
  for _ in range(reps):
    check(lib().fp_conv3x3_wino_f16(ctx.handle, ptr(x), N, HW, C, C, w32.data_ptr(), ptr(b), ptr(res), 1, ptr(out), stream_ptr()))
  torch.cuda.synchronize()
  print('done', C, HW, N, flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from foundationpose_amd._lib import check, lib, ptr, stream_ptr, Context
from tests.test_gpu_kernels import _pack_conv_weight
ctx = Context.get(torch.device('cuda', 0))
for C, HW, sizes in [(512, 20, (3, 32, 50, 70, 40, 20)), (256, 40, (2, 12, 63, 20)), (128, 40, (3, 40, 24, 63))]:
  g = torch.Generator(device='cuda').manual_seed(77 + C)
  nmax = max(sizes)
  x = torch.randn((nmax, HW, HW, C), device='cuda', generator=g).half().relu()
  w = (torch.randn((C, C, 3, 3), device='cuda', generator=g) * (2.0 / (C * 9)) ** 0.5).half()
  b = torch.randn((C,), device='cuda', generator=g) * 0.1
  res = torch.randn((nmax, HW, HW, C), device='cuda', generator=g).half()
  wp = _pack_conv_weight(w.float().cpu(), C).cuda()
  ref = torch.relu(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1) + res.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
  outs = {}
  for n in sizes:
    out = torch.full((n, HW, HW, C), float('nan'), dtype=torch.float16, device='cuda')
    check(lib().fp_conv2d_f16(ctx.handle, ptr(x), n, HW, HW, C, ptr(wp), ptr(b), C, 3, 3, 1, 1, ptr(res), 1, ptr(out), 0, stream_ptr()))
    torch.cuda.synchronize()
    outs[n] = out
    out2 = torch.full((n, HW, HW, C), float('nan'), dtype=torch.float16, device='cuda')
    check(lib().fp_conv2d_f16(ctx.handle, ptr(x), n, HW, HW, C, ptr(wp), ptr(b), C, 3, 3, 1, 1, ptr(res), 1, ptr(out2), 0, stream_ptr()))
    torch.cuda.synchronize()
    print('   rerun differs in', int((out2 != out).sum()), 'elements; bitwise', int((out2.view(torch.int16) != out.view(torch.int16)).sum()))
    err = (out.float() - ref[:n]).abs().reshape(n * HW * HW, C)
    bad = (err > 2e-2).any(1).nonzero().flatten()
    badc = (err > 2e-2).any(0).nonzero().flatten()
    print(f'C {C} HW {HW} n {n}: max err {float(err.max()):.4f} bad pixels {bad.numel()}', (bad[:8].tolist(), bad[-4:].tolist(), badc[:4].tolist(), badc[-2:].tolist()) if bad.numel() else '')

  ns = sorted(sizes)
  for a in ns:
    for bb in ns:
      if a < bb:
        d = (outs[a].view(torch.int16) != outs[bb][:a].view(torch.int16)).reshape(a * HW * HW, C)
        px = d.any(1).nonzero().flatten()
        if px.numel():
          ch = d.any(0).nonzero().flatten()
          q = int(px[0])
          print(f'  {a} vs {bb}: {px.numel()} pixels differ, first {px[:6].tolist()} last {px[-3:].tolist()} channels {ch[:4].tolist()}..{ch[-2:].tolist()} ({ch.numel()}); e.g. {outs[a].reshape(-1, C)[q][d[q]][:3].tolist()} vs {outs[bb][:a].reshape(-1, C)[q][d[q]][:3].tolist()}')

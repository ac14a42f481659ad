"""Layer shapes of small shards alone, 200 back-to-back launches: time per launch for the tile shape FP_HALO_NT forces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
ctx = _lib.Context.get('cuda:0')
reps = 200
for C, HW, Ns in ((512, 20, (16, 32, 63, 64)), (256, 40, (16, 32, 63, 64)), (128, 40, (32, 64, 126))):
  for N in Ns:
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn((N, HW, HW, C), device='cuda', generator=g).half().relu()
    w = (torch.randn((C, 9 * C), device='cuda', generator=g) * (2.0 / (9 * C)) ** 0.5).half()
    b = torch.randn((C,), device='cuda', generator=g) * 0.1
    out = torch.empty((N, HW, HW, C), device='cuda', dtype=torch.float16)
    run = lambda: check(lib().fp_conv2d_f16(ctx.handle, ptr(x), N, HW, HW, C, ptr(w), ptr(b), C, 3, 3, 1, 1, None, 1, ptr(out), 0, stream_ptr()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    M = N * HW * HW
    print(f'NT={os.environ.get("FP_HALO_NT", "auto"):4s} C {C} N {N:3d}: {us:7.1f} us  {2.0 * M * C * 9 * C / us / 1e6:7.1f} TF/s', flush=True)

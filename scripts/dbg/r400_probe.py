"""Is a 400-pixel tile shorter than a 512-pixel one?  C = 256, 40x40 maps, 200 / 256 workgroups in one round, back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
ctx = _lib.Context.get('cuda:0')
reps = int(os.environ.get('REPS', '200'))
for C, HW in ((256, 40), (512, 20), (128, 40)):
  n_ct = C // 128
  for N in ({256: (25, 32, 16), 512: (50, 64, 63), 128: (50, 64, 32)}[C]):
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
    print(f'C {C} N {N:3d}: {us:7.1f} us   tiles512 {((M + 511) // 512) * n_ct:4d}  tiles400 {(M // 400) * n_ct:4d}   {2.0 * M * C * 9 * C / us / 1e6:7.1f} TF/s   R400={os.environ.get("FP_HALO_R400", "1")}')

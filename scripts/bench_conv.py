"""Micro-benchmark of the convolution kernels at the network's layer shapes (252 hypotheses)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr

SHAPES = {  # name: (N, H, W, Cin, Cout, k, stride, res)
  'encA_res_128': (504, 40, 40, 128, 128, 3, 1, True),
  'encAB_res_256': (252, 40, 40, 256, 256, 3, 1, True),
  'encAB_res_512': (252, 20, 20, 512, 512, 3, 1, True),
  'down_256_512': (252, 40, 40, 256, 512, 3, 2, False),
  'down_64_128': (504, 80, 80, 64, 128, 3, 2, False),
  'stem_7x7': (504, 160, 160, 8, 64, 7, 2, False),
  'linear_512_1024': (1, 1, 100800, 512, 1024, 1, 1, False),
  'linear_512_512': (1, 1, 100800, 512, 512, 1, 1, True),
  'trk_128': (1, 40, 40, 128, 128, 3, 1, True),        # tracking, one hypothesis: a side's encodeA layer (13 tiles of 128 px)
  'trk_256': (1, 40, 40, 256, 256, 3, 1, True),        # ... encodeAB at 40x40 (split-K)
  'trk_512': (1, 20, 20, 512, 512, 3, 1, True),        # ... encodeAB at 20x20 (split-K)
  'trk2_128': (2, 40, 40, 128, 128, 3, 1, True), 'trk2_256': (2, 40, 40, 256, 256, 3, 1, True), 'trk2_512': (2, 20, 20, 512, 512, 3, 1, True),
  'trk4_128': (4, 40, 40, 128, 128, 3, 1, True), 'trk4_256': (4, 40, 40, 256, 256, 3, 1, True), 'trk4_512': (4, 20, 20, 512, 512, 3, 1, True),
  'trk8_128': (8, 40, 40, 128, 128, 3, 1, True), 'trk8_256': (8, 40, 40, 256, 256, 3, 1, True), 'trk8_512': (8, 20, 20, 512, 512, 3, 1, True),
}

def main():
  names = sys.argv[1:] or list(SHAPES)
  reps = int(os.environ.get('REPS', '10'))
  ctx = _lib.Context.get('cuda:0')
  for name in names:
    N, H, W, Cin, Cout, k, stride, use_res = SHAPES[name]
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    g = torch.Generator(device='cuda').manual_seed(0)
    x = (torch.randn((N, H, W, Cin), device='cuda', generator=g)).half()
    kpad = (k * k * Cin + 31) // 32 * 32
    w = (torch.randn((Cout, kpad), device='cuda', generator=g) * (2.0 / (k * k * Cin)) ** 0.5).half()
    b = torch.randn((Cout,), device='cuda', generator=g) * 0.1
    res = torch.randn((N, Ho, Wo, Cout), device='cuda', generator=g).half() if use_res else None
    out = torch.empty((N, Ho, Wo, Cout), device='cuda', dtype=torch.float16)
    def run():
      check(lib().fp_conv2d_f16(ctx.handle, ptr(x), N, H, W, Cin, ptr(w), ptr(b), Cout, k, k, stride, pad, ptr(res), 1, ptr(out), 0, stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    cin_eff = 6 if Cin == 8 else Cin
    flops = 2.0 * N * Ho * Wo * Cout * k * k * cin_eff
    print(f'{name:18s} {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s')

if __name__ == '__main__':
  main()

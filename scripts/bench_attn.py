"""Attention core alone: 252 hypotheses x 400 tokens x 4 heads (82.6 GFLOP), back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
ctx = _lib.Context.get('cuda:0')
B, T = 252, 400
g = torch.Generator(device='cuda').manual_seed(0)
qk = (torch.randn((B * T, 1024), device='cuda', generator=g) * 1.5).half()
vt = torch.zeros((B, 4, 128, 416), device='cuda', dtype=torch.float16)
vt[..., :T] = torch.randn((B, 4, 128, T), device='cuda', generator=g).half()
out = torch.empty((B * T, 512), device='cuda', dtype=torch.float16)
run = lambda: check(lib().fp_attention_f16(ctx.handle, ptr(qk), ptr(vt), B, T, ptr(out), stream_ptr()))
for _ in range(5): run()
torch.cuda.synchronize()
reps = int(os.environ.get('REPS', 50))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f'attention 252x400x4x128  {us:7.1f} us  {4.0 * B * 4 * T * T * 128 / us / 1e6:6.1f} TF/s')

"""The fused render (vertex pre-pass + face binning + strip rasteriser -> network-ready fp16 crops) alone, at the batch sizes of
a 1 / 2 / 4 / 8-GPU job and of tracking; under rocprofv3 --kernel-trace --stats the three kernels show separately.
usage: python scripts/bench_render.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, k_ptr, lib, ptr, stream_ptr

dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
ob = objects[0]
ctx = est.refiner.ctx
dm = _lib.device_mesh(ctx, est.mesh_tensors)
Kd, Kp = k_ptr(ob['K'])
reps = int(os.environ.get('REPS', 20))
for N in [int(a) for a in sys.argv[1:]] or [252, 126, 63, 32, 1]:
  poses = ob['poses'][:N].contiguous()
  tf = torch.empty((N, 3, 3), device=dev)
  bbox = torch.empty((N, 4), device=dev)
  net = torch.empty((N, 160, 160, 8), device=dev, dtype=torch.float16)
  check(lib().fp_crop_window_tf(ctx.handle, ptr(poses), N, Kp, 1.2, float(est.diameter), 160, 160, ptr(tf), ptr(bbox), stream_ptr(dev)))
  run = lambda: check(lib().fp_render_net(ctx.handle, dm.handle, ptr(poses), N, Kp, 480, 640, ptr(bbox), 160, 160, float(est.diameter), 1, 0.001, ptr(net),
                                          stream_ptr(dev)))
  for _ in range(3):
    run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    run()
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / reps * 1e3
  if os.environ.get('FLUSH'):
    # every launch behind a 96-MB fill (the L2s hold none of the mesh, as inside a frame behind 33 MB of weights per pass); events around the render only
    junk = torch.empty((96 << 20,), dtype=torch.uint8, device=dev)
    tot = 0.0
    for _ in range(reps):
      junk.fill_(1)
      e0.record(); run(); e1.record()
      torch.cuda.synchronize()
      tot += e0.elapsed_time(e1)
    print(f'render N={N:4d}: {tot / reps * 1e3:8.1f} us per launch behind a cache-flushing fill')
  cov = float((net[..., 5] != 0).float().mean())
  print(f'render N={N:4d}: {us:8.1f} us per launch triple  ({N * 160 * 160 * 16 / us / 1e6:6.2f} TB/s written; {cov * 100:.0f} % of the pixels covered)')

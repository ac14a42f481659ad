import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from foundationpose_amd import synthetic as S
from foundationpose_amd.Utils import nvdiffrast_render
from foundationpose_amd.synthetic import trajectory
dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(64)
n = 200
K = S.YCB_K
poses = torch.as_tensor(trajectory(n), device=dev)
c, d, _ = nvdiffrast_render(K=K, H=480, W=640, ob_in_cams=poses, mesh_tensors=est.mesh_tensors, use_light=True)
rgbs = (c * 255).clamp(0, 255).to(torch.uint8)
depths = torch.where(d > 0, d, torch.full_like(d, 1.2))
for mode in ('one', 'multi'):
  est.pose_last = poses[0].clone()
  for f in range(n):
    if mode == 'one': est.track_one(rgbs[f], depths[f], K, iteration=2)
    else: est.track_multi(rgbs[f], depths[f], K, iteration=2, n_hypotheses=64)
    if f in (0, 1, 2, 5, 10, 20, 50, 100, 199):
      dt = float((est.pose_last[:3, 3] - poses[f][:3, 3]).norm())
      print(mode, 'frame', f, 'translation error vs trajectory %.4f m' % dt, 'z %.3f' % float(est.pose_last[2, 3]))

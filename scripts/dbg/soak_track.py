"""Soak: N tracking frames (teacher-forced start poses), eager and as a hipGraph replay, one and 64 hypotheses: every frame's pose must be
bit-identical between the two modes and between two eager passes (rare races in the small fused kernels would show as a differing frame)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from foundationpose_amd import synthetic as S
from foundationpose_amd.Utils import nvdiffrast_render
from foundationpose_amd.synthetic import trajectory

dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(64)
n = int(os.environ.get('FRAMES', '300'))
n_hyp = int(os.environ.get('HYP', '64'))
est.refiner.ctx.reserve(max(64, n_hyp))
K = S.YCB_K
poses = torch.as_tensor(trajectory(n), device=dev)
rgbs, depths = [], []
for s0 in range(0, n, 50):
  c, d, _ = nvdiffrast_render(K=K, H=480, W=640, ob_in_cams=poses[s0:s0 + 50], mesh_tensors=est.mesh_tensors, use_light=True)
  rgbs.append((c * 255).clamp(0, 255).to(torch.uint8)); depths.append(torch.where(d > 0, d, torch.full_like(d, 1.2)))
rgbs, depths = torch.cat(rgbs), torch.cat(depths)
starts = [poses[max(f - 1, 0)].clone() for f in range(n)]
def run(mode, graph):
  est.enable_track_graph(graph)
  out = []
  for f in range(n):
    est.pose_last = starts[f]
    out.append(est.track_one(rgbs[f], depths[f], K, iteration=2) if mode == 'one' else est.track_multi(rgbs[f], depths[f], K, iteration=2, n_hypotheses=n_hyp))
  return np.stack(out)
for mode in ('one', 'multi'):
  a, b, c = run(mode, False), run(mode, False), run(mode, True)
  bad_e = int((a != b).any(axis=(1, 2)).sum()); bad_g = int((a != c).any(axis=(1, 2)).sum())
  print(f'{mode} ({1 if mode == "one" else n_hyp} hypotheses): {n} frames, eager vs eager differing frames {bad_e}, eager vs graph {bad_g}, finite {bool(np.isfinite(a).all())}')
  assert bad_e == 0 and bad_g == 0
print('soak ok')

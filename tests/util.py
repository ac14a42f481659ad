"""Shared builders for the parity tests (seeded synthetic scene, oracle handles)."""
import functools

import numpy as np
import torch

from foundationpose_amd import synthetic as S
from foundationpose_amd.mesh_tensors import make_mesh_tensors


@functools.lru_cache(maxsize=4)
def scene(seed=0, textured=False, n_theta=96, n_z=84):
  """Centred mesh, CPU mesh_tensors, diameter, rotation grid and one RGB-D frame rendered by the ORACLE."""
  from oracle import geometry as G
  from oracle.render import nvdiffrast_render as oracle_render
  mesh = S.make_mustard_mesh(seed=seed, n_theta=n_theta, n_z=n_z, textured=textured)
  center = (mesh.vertices.min(0) + mesh.vertices.max(0)) / 2
  mesh.vertices = mesh.vertices - center
  mt = make_mesh_tensors(mesh, device='cpu')

  def rf(K, H, W, pose):
    c, d, _ = oracle_render(K=K, H=H, W=W, ob_in_cams=pose, mesh_tensors=mt, use_light=True)
    return c[0].numpy(), d[0].numpy()
  sc = S.make_scene(rf, mt, seed=seed)
  diam = G.compute_mesh_diameter(mesh.vertices, 10000, np.random.RandomState(0))
  grid = G.make_rotation_grid()
  return dict(mesh=mesh, mt=mt, center=center, diameter=diam, grid=grid, **sc)


def hypotheses(sc, n, jitter_seed=None):
  """First n grid rotations at the guessed translation (float32 (n,4,4))."""
  from oracle import geometry as G
  d = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  c = G.guess_translation(d, sc['mask'], sc['K'])
  poses = sc['grid'][:n].copy()
  poses[:, :3, 3] = c.astype(np.float32)
  if jitter_seed is not None:
    rs = np.random.RandomState(jitter_seed)
    poses[:, :3, 3] += (rs.randn(n, 3) * 0.01).astype(np.float32)
  return poses.astype(np.float32)


def to_dev(mt):
  return {k: v.cuda() for k, v in mt.items()}


def mismatch_report(a, b, atol):
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  d = np.abs(a - b)
  bad = d > atol
  return float(bad.mean()), float(d.max()), float(np.median(d))


def net_inputs(seed, n):
  """(A, B) network inputs (n,6,160,160) float32 that DIFFER from sample to sample the way rendered / observed crops do:
  smooth colour and coordinate fields of seeded frequency and amplitude inside a seeded elliptic silhouette, exact zeros
  outside (masked background), a little pixel noise.  The across-sample spread of the network outputs is what the golden
  tests measure their tolerance against (a kernel that ignored or permuted its input must fail them)."""
  rs = np.random.RandomState(seed)
  vs, us = np.meshgrid(np.arange(160.0), np.arange(160.0), indexing='ij')
  sides = []
  for _ in range(2):
    imgs = np.zeros((n, 6, 160, 160), dtype=np.float32)
    for i in range(n):
      f, ph = rs.uniform(0.02, 0.3, (6, 2)), rs.uniform(0, 2 * np.pi, 6)
      base = np.stack([np.sin(us * f[c, 0] + vs * f[c, 1] + ph[c]) for c in range(6)])
      rgb = np.clip(0.5 + 0.45 * rs.uniform(0.3, 1.0) * base[:3] + rs.randn(3, 160, 160) * 0.04, 0, 1)
      xyz = rs.uniform(0.2, 1.2) * base[3:] + rs.randn(3, 160, 160) * 0.04
      cx, cy, rx, ry = rs.uniform(60, 100), rs.uniform(60, 100), rs.uniform(25, 75), rs.uniform(25, 75)
      inside = ((us - cx) / rx) ** 2 + ((vs - cy) / ry) ** 2 < 1
      xyz[:, ~inside] = 0
      rgb[:, ~inside] *= rs.uniform(0.0, 1.0)          # dimmed or black background
      imgs[i, :3], imgs[i, 3:] = rgb, xyz
    sides.append(torch.from_numpy(imgs))
  return sides


def nearest_pose_error(got, want):
  """max over `got` poses of the distance to the nearest `want` pose (max-abs over the 16 entries): compares two pose SETS whose
  order (a ranking by nearly tied scores) may differ."""
  got, want = np.asarray(got, dtype=np.float64).reshape(-1, 16), np.asarray(want, dtype=np.float64).reshape(-1, 16)
  return float(np.abs(got[:, None] - want[None]).max(-1).min(1).max())

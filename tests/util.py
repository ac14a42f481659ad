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

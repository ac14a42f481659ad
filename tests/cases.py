"""The parity cases behind tests/golden/fullsize.npz, defined ONCE for the fixture generator
(tests/golden/gen_fullsize.py, CPU oracle, build container) and for the GPU tests that read the fixture.

A case = one object in one RGB-D frame + a hypothesis set + the number of refinement iterations; the
expected values are the oracle's refined poses (hypothesis order), ScoreNet logits on those poses, the
argmax and the top-1 / top-2 logit margin.  Sizes follow BASELINE.json: c1 = configs[1] (252 hypotheses,
est_refine_iter=5), c3_* = the four objects of configs[3], c0_* = configs[0] (32 hypotheses), trk_* =
configs[4] (64-hypothesis tracking along a seeded smooth SE(3) trajectory)."""
import functools

import numpy as np

from tests import util

REFINE_SEED, SCORE_SEED = 0, 1


def _prelude(sc):
  """Depth filtering + back-projection + translation guess on the oracle (src/estimater.py:173-214)."""
  from oracle import geometry as G
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  return depth, G.depth2xyzmap(depth, sc['K'])


@functools.lru_cache(maxsize=None)
def smoke_scene():
  """The scene of __graft_entry__.smoke(): coarse mesh (2018 vertices), NOT pre-centred (the estimator centres it)."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.mesh_tensors import make_mesh_tensors
  from oracle import geometry as G
  from oracle.render import nvdiffrast_render as oracle_render
  mesh = S.make_mustard_mesh(seed=0, n_theta=48, n_z=42)
  center = (mesh.vertices.min(0) + mesh.vertices.max(0)) / 2
  centred = mesh.copy()
  centred.vertices = centred.vertices - center
  mt = make_mesh_tensors(centred, device='cpu')

  def rf(K, H, W, pose):
    c, d, _ = oracle_render(K=K, H=H, W=W, ob_in_cams=pose, mesh_tensors=mt, use_light=True)
    return c[0].numpy(), d[0].numpy()
  sc = S.make_scene(rf, mt, seed=0)
  diam = G.compute_mesh_diameter(centred.vertices, 10000, np.random.RandomState(0))
  return dict(mesh=mesh, mt=mt, center=center, diameter=diam, grid=G.make_rotation_grid(), **sc)


def case(name):
  """-> dict(sc, poses0 (n,4,4) f32, iteration, depth, xyz_map, refine_sd_kw, score_sd_kw, refine_cfg, score_cfg)."""
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  rkw, skw = dict(seed=REFINE_SEED), dict(seed=SCORE_SEED)
  rcfg, scfg = dict(REFINE_DEFAULT), dict(SCORE_DEFAULT)
  if name == 'c1':
    sc, n, it = util.scene(0), 252, 5
  elif name.startswith('c3_'):
    sc, n, it = util.scene(int(name[3:])), 252, 5
  elif name in ('c0_it1', 'c0_it2'):
    sc, n, it = util.scene(0), 32, int(name[-1])
  elif name == 'tex24':
    sc, n, it = util.scene(0, textured=True), 24, 1
  elif name == 'smoke8':
    sc, n, it = smoke_scene(), 8, 1
  else:
    raise KeyError(name)
  depth, xyz_map = _prelude(sc)
  if name == 'tex24':
    from oracle import geometry as G
    sym = np.stack([np.eye(4), np.diag([-1.0, -1.0, 1.0, 1.0])])
    grid = G.make_rotation_grid(symmetry_tfs=sym)
    poses0 = grid[:n].copy()
    poses0[:, :3, 3] = G.guess_translation(depth, sc['mask'], sc['K']).astype(np.float32)
    poses0 = poses0.astype(np.float32)
  else:
    poses0 = util.hypotheses(sc, n)
  return dict(sc=sc, poses0=poses0, iteration=it, depth=depth, xyz_map=xyz_map, refine_sd_kw=rkw, score_sd_kw=skw,
              refine_cfg=rcfg, score_cfg=scfg)


REGISTER_CASES = ['c1', 'c3_1', 'c3_2', 'c3_3', 'c0_it1', 'c0_it2', 'tex24', 'smoke8']


# ---- configs[4]: tracking along a seeded smooth SE(3) trajectory ---------------------------------------------------
def trajectory(n_frames, seed=0, t0=(0.02, -0.03, 0.75)):
  """n_frames object poses: per-frame increments are smooth (low-pass filtered seeded noise), at most 1 cm and
  2 degrees per frame."""
  from foundationpose_amd import synthetic as S
  rs = np.random.RandomState(seed + 4000)
  k = np.ones(25) / 25.0
  lin = np.stack([np.convolve(rs.randn(n_frames + 24), k, mode='valid') for _ in range(3)], 1)
  ang = np.stack([np.convolve(rs.randn(n_frames + 24), k, mode='valid') for _ in range(3)], 1)
  lin *= 0.004 / max(np.abs(lin).max(), 1e-9)            # <= 4 mm per axis per frame (< 1 cm in norm)
  ang *= np.deg2rad(1.0) / max(np.abs(ang).max(), 1e-9)  # <= 1 degree per axis per frame (< 2 degrees in norm)
  pose = np.eye(4)
  pose[:3, :3] = S.random_rotation(np.random.RandomState(seed + 1000))
  pose[:3, 3] = t0
  out = []
  for f in range(n_frames):
    w = ang[f]
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    dR = np.eye(3) + (np.sin(th) / max(th, 1e-12)) * Kx + ((1 - np.cos(th)) / max(th * th, 1e-12)) * (Kx @ Kx)
    pose = pose.copy()
    pose[:3, :3] = dR @ pose[:3, :3]
    pose[:3, 3] = pose[:3, 3] + lin[f]
    # keep the object in front of the camera and inside the frame
    pose[:3, 3] = np.clip(pose[:3, 3], [-0.12, -0.10, 0.55], [0.12, 0.10, 0.95])
    out.append(pose.astype(np.float32))
  return np.stack(out)


def tracking_frames(n_frames, seed=0):
  """RGB-D frames of the scene-0 object moving along trajectory(): list of dict(rgb, depth, K, gt_pose)."""
  from foundationpose_amd import synthetic as S
  from oracle.render import nvdiffrast_render as oracle_render
  sc = util.scene(0)

  def rf(K, H, W, pose):
    c, d, _ = oracle_render(K=K, H=H, W=W, ob_in_cams=pose, mesh_tensors=sc['mt'], use_light=True)
    return c[0].numpy(), d[0].numpy()
  frames = []
  for f, pose in enumerate(trajectory(n_frames, seed)):
    frames.append(S.make_scene(rf, sc['mt'], seed=seed + 100 + f, gt_pose=pose))
  return sc, frames

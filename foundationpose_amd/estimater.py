"""FoundationPose estimator on the MI355X hot path.

Drop-in for the reference orchestrator `src/estimater.py:18-268`: same constructor and method
signatures, same public attributes (`pose_last, poses, scores, best_id, diameter, mesh, mesh_tensors,
rot_grid, symmetry_tfs, model_center, glctx, scorer, refiner, ...`), same return types and the same
three degenerate-input behaviours (empty mask -> zero translation guess, <4 valid pixels -> identity
rotation + guessed translation as float64, track before register -> RuntimeError).  The compute is
the HIP library; with `dist_group` set the hypotheses are sharded over the ranks of one node
(foundationpose_amd/dist.py) - the reference is single-GPU only.
"""
import logging
import os

import numpy as np
import torch

from . import Utils as U
from .predict_pose_refine import PoseRefinePredictor
from .predict_score import ScorePredictor

_MIN_VALID_PIXELS = 4          # src/estimater.py:185
_MIN_DEPTH = 0.001


def _voxel_centroids(points, normals, voxel):
  """One centroid (and mean normal) per occupied voxel - what open3d's `voxel_down_sample` yields at
  src/estimater.py:59-60.  Only feeds `self.pts/normals/max_xyz/min_xyz`, which the hot path never reads."""
  points = np.asarray(points, dtype=np.float64)
  cell = np.floor((points - points.min(axis=0)) / voxel).astype(np.int64)
  _, owner, count = np.unique(cell, axis=0, return_inverse=True, return_counts=True)
  owner = owner.reshape(-1)
  centroid = np.zeros((len(count), 3))
  np.add.at(centroid, owner, points)
  centroid /= count[:, None]
  mean_n = None
  if normals is not None:
    mean_n = np.zeros((len(count), 3))
    np.add.at(mean_n, owner, np.asarray(normals, dtype=np.float64))
    mean_n /= np.maximum(np.linalg.norm(mean_n, axis=1, keepdims=True), 1e-12)
  return centroid, mean_n


class FoundationPose:
  def __init__(self, model_pts, model_normals, symmetry_tfs=None, mesh=None, scorer: ScorePredictor = None,
               refiner: PoseRefinePredictor = None, glctx=None, debug=0, debug_dir='/tmp/foundationpose_amd_debug/',
               dist_group=None):
    self.gt_pose = None
    self.ignore_normal_flip = True
    self.debug, self.debug_dir = debug, debug_dir
    os.makedirs(debug_dir, exist_ok=True)
    self.dist_group = dist_group
    self.reset_object(model_pts, model_normals, symmetry_tfs=symmetry_tfs, mesh=mesh)
    self.make_rotation_grid(min_n_views=40, inplane_step=60)
    self.glctx = glctx
    self.scorer = ScorePredictor() if scorer is None else scorer
    self.refiner = PoseRefinePredictor() if refiner is None else refiner
    self.pose_last = None          # pose of the centred mesh, kept for track_one

  # ------------------------------------------------------------------ object set-up (cold path)
  def reset_object(self, model_pts, model_normals, symmetry_tfs=None, mesh=None):
    """src/estimater.py:44-78: centre the mesh on its bounding-box centre, measure it, upload it."""
    lo, hi = mesh.vertices.min(axis=0), mesh.vertices.max(axis=0)
    self.model_center = (lo + hi) / 2
    self.mesh_ori = mesh.copy()
    centred = mesh.copy()
    centred.vertices = centred.vertices - self.model_center.reshape(1, 3)
    self.mesh = centred
    self.diameter = U.compute_mesh_diameter(model_pts=centred.vertices, n_sample=10000)
    self.vox_size = max(self.diameter / 20.0, 0.003)
    self.dist_bin, self.angle_bin = self.vox_size / 2, 20
    logging.info(f'self.diameter:{self.diameter}, vox_size:{self.vox_size}')
    pts, nrm = _voxel_centroids(centred.vertices, model_normals, self.vox_size)
    self.max_xyz, self.min_xyz = pts.max(axis=0), pts.min(axis=0)
    self.pts = torch.as_tensor(pts, dtype=torch.float32, device='cuda')
    self.normals = None if nrm is None else torch.as_tensor(nrm, dtype=torch.float32, device='cuda')
    # (the reference also writes the centred mesh to /tmp/<uuid>.obj here, src/estimater.py:69-70; nothing on the path reads that
    # file back, so none is written: no file is left behind per object)
    self.mesh_path = None
    self.mesh_tensors = U.make_mesh_tensors(centred)
    sym = torch.eye(4)[None] if symmetry_tfs is None else torch.as_tensor(symmetry_tfs)
    self.symmetry_tfs = sym.to(device='cuda', dtype=torch.float)
    # captured tracking graphs hold the addresses of the previous object's mesh and centring matrix: none survives a new object
    self._graphs = {}
    self._tf_centered_key = None
    logging.info("reset done")

  def get_tf_to_centered_mesh(self):
    tf = torch.eye(4, dtype=torch.float, device='cuda')
    tf[:3, 3] = -torch.as_tensor(self.model_center, device='cuda', dtype=torch.float)
    return tf

  def _tf_to_centered_cached(self):
    """get_tf_to_centered_mesh() built once per model centre (a tracking frame is ~1 ms: the three small launches that build the
    matrix are worth keeping out of it)."""
    key = np.asarray(self.model_center, dtype=np.float64).tobytes()
    if getattr(self, '_tf_centered_key', None) != key:
      self._tf_centered, self._tf_centered_key = self.get_tf_to_centered_mesh(), key
    return self._tf_centered

  def to_device(self, s='cuda:0'):
    for name, value in list(vars(self).items()):
      if torch.is_tensor(value):
        setattr(self, name, value.to(s))
    self.mesh_tensors = {k: v.to(s) for k, v in self.mesh_tensors.items()}
    if self.refiner is not None:        # src/estimater.py:97-100: the two networks move as well
      self.refiner.to_device(s)
    if self.scorer is not None:
      self.scorer.to_device(s)
    if self.glctx is not None:
      self.glctx = U.RasterizeContext(s)

  def make_rotation_grid(self, min_n_views=40, inplane_step=60):
    """src/estimater.py:106-124: 42 icosphere viewpoints x in-plane rotations, inverted to ob_in_cam,
    then greedy symmetry-aware clustering at 30 degrees (native: fp_cluster_poses)."""
    views = U.sample_views_icosphere(n_views=min_n_views)
    rolls = [U.euler_matrix(0, 0, a) for a in np.deg2rad(np.arange(0, 360, inplane_step))]
    grid = np.asarray([np.linalg.inv(cam_in_ob @ roll) for cam_in_ob in views for roll in rolls])
    logging.info(f"rot_grid:{grid.shape}")
    kept = np.asarray(U.cluster_poses(30, 99999, grid, self.symmetry_tfs.data.cpu().numpy()))
    self.rot_grid = torch.as_tensor(kept, device='cuda', dtype=torch.float)
    logging.info(f"self.rot_grid: {self.rot_grid.shape}")

  # ------------------------------------------------------------------ hypotheses
  def guess_translation(self, depth, mask, K, stats=None):
    """src/estimater.py:137-156: bounding-box centre of the mask back-projected at the median valid depth.  With a device
    depth image the reductions run on the device (`stats` = Utils.mask_depth_stats result; computed here when not given);
    numpy inputs take the host path of the reference."""
    if stats is None and torch.is_tensor(depth) and depth.is_cuda:
      stats = U.mask_depth_stats(depth, mask, _MIN_DEPTH)
    if stats is not None:
      if stats['n_mask'] == 0:
        logging.info('mask is all zero')
        return np.zeros((3))
      if stats['n_usable'] == 0:
        logging.info("valid is empty")
        return np.zeros((3))
      pixel = np.asarray([(stats['cmin'] + stats['cmax']) / 2.0, (stats['rmin'] + stats['rmax']) / 2.0, 1.0]).reshape(3, 1)
      return ((np.linalg.inv(K) @ pixel) * stats['median']).reshape(3)
    rows, cols = np.nonzero(mask > 0)
    if len(cols) == 0:
      logging.info('mask is all zero')
      return np.zeros((3))
    usable = mask.astype(bool) & (depth >= _MIN_DEPTH)
    if not usable.any():
      logging.info("valid is empty")
      return np.zeros((3))
    pixel = np.asarray([(cols.min() + cols.max()) / 2.0, (rows.min() + rows.max()) / 2.0, 1.0]).reshape(3, 1)
    return ((np.linalg.inv(K) @ pixel) * np.median(depth[usable])).reshape(3)

  def generate_random_pose_hypo(self, K, rgb, depth, mask, scene_pts=None, stats=None):
    hyp = self.rot_grid.clone()
    hyp[:, :3, 3] = torch.as_tensor(self.guess_translation(depth=depth, mask=mask, K=K, stats=stats), device='cuda', dtype=torch.float).reshape(1, 3)
    return hyp

  def compute_add_err_to_gt_pose(self, poses):
    return -torch.ones(len(poses), device='cuda', dtype=torch.float)

  # ------------------------------------------------------------------ hot path
  def _refine_and_score(self, K, rgb, depth, xyz_map, hyp, iteration):
    if self.dist_group is not None:
      from .dist import sharded_refine_and_score
      return sharded_refine_and_score(self, K, rgb, depth, xyz_map, hyp, iteration)
    shared = dict(mesh=self.mesh, mesh_tensors=self.mesh_tensors, rgb=rgb, depth=depth, K=K, glctx=self.glctx,
                  mesh_diameter=self.diameter, get_vis=self.debug >= 2)
    refined, vis = self.refiner.predict(ob_in_cams=hyp, xyz_map=xyz_map, normal_map=None, iteration=iteration, **shared)
    if vis is not None:                  # src/estimater.py:216-217 (debug >= 2)
      from .vis import write_png
      write_png(f'{self.debug_dir}/vis_refiner.png', vis)
    scores, vis = self.scorer.predict(ob_in_cams=refined, normal_map=None, **shared)
    if vis is not None:                  # src/estimater.py:220-221
      from .vis import write_png
      write_png(f'{self.debug_dir}/vis_score.png', vis)
    return refined, scores

  def register(self, K, rgb, depth, ob_mask, ob_id=None, glctx=None, iteration=5):
    """src/estimater.py:159-240: pose of the object in one RGB-D frame + mask -> np (4,4) float32
    (np.eye(4)-based float64 with the guessed translation when fewer than 4 valid pixels)."""
    U.set_seed(0)
    if self.glctx is None:
      self.glctx = U.RasterizeContext() if glctx is None else glctx
    # the frame goes to the device once; filtering, the validity count, guess_translation's reductions and the
    # back-projection all run there (the reference round-trips through numpy at each of these steps)
    depth = torch.as_tensor(np.ascontiguousarray(depth), dtype=torch.float, device='cuda')
    depth = U.bilateral_filter_depth(U.erode_depth(depth, radius=2, device='cuda'), radius=2, device='cuda')
    stats = U.mask_depth_stats(depth, ob_mask, _MIN_DEPTH)
    if stats['n_usable'] < _MIN_VALID_PIXELS:
      logging.info('valid too small, return')
      fallback = np.eye(4)
      fallback[:3, 3] = self.guess_translation(depth=depth, mask=ob_mask, K=K, stats=stats)
      return fallback
    self.H, self.W = depth.shape[:2]
    self.K, self.ob_id, self.ob_mask = K, ob_id, ob_mask
    hyp = self.generate_random_pose_hypo(K=K, rgb=rgb, depth=depth, mask=ob_mask, stats=stats)
    # the colour image goes up once, in its own dtype (uint8: 0.9 MB), and is widened on the device; refiner and scorer
    # both read that tensor
    rgb = torch.as_tensor(np.ascontiguousarray(rgb) if isinstance(rgb, np.ndarray) else rgb, device='cuda').to(torch.float)
    refined, scores = self._refine_and_score(K, rgb, depth, U.depth2xyzmap(depth, K), hyp, iteration)
    order = torch.as_tensor(scores).argsort(descending=True)
    self.poses, self.scores = refined[order], scores[order]
    self.best_id = order[0]
    self.pose_last = self.poses[0]
    return (self.pose_last @ self.get_tf_to_centered_mesh()).data.cpu().numpy()

  # ------------------------------------------------------------------ tracking
  def _track_frame(self, rgb, depth, K, pose_in, iteration, n_hyp, sigmas):
    """Device work of one tracking frame, free of host synchronisation (so that it can be captured in a hipGraph): depth
    filtering, back-projection, refinement of the previous pose (n_hyp == 1: src/estimater.py:256-266) or of n_hyp seeded
    perturbations of it + scoring.  Returns (new pose (4,4), poses, scores, best_id, new pose @ get_tf_to_centered_mesh()) -
    device tensors."""
    from .tracking import tracking_hypotheses
    # erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch (src/estimater.py:256-260; the reference back-projects with the
    # float32 camera matrix here), one launch; a uint8 frame's colours become float in the same launch
    if rgb.dtype == torch.uint8 and rgb.is_contiguous() and tuple(rgb.shape) == tuple(depth.shape) + (3,):
      depth, xyz_map, rgb = U.depth_prefilter(depth, K, radius=2, rgb_u8=rgb)
    else:
      rgb = rgb.to(torch.float)
      depth, xyz_map = U.depth_prefilter(depth, K, radius=2)
    shared = dict(mesh=self.mesh, mesh_tensors=self.mesh_tensors, rgb=rgb, depth=depth, K=K, glctx=self.glctx, mesh_diameter=self.diameter)
    if n_hyp == 1:
      pose, _ = self.refiner.predict(ob_in_cams=pose_in.reshape(-1, 4, 4), normal_map=None, xyz_map=xyz_map, iteration=iteration,
                                     get_vis=False, **shared)
      pose = pose.reshape(4, 4)
      return pose, None, None, None, pose @ self._tf_to_centered_cached()
    hyp = tracking_hypotheses(pose_in.reshape(4, 4), n_hyp, *sigmas)
    refined, _ = self.refiner.predict(ob_in_cams=hyp, xyz_map=xyz_map, normal_map=None, iteration=iteration, **shared)
    scores, _ = self.scorer.predict(ob_in_cams=refined, normal_map=None, **shared)
    best = scores.argmax()
    pose = refined.index_select(0, best.reshape(1))[0]                             # (indexing by a 0-d tensor would synchronise)
    return pose, refined, scores, best, pose @ self._tf_to_centered_cached()

  def enable_track_graph(self, on=True):
    """Replay a tracking frame as ONE hipGraph: at 1 .. 64 hypotheses a frame is ~100 kernels of one workgroup round or less
    each, so launch overhead and the gaps between launches are a large part of it.  The graph is captured at the first frame of a
    given (mode, frame size, camera, iteration count) and re-captured if the library's workspace is re-allocated."""
    self._graph_on = bool(on)
    self._graphs = {}

  def _run_frame(self, rgb, depth, K, iteration, n_hyp, sigmas):
    depth = torch.as_tensor(depth, device='cuda', dtype=torch.float)
    rgb = torch.as_tensor(np.ascontiguousarray(rgb) if isinstance(rgb, np.ndarray) else rgb, device='cuda')
    pose_in = self.pose_last.reshape(4, 4).to(torch.float)
    if not getattr(self, '_graph_on', False):
      return self._track_frame(rgb, depth, K, pose_in, iteration, n_hyp, sigmas)
    ctx = self.refiner.ctx
    key = (n_hyp, sigmas, int(iteration), tuple(rgb.shape), rgb.dtype, np.asarray(K, dtype=np.float64).tobytes(), id(self.mesh_tensors['pos']))
    entry = self._graphs.get(key)
    if entry is not None and entry['generation'] != ctx.arena_generation():
      entry = None                                          # the workspace moved: the captured addresses are stale
    if entry is None:
      ctx.reserve(max(64, n_hyp))
      st = dict(rgb=rgb.clone(), depth=depth.clone(), pose=pose_in.clone())
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):                          # eager passes first: lazy initialisation, allocator warm-up
        for _ in range(2):
          self._track_frame(st['rgb'], st['depth'], K, st['pose'], iteration, n_hyp, sigmas)
      torch.cuda.current_stream().wait_stream(side)
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph):
        st['out'] = self._track_frame(st['rgb'], st['depth'], K, st['pose'], iteration, n_hyp, sigmas)
      entry = self._graphs[key] = dict(graph=graph, st=st, generation=ctx.arena_generation())
    st = entry['st']
    st['rgb'].copy_(rgb)
    st['depth'].copy_(depth)
    st['pose'].copy_(pose_in)
    entry['graph'].replay()
    out = st['out']      # the static outputs are overwritten by the next replay: cloned, except the last (the caller copies it to the host at once)
    return tuple(None if t is None else t.clone() for t in out[:-1]) + (out[-1],)

  def track_one(self, rgb, depth, K, iteration, extra={}):
    """src/estimater.py:250-268: refine the previous pose against a new frame (no scoring)."""
    if self.pose_last is None:
      logging.info("Please init pose by register first")
      raise RuntimeError
    pose_in = self.pose_last
    pose, _, _, _, pose_of_mesh = self._run_frame(rgb, depth, K, iteration, 1, None)
    if self.debug >= 2:            # src/estimater.py:263-266: the refiner's canvas for this frame (debug only: the frame is refined a second time for it)
      d = U.bilateral_filter_depth(U.erode_depth(torch.as_tensor(depth, device='cuda', dtype=torch.float), radius=2, device='cuda'), radius=2, device='cuda')
      xyz_map = U.depth2xyzmap_batch(d[None], np.asarray(K, dtype=np.float32)[None], zfar=np.inf)[0]
      _, extra['vis'] = self.refiner.predict(mesh=self.mesh, mesh_tensors=self.mesh_tensors, rgb=rgb, depth=d, K=K, glctx=self.glctx,
                                             mesh_diameter=self.diameter, ob_in_cams=pose_in.reshape(-1, 4, 4), normal_map=None, xyz_map=xyz_map,
                                             iteration=iteration, get_vis=True)
    self.pose_last = pose
    return pose_of_mesh.data.cpu().numpy().reshape(4, 4)      # src/estimater.py:268: pose @ get_tf_to_centered_mesh() (inside the frame's graph)

  def track_multi(self, rgb, depth, K, iteration, n_hypotheses=64, trans_sigma=0.01, rot_sigma_deg=5.0, extra={}):
    """Multi-hypothesis tracking (BASELINE.json configs[4]; a build extension, the reference's track_one refines one pose
    and never scores): the previous pose and n-1 fixed seeded perturbations of it (tracking.tracking_hypotheses) are refined
    together, scored by ScoreNet, and the best-scoring refined pose becomes `pose_last`.  Same prelude and return value as
    track_one; `self.poses` / `self.scores` hold all hypotheses of the frame in hypothesis order, `self.best_id` the winner."""
    if self.pose_last is None:
      logging.info("Please init pose by register first")
      raise RuntimeError
    pose, self.poses, self.scores, self.best_id, pose_of_mesh = self._run_frame(rgb, depth, K, iteration, int(n_hypotheses),
                                                                                (float(trans_sigma), float(rot_sigma_deg)))
    self.pose_last = pose
    return pose_of_mesh.data.cpu().numpy().reshape(4, 4)

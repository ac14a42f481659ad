"""FoundationPose estimator on the MI355X hot path.

Drop-in for the reference orchestrator `src/estimater.py:18-268`: same constructor and method
signatures, same public attributes (`pose_last, poses, scores, best_id, diameter, mesh, mesh_tensors,
rot_grid, symmetry_tfs, model_center, glctx, scorer, refiner, ...`), same return types and the same
three degenerate-input behaviours (empty mask -> zero translation guess, <4 valid pixels -> identity
rotation + guessed translation as float64, track before register -> RuntimeError).  The compute is
the HIP library; with `dist_group` set the hypotheses are sharded over the ranks of one node
(foundationpose_amd/dist.py) - the reference is single-GPU only.
"""
import ctypes
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
    # tracking workspaces (and their captured graphs) hold the previous object's mesh handle and centre: none survives a new object
    self._track_ws = {}
    logging.info("reset done")

  def get_tf_to_centered_mesh(self):
    tf = torch.eye(4, dtype=torch.float, device='cuda')
    tf[:3, 3] = -torch.as_tensor(self.model_center, device='cuda', dtype=torch.float)
    return tf

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
    # (register's hypotheses are the rotation grid around ONE guessed centre: the refiner encodes the observed crop of its first iteration once)
    refined, vis = self.refiner.predict(ob_in_cams=hyp, xyz_map=xyz_map, normal_map=None, iteration=iteration, shared_translation=True, **shared)
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
  def enable_track_graph(self, on=True):
    """Replay a tracking frame as ONE hipGraph: at 1 .. 64 hypotheses a frame is ~80 kernels of one workgroup round or less
    each, so launch overhead and the gaps between launches are a large part of it.  The graph is captured at the first frame of a
    given (mode, frame size, camera, iteration count) and re-captured if the library's workspace is re-allocated."""
    self._graph_on = bool(on)
    for ws in getattr(self, '_track_ws', {}).values():
      ws['graph'] = None

  def _track_workspace(self, n_hyp, sigmas, iteration, shape, is_u8, K):
    """Static device buffers of one tracking mode (every launch of a frame reads and writes fixed addresses, so the frame can be
    replayed as a hipGraph): the frame itself - depth (float32) and colours behind one another in ONE buffer, so that a frame is one
    host-to-device copy -, the library's workspace, the pose (refined in place from frame to frame) and the outputs."""
    from . import _lib
    from .tracking import perturbation_set
    key = (n_hyp, sigmas, int(iteration), tuple(shape), bool(is_u8), np.asarray(K, dtype=np.float64).tobytes(), id(self.mesh_tensors['pos']),
           np.asarray(self.model_center, dtype=np.float64).tobytes())
    ws = self._track_ws.get(key) if hasattr(self, '_track_ws') else None
    if ws is not None:
      return ws
    if not hasattr(self, '_track_ws'):
      self._track_ws = {}
    H, W = shape
    dev = self.mesh_tensors['pos'].device
    n_rgb = H * W * 3 * (1 if is_u8 else 4)
    frame = torch.empty((H * W * 4 + n_rgb,), dtype=torch.uint8, device=dev)
    ws = dict(frame=frame, depth=frame[:H * W * 4].view(torch.float).reshape(H, W),
              rgb=(frame[H * W * 4:].reshape(H, W, 3) if is_u8 else frame[H * W * 4:].view(torch.float).reshape(H, W, 3)),
              host=torch.empty((H * W * 4 + n_rgb,), dtype=torch.uint8).pin_memory(),
              depth_f=torch.empty((H, W), dtype=torch.float, device=dev), xyz=torch.empty((H, W, 3), dtype=torch.float, device=dev),
              rgb_f=torch.empty((H, W, 3), dtype=torch.float, device=dev) if is_u8 else None,
              pose=torch.eye(4, dtype=torch.float, device=dev), pose_of_mesh=torch.eye(4, dtype=torch.float).pin_memory(),
              Kd=np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3)), cfg=self.refiner._c_cfg(), holds_pose=None, graph=None)
    a = _lib.FpTrackArgs()
    a.struct_size = ctypes.sizeof(a)
    a.refine_net, a.mesh = self.refiner.model.handle, _lib.device_mesh(self.refiner.ctx, self.mesh_tensors).handle
    a.d_rgb, a.rgb_is_u8, a.d_depth, a.H, a.W = ws['rgb'].data_ptr(), 1 if is_u8 else 0, ws['depth'].data_ptr(), H, W
    a.K, a.mesh_diameter = ws['Kd'].ctypes.data, float(self.diameter)
    a.refine_cfg = ctypes.addressof(ws['cfg'])
    a.iteration, a.n_hyp = int(iteration), int(n_hyp)
    a.model_center[:] = [float(x) for x in np.asarray(self.model_center, dtype=np.float32)]
    a.d_pose, a.d_pose_of_mesh = ws['pose'].data_ptr(), ws['pose_of_mesh'].data_ptr()
    a.d_depth_f, a.d_xyz = ws['depth_f'].data_ptr(), ws['xyz'].data_ptr()
    a.d_rgb_f = ws['rgb_f'].data_ptr() if is_u8 else None
    if n_hyp > 1:
      ws['perturb'] = torch.as_tensor(perturbation_set(int(n_hyp), *sigmas), device=dev).contiguous()
      ws['poses'] = torch.empty((n_hyp, 4, 4), dtype=torch.float, device=dev)
      ws['scores'] = torch.empty((n_hyp,), dtype=torch.float, device=dev)
      ws['best'] = torch.zeros((1,), dtype=torch.int32, device=dev)
      a.score_net = self.scorer.model.handle
      a.score_crop_ratio, a.score_normalize_xyz = float(self.scorer.cfg['crop_ratio']), 1 if self.scorer.cfg['normalize_xyz'] else 0
      a.d_perturb, a.d_poses, a.d_scores, a.d_best = ws['perturb'].data_ptr(), ws['poses'].data_ptr(), ws['scores'].data_ptr(), ws['best'].data_ptr()
    ws['args'] = a
    self._track_ws[key] = ws
    return ws

  def _run_frame(self, rgb, depth, K, iteration, n_hyp, sigmas):
    """One tracking frame: upload (ONE copy), fp_track_frame (eager, or the replay of its captured hipGraph), the 4x4 result read from pinned host memory.
    Returns (pose_of_mesh as a HOST (4,4) float32 array - what track_one returns -, workspace).  Contract: `self.pose_last` and, in the
    multi-hypothesis mode, `self.poses` / `self.scores` / `self.best_id` are VIEWS of the mode's static buffers, overwritten by the
    next frame of the same mode; clone them to keep them."""
    from . import _lib
    from ._lib import check, lib, stream_ptr
    ctx = self.refiner.ctx
    is_np = isinstance(rgb, np.ndarray)
    is_u8 = (rgb.dtype == np.uint8) if is_np else (rgb.dtype == torch.uint8)
    H, W = depth.shape[:2]
    ws = self._track_workspace(n_hyp, sigmas, iteration, (H, W), is_u8, K)
    # the frame: host arrays go through one pinned staging buffer and ONE host-to-device copy; device tensors that already lie behind
    # one another as [depth | rgb] (bench.py packs its resident frames that way) are one device copy, others two
    if is_np or not torch.is_tensor(depth) or not depth.is_cuda:
      hb = ws['host'].numpy()
      hb[:H * W * 4].view(np.float32)[:] = np.asarray(depth.cpu() if torch.is_tensor(depth) else depth, dtype=np.float32).reshape(-1)
      r = np.asarray(rgb.cpu() if torch.is_tensor(rgb) else rgb)
      if is_u8:
        hb[H * W * 4:] = r.reshape(-1)
      else:
        hb[H * W * 4:].view(np.float32)[:] = r.astype(np.float32, copy=False).reshape(-1)
      ws['frame'].copy_(ws['host'], non_blocking=True)
    else:
      d = depth.to(torch.float)
      r = rgb if is_u8 else rgb.to(torch.float)
      packed = (d.is_contiguous() and r.is_contiguous() and d.untyped_storage().data_ptr() == r.untyped_storage().data_ptr() and
                r.data_ptr() == d.data_ptr() + H * W * 4)
      if packed:
        ws['frame'].copy_(torch.as_strided(d.view(torch.uint8).reshape(-1), (ws['frame'].numel(),), (1,)))
      else:
        ws['depth'].copy_(d)
        ws['rgb'].copy_(r)
    # the pose lives in the workspace and is refined in place; it is (re)loaded only when pose_last was set by someone else (register)
    if ws['holds_pose'] is not self.pose_last:
      ws['pose'].copy_(torch.as_tensor(self.pose_last, device=ws['pose'].device, dtype=torch.float).reshape(4, 4))
    st = torch.cuda.current_stream(ws['pose'].device)
    if getattr(self, '_graph_on', False):
      g = ws['graph']
      if g is not None and g[1] != ctx.arena_generation():
        g = None                                            # the library's arena moved: the captured addresses are stale
      if g is None:
        ctx.reserve(max(64, n_hyp))
        keep = ws['pose'].clone()
        side = torch.cuda.Stream()
        side.wait_stream(st)
        with torch.cuda.stream(side):                        # eager passes first: lazy initialisation, allocator warm-up
          for _ in range(2):
            check(lib().fp_track_frame(ctx.handle, ctypes.byref(ws['args']), stream_ptr(ws['pose'].device)))
            ws['pose'].copy_(keep)
        st.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
          check(lib().fp_track_frame(ctx.handle, ctypes.byref(ws['args']), stream_ptr(ws['pose'].device)))
        ws['pose'].copy_(keep)                               # (capturing does not execute)
        g = ws['graph'] = (graph, ctx.arena_generation())
      g[0].replay()
    else:
      ctx.reserve(max(64, n_hyp))
      check(lib().fp_track_frame(ctx.handle, ctypes.byref(ws['args']), stream_ptr(ws['pose'].device)))
    # the result lands in pinned host memory, written there by the frame's last launch (the device addresses it directly): no
    # device-to-host copy is enqueued, the frame ends with one wait for the stream
    st.synchronize()
    out = ws['pose_of_mesh'].numpy().copy()
    return out, ws

  def track_one(self, rgb, depth, K, iteration, extra={}):
    """src/estimater.py:250-268: refine the previous pose against a new frame (no scoring)."""
    if self.pose_last is None:
      logging.info("Please init pose by register first")
      raise RuntimeError
    pose_in = torch.as_tensor(self.pose_last).clone() if self.debug >= 2 else None      # (the debug canvas re-refines the frame from it)
    pose_of_mesh, ws = self._run_frame(rgb, depth, K, iteration, 1, None)
    if self.debug >= 2:            # src/estimater.py:263-266: the refiner's canvas for this frame (debug only: the frame is refined a second time for it)
      d = U.bilateral_filter_depth(U.erode_depth(torch.as_tensor(depth, device='cuda', dtype=torch.float), radius=2, device='cuda'), radius=2, device='cuda')
      xyz_map = U.depth2xyzmap_batch(d[None], np.asarray(K, dtype=np.float32)[None], zfar=np.inf)[0]
      _, extra['vis'] = self.refiner.predict(mesh=self.mesh, mesh_tensors=self.mesh_tensors, rgb=rgb, depth=d, K=K, glctx=self.glctx,
                                             mesh_diameter=self.diameter, ob_in_cams=pose_in.reshape(-1, 4, 4), normal_map=None, xyz_map=xyz_map,
                                             iteration=iteration, get_vis=True)
    self.pose_last = ws['pose'].reshape(1, 4, 4)          # src/estimater.py:267: the refiner's (1,4,4) output (a view of the tracking workspace)
    ws['holds_pose'] = self.pose_last
    return pose_of_mesh                                   # src/estimater.py:268: pose @ get_tf_to_centered_mesh() (written by the frame's last launch)

  def track_multi(self, rgb, depth, K, iteration, n_hypotheses=64, trans_sigma=0.01, rot_sigma_deg=5.0, extra={}):
    """Multi-hypothesis tracking (BASELINE.json configs[4]; a build extension, the reference's track_one refines one pose
    and never scores): the previous pose and n-1 fixed seeded perturbations of it (tracking.perturbation_set) are refined
    together, scored by ScoreNet, and the best-scoring refined pose becomes `pose_last`.  Same prelude and return value as
    track_one; `self.poses` / `self.scores` hold all hypotheses of the frame in hypothesis order, `self.best_id` the winner
    (views of the tracking workspace: the next frame overwrites them)."""
    if self.pose_last is None:
      logging.info("Please init pose by register first")
      raise RuntimeError
    pose_of_mesh, ws = self._run_frame(rgb, depth, K, iteration, int(n_hypotheses), (float(trans_sigma), float(rot_sigma_deg)))
    self.poses, self.scores, self.best_id = ws['poses'], ws['scores'], ws['best'][0]
    self.pose_last = ws['pose'].reshape(1, 4, 4)
    ws['holds_pose'] = self.pose_last
    return pose_of_mesh

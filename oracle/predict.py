"""Oracle: the two predictors and the estimator loop, restated on torch-CPU fp32.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Citations are reference file:line.
"""
import numpy as np
import torch

from . import geometry as G
from . import nets
from .render import nvdiffrast_render
from .warp import warp_perspective, warp_perspective_nearest

DEFAULT_REFINE_CFG = dict(input_resize=(160, 160), c_in=6, use_BN=True, normalize_xyz=True, use_normal=False,
                          crop_ratio=1.2, trans_rep='tracknet', rot_rep='axis_angle',
                          trans_normalizer=[0.02, 0.02, 0.05], rot_normalizer=0.3490659)
DEFAULT_SCORE_CFG = dict(input_resize=(160, 160), c_in=6, use_BN=True, normalize_xyz=True, use_normal=False,
                         crop_ratio=1.1)


def _render_batch(cfg, K, H, W, poseA, mesh_tensors, tf_to_crops, want_normal=False):
  """predict_pose_refine.py:44-56 / predict_score.py:71-86: render at input_resize inside bbox2d_ori."""
  bbox2d_ori = G.crop_bbox2d_ori(tf_to_crops, cfg['input_resize'])
  extra = {}
  rgb_r, depth_r, normal_r = nvdiffrast_render(K=K, H=H, W=W, ob_in_cams=poseA, get_normal=cfg['use_normal'],
                                               mesh_tensors=mesh_tensors, output_size=cfg['input_resize'],
                                               bbox2d=bbox2d_ori, use_light=True, extra=extra)
  rgb_rs = rgb_r.permute(0, 3, 1, 2) * 255
  depth_rs = depth_r[..., None].permute(0, 3, 1, 2)
  xyz_map_rs = extra['xyz_map'].permute(0, 3, 1, 2)
  if want_normal:
    return rgb_rs, depth_rs, xyz_map_rs, normal_r.permute(0, 3, 1, 2)
  return rgb_rs, depth_rs, xyz_map_rs


def _xyz_transform(xyz, poseA, mesh_diameters, normalize_xyz, invalid_thres, invalid_always):
  """h5_dataset.py:92-99 (Pair: threshold 0.001, `invalid` only exists when normalize_xyz) and
  :151-156 (Triplet: threshold 0.1, computed always, applied only when normalize_xyz)."""
  bs = len(xyz)
  mesh_radius = mesh_diameters / 2
  if normalize_xyz or invalid_always:
    invalid = xyz[:, 2:3] < invalid_thres
  xyz = xyz - poseA[:, :3, 3].reshape(bs, 3, 1, 1)
  if normalize_xyz:
    xyz = xyz * (1 / mesh_radius.reshape(bs, 1, 1, 1))
    invalid = invalid.expand(bs, 3, -1, -1) | (torch.abs(xyz) >= 2)
    xyz = xyz.clone()
    xyz[invalid.expand(bs, 3, -1, -1)] = 0
  return xyz


def make_crop_data_batch_refine(cfg, ob_in_cams, mesh_tensors, rgb, depth, K, xyz_map, mesh_diameter, normal_map=None):
  """predict_pose_refine.py:26-89 + PairH5Dataset.transform_batch (h5_dataset.py:79-127,210-219).
  rgb (H,W,3) float tensor [0,255], xyz_map (H,W,3).  Returns dict of network-ready tensors.  Under cfg['use_normal']
  also normalAs / normalBs (:74-76: BOTH warped by tf_to_crops, nearest - the rendered normals are in crop coordinates
  already, the reference warps them again regardless; transform_batch leaves them alone)."""
  H, W = depth.shape[:2]
  render_size = cfg['input_resize']
  poseA = torch.as_tensor(ob_in_cams, dtype=torch.float32)
  B = len(poseA)
  tf_to_crops = G.compute_crop_window_tf_batch(poseA, K, cfg['crop_ratio'], (render_size[1], render_size[0]), mesh_diameter)
  normals = {}
  if cfg.get('use_normal', False):
    rgb_rs, _, xyz_map_rs, normal_rs = _render_batch(cfg, K, H, W, poseA, mesh_tensors, tf_to_crops, want_normal=True)
    normals['normalAs'] = warp_perspective_nearest(normal_rs.contiguous(), tf_to_crops, render_size)
    nm = torch.as_tensor(normal_map, dtype=torch.float32)
    normals['normalBs'] = warp_perspective_nearest(nm.permute(2, 0, 1)[None].expand(B, -1, -1, -1).contiguous(), tf_to_crops, render_size)
  else:
    rgb_rs, _, xyz_map_rs = _render_batch(cfg, K, H, W, poseA, mesh_tensors, tf_to_crops)
  rgbBs = warp_perspective(rgb.permute(2, 0, 1)[None].expand(B, -1, -1, -1), tf_to_crops, dsize=render_size, mode='bilinear', align_corners=False)
  xyz_mapBs = warp_perspective_nearest(xyz_map.permute(2, 0, 1)[None].expand(B, -1, -1, -1).contiguous(), tf_to_crops, render_size)
  mesh_diameters = torch.ones((B,), dtype=torch.float32) * mesh_diameter
  rgbAs = rgb_rs / 255.0
  rgbBs = rgbBs / 255.0
  xyz_mapAs = _xyz_transform(xyz_map_rs, poseA, mesh_diameters, cfg['normalize_xyz'], 0.001, False)
  xyz_mapBs = _xyz_transform(xyz_mapBs, poseA, mesh_diameters, cfg['normalize_xyz'], 0.001, False)
  return dict(rgbAs=rgbAs, rgbBs=rgbBs, xyz_mapAs=xyz_mapAs, xyz_mapBs=xyz_mapBs, poseA=poseA, tf_to_crops=tf_to_crops, **normals)


def make_crop_data_batch_score(cfg, ob_in_cams, mesh_tensors, rgb, depth, K, mesh_diameter):
  """predict_score.py:57-114 + TripletH5Dataset.transform_batch (h5_dataset.py:137-179): side B's
  xyz is rebuilt from the cropped depth through the full-resolution nearest round trip (:158-161)."""
  H, W = depth.shape[:2]
  render_size = cfg['input_resize']
  poseA = torch.as_tensor(ob_in_cams, dtype=torch.float32)
  B = len(poseA)
  tf_to_crops = G.compute_crop_window_tf_batch(poseA, K, cfg['crop_ratio'], (render_size[1], render_size[0]), mesh_diameter)
  rgb_rs, depth_rs, xyz_map_rs = _render_batch(cfg, K, H, W, poseA, mesh_tensors, tf_to_crops)
  rgbBs = warp_perspective(rgb.permute(2, 0, 1)[None].expand(B, -1, -1, -1), tf_to_crops, dsize=render_size, mode='bilinear', align_corners=False)
  depthBs = warp_perspective_nearest(depth[None, None].expand(B, -1, -1, -1).contiguous(), tf_to_crops, render_size)
  mesh_diameters = torch.ones((B,), dtype=torch.float32) * mesh_diameter
  Ks = torch.as_tensor(np.asarray(K), dtype=torch.float32).reshape(1, 3, 3).expand(B, 3, 3)
  rgbAs = rgb_rs / 255.0
  rgbBs = rgbBs / 255.0
  xyz_mapAs = _xyz_transform(xyz_map_rs, poseA, mesh_diameters, cfg['normalize_xyz'], 0.1, True)
  crop_to_oris = torch.linalg.inv(tf_to_crops)
  chunks = []
  for s in range(0, B, 16):   # the reference materialises (B,H,W,3) at once; chunked here for memory
    depthBs_ori = warp_perspective_nearest(depthBs[s:s + 16], crop_to_oris[s:s + 16], (H, W))
    xyz_full = G.depth2xyzmap_batch(depthBs_ori[:, 0], Ks[s:s + 16], zfar=np.inf).permute(0, 3, 1, 2)
    chunks.append(warp_perspective_nearest(xyz_full, tf_to_crops[s:s + 16], render_size))
  xyz_mapBs = torch.cat(chunks, 0)
  xyz_mapBs = _xyz_transform(xyz_mapBs, poseA, mesh_diameters, cfg['normalize_xyz'], 0.1, True)
  return dict(rgbAs=rgbAs, rgbBs=rgbBs, xyz_mapAs=xyz_mapAs, xyz_mapBs=xyz_mapBs, depthAs=depth_rs, depthBs=depthBs,
              poseA=poseA, tf_to_crops=tf_to_crops)


def pose_update(cfg, poseA, trans, rot, mesh_diameter, tf_to_crops=None, Ks=None):
  """predict_pose_refine.py:195-231 (tracknet / deepim; axis_angle|6d branches).  deepim needs the pass's tf_to_crops (N,3,3) and
  Ks (N or 1,3,3), as the reference reads them from pose_data."""
  if cfg['trans_rep'] == 'tracknet':
    if not cfg['normalize_xyz']:
      tn = cfg['trans_normalizer']
      if not isinstance(tn, float):
        tn = torch.as_tensor(list(tn), dtype=torch.float32).reshape(1, 3)
      trans_delta = torch.tanh(trans) * tn
    else:
      trans_delta = trans
  elif cfg['trans_rep'] == 'deepim':                     # predict_pose_refine.py:201-215
    n = len(trans)
    tf = torch.as_tensor(tf_to_crops, dtype=torch.float32).reshape(-1, 3, 3)
    Kt = torch.as_tensor(Ks, dtype=torch.float32).reshape(-1, 3, 3).expand(n, 3, 3)

    def project_and_transform_to_crop(centers):
      uvs = (Kt @ centers.reshape(-1, 3, 1)).reshape(-1, 3)
      uvs = uvs / uvs[:, 2:3]
      uvs = (tf @ uvs.reshape(-1, 3, 1)).reshape(-1, 3)
      return uvs[:, :2]

    z_pred = trans[:, 2] * poseA[..., 2, 3]
    uvA_crop = project_and_transform_to_crop(poseA[..., :3, 3])
    uv_pred_crop = uvA_crop + trans[:, :2] * cfg['input_resize'][0]
    tfi = tf.inverse()
    uv_pred = (tfi[:, :2, :2] @ uv_pred_crop[..., None])[..., 0] + tfi[:, :2, 2]          # transform_pts (src/Utils.py:529-536)
    center_pred = torch.cat([uv_pred, torch.ones((n, 1), dtype=torch.float32)], dim=-1)
    center_pred = (Kt.inverse() @ center_pred.reshape(n, 3, 1)).reshape(n, 3) * z_pred.reshape(n, 1)
    trans_delta = center_pred - poseA[..., :3, 3]
  else:
    trans_delta = trans
  if cfg['rot_rep'] == 'axis_angle':
    rot_mat_delta = torch.tanh(rot) * cfg['rot_normalizer']
    rot_mat_delta = G.so3_exp_map(rot_mat_delta).permute(0, 2, 1)
  elif cfg['rot_rep'] == '6d':
    rot_mat_delta = G.rotation_6d_to_matrix(rot).permute(0, 2, 1)
  else:
    raise RuntimeError
  if cfg['normalize_xyz']:
    trans_delta = trans_delta * (mesh_diameter / 2)
  return G.egocentric_delta_pose_to_pose(poseA, trans_delta=trans_delta, rot_mat_delta=rot_mat_delta), trans_delta, rot_mat_delta


@torch.no_grad()
def refine_predict(cfg, sd, rgb, depth, K, ob_in_cams, xyz_map, mesh_tensors, mesh_diameter, iteration=5, chunk=16, trace=None, autocast=False, recipe=None):
  """PoseRefinePredictor.predict (predict_pose_refine.py:150-237), fp32; recipe='d16': the network with the HIP kernels' rounding points
  (nets.refine_forward_d16; everything outside the network stays the fp32 oracle); autocast=True: the network under
  torch.autocast('cpu', dtype=torch.float16) - the reference's own precision on the GPU (predict_pose_refine.py:190) - with the
  outputs taken back to fp32 for the pose update, as the reference does (`output[k].float()`, :195-200)."""
  B_in_cams = torch.as_tensor(ob_in_cams, dtype=torch.float32)
  rgb_t = torch.as_tensor(rgb, dtype=torch.float32)
  depth_t = torch.as_tensor(depth, dtype=torch.float32)
  xyz_t = torch.as_tensor(xyz_map, dtype=torch.float32)
  folded = nets.fold_trunk_d16(sd, 'encodeA', 'encodeAB', cfg['use_BN']) if recipe == 'd16' else None
  for it in range(iteration):
    pd = make_crop_data_batch_refine(cfg, B_in_cams, mesh_tensors, rgb_t, depth_t, K, xyz_t, mesh_diameter)
    outs = []
    for b in range(0, len(B_in_cams), chunk):
      A = torch.cat([pd['rgbAs'][b:b + chunk], pd['xyz_mapAs'][b:b + chunk]], dim=1).float()
      Bt = torch.cat([pd['rgbBs'][b:b + chunk], pd['xyz_mapBs'][b:b + chunk]], dim=1).float()
      if recipe == 'd16':
        o = nets.refine_forward_d16(sd, A, Bt, cfg['use_BN'], folded=folded)
      elif autocast:
        with torch.autocast('cpu', dtype=torch.float16):
          o = nets.refine_forward(sd, A, Bt, cfg['use_BN'])
        o = {k: v.float() for k, v in o.items()}
      else:
        o = nets.refine_forward(sd, A, Bt, cfg['use_BN'])
      new_pose, td, rd = pose_update(cfg, pd['poseA'][b:b + chunk], o['trans'], o['rot'], mesh_diameter,
                                     tf_to_crops=pd['tf_to_crops'][b:b + chunk], Ks=np.asarray(K, dtype=np.float32))
      outs.append((new_pose, o['trans'], o['rot']))
    if trace is not None:
      trace.append(dict(poseA=pd['poseA'].clone(), trans=torch.cat([o[1] for o in outs]), rot=torch.cat([o[2] for o in outs]),
                        tf_to_crops=pd['tf_to_crops'].clone()))
    B_in_cams = torch.cat([o[0] for o in outs], dim=0).reshape(len(ob_in_cams), 4, 4)
  return B_in_cams


@torch.no_grad()
def score_predict(cfg, sd, rgb, depth, K, ob_in_cams, mesh_tensors, mesh_diameter, chunk=16, trace=None):
  """ScorePredictor.predict (predict_score.py:161-226): one forward over all hypotheses (the
  tournament loop exits in its first round because bs == N), scores = logits + 100 (:209)."""
  rgb_t = torch.as_tensor(rgb, dtype=torch.float32)
  depth_t = torch.as_tensor(depth, dtype=torch.float32)
  pd = make_crop_data_batch_score(cfg, ob_in_cams, mesh_tensors, rgb_t, depth_t, K, mesh_diameter)
  A = torch.cat([pd['rgbAs'], pd['xyz_mapAs']], dim=1).float()
  Bt = torch.cat([pd['rgbBs'], pd['xyz_mapBs']], dim=1).float()
  out = nets.score_forward(sd, A, Bt, L=len(A), use_bn=cfg['use_BN'], chunk=chunk)
  scores_cur = out['score_logit'].float().reshape(-1)
  if trace is not None:
    trace.append(dict(feats=out['feats'], logits=scores_cur.clone()))
  return scores_cur + 100


class OracleFoundationPose:
  """FoundationPose.register / track_one (src/estimater.py:159-268) on the CPU oracle.  The object
  set-up (reset_object :44-78, make_rotation_grid :106-124) is passed in pre-computed: mesh is
  already centred, `rot_grid` is the shared input fixture (icosphere order unpinned)."""

  def __init__(self, mesh_tensors, diameter, model_center, rot_grid, refine_sd, score_sd,
               refine_cfg=None, score_cfg=None):
    self.mesh_tensors = mesh_tensors
    self.diameter = float(diameter)
    self.model_center = np.asarray(model_center, dtype=np.float64)
    self.rot_grid = torch.as_tensor(rot_grid, dtype=torch.float32)
    self.refine_sd, self.score_sd = refine_sd, score_sd
    self.refine_cfg = dict(DEFAULT_REFINE_CFG, **(refine_cfg or {}))
    self.score_cfg = dict(DEFAULT_SCORE_CFG, **(score_cfg or {}))
    self.pose_last = None

  def get_tf_to_centered_mesh(self):
    tf = torch.eye(4, dtype=torch.float32)
    tf[:3, 3] = -torch.as_tensor(self.model_center, dtype=torch.float32)
    return tf

  def register(self, K, rgb, depth, ob_mask, iteration=5, chunk=16, trace=None):
    depth = G.erode_depth(depth, radius=2)
    depth = G.bilateral_filter_depth(depth, radius=2)
    valid = (depth >= 0.001) & (ob_mask > 0)
    if valid.sum() < 4:
      pose = np.eye(4)
      pose[:3, 3] = G.guess_translation(depth=depth, mask=ob_mask, K=K)
      return pose
    center = G.guess_translation(depth=depth, mask=ob_mask, K=K)
    poses = self.rot_grid.clone()
    poses[:, :3, 3] = torch.as_tensor(center.reshape(1, 3), dtype=torch.float32)
    xyz_map = G.depth2xyzmap(depth, K)
    tr = None if trace is None else trace.setdefault('refine', [])
    poses = refine_predict(self.refine_cfg, self.refine_sd, rgb, depth, K, poses.numpy(), xyz_map, self.mesh_tensors,
                           self.diameter, iteration=iteration, chunk=chunk, trace=tr)
    ts = None if trace is None else trace.setdefault('score', [])
    scores = score_predict(self.score_cfg, self.score_sd, rgb, depth, K, poses.numpy(), self.mesh_tensors, self.diameter,
                           chunk=chunk, trace=ts)
    ids = torch.as_tensor(scores).argsort(descending=True)
    scores = scores[ids]
    poses = poses[ids]
    best_pose = poses[0] @ self.get_tf_to_centered_mesh()
    self.pose_last = poses[0]
    self.best_id = ids[0]
    self.poses = poses
    self.scores = scores
    return best_pose.numpy()

  def track_one(self, rgb, depth, K, iteration, chunk=16):
    if self.pose_last is None:
      raise RuntimeError
    depth = G.erode_depth(depth, radius=2)
    depth = G.bilateral_filter_depth(depth, radius=2)
    depth_t = torch.as_tensor(depth, dtype=torch.float32)
    xyz_map = G.depth2xyzmap_batch(depth_t[None], torch.as_tensor(np.asarray(K), dtype=torch.float32)[None], zfar=np.inf)[0]
    pose = refine_predict(self.refine_cfg, self.refine_sd, rgb, depth, K, self.pose_last.reshape(-1, 4, 4).numpy(), xyz_map,
                          self.mesh_tensors, self.diameter, iteration=iteration, chunk=chunk)
    self.pose_last = pose
    return (pose @ self.get_tf_to_centered_mesh()).numpy().reshape(-1, 4, 4)[0] if len(pose) == 1 else (pose @ self.get_tf_to_centered_mesh()).numpy()

"""ScorePredictor - mirror of learning/training/predict_score.py:118-226 on the HIP library."""
import logging

import numpy as np
import torch

from . import _lib
from ._lib import FP_NET_SCORE, check, k_ptr, lib, ptr, stream_ptr
from .Utils import _ctx_of, make_mesh_tensors
from .config import Cfg, check_network_cfg, load_run_dir
from .pose_dataset import BatchPoseData, planar_views
from .predict_pose_refine import crop_net_input


@torch.inference_mode()
def make_crop_data_batch(render_size, ob_in_cams, mesh, rgb, depth, K, crop_ratio, normal_map=None, mesh_diameter=None, glctx=None,
                         mesh_tensors=None, dataset=None, cfg=None):
  """predict_score.py:56-114 including dataset.transform_batch (h5_dataset.py:137-179): the batch ScoreNet reads.
  Side B's xyz comes from the cropped depth through the full-resolution round trip of h5_dataset.py:158-161, composed
  per pixel inside fp_crop_observed.  depthAs / depthBs are float32 (render depth; nearest crop of `depth`)."""
  cfg = cfg if cfg is not None else {}
  # cfg['use_normal'] only makes the reference's render compute normals it then drops (predict_score.py:79 get_normal=..., normal_r
  # unused; :103-104 normalAs = normalBs = None whatever the flag), and `normal_map` is never read: both change nothing here
  ctx = _ctx_of(glctx)
  dev = torch.device('cuda', ctx.device_index)
  if mesh_tensors is None:
    mesh_tensors = make_mesh_tensors(mesh, device=dev)
  dm = _lib.device_mesh(ctx, mesh_tensors)
  poses = torch.as_tensor(ob_in_cams, device=dev, dtype=torch.float).reshape(-1, 4, 4).contiguous()
  rgb_t = torch.as_tensor(rgb, device=dev, dtype=torch.float).contiguous()
  depth_t = torch.as_tensor(depth, device=dev, dtype=torch.float).contiguous()
  H, W = depth_t.shape[:2]
  net, tf, bbox = crop_net_input(ctx, dm, poses, rgb_t, depth_t, K, crop_ratio, mesh_diameter, cfg.get('normalize_xyz', False), 1, render_size)
  rgbAs, xyz_mapAs, rgbBs, xyz_mapBs = planar_views(net)
  N = len(poses)
  h, w = int(render_size[0]), int(render_size[1])
  Kd, Kp = k_ptr(K)
  s = stream_ptr(dev)
  depthAs = torch.empty((N, 1, h, w), device=dev, dtype=torch.float)
  check(lib().fp_render(ctx.handle, dm.handle, ptr(poses), N, Kp, H, W, ptr(bbox), h, w, 1, 0.8, 0.5, None, ptr(depthAs), None, None, s))
  # depthBs: the nearest-sampled crop alone = the refiner-mode kernel on (0,0,depth) with nothing to centre on
  zmap = torch.zeros((H, W, 3), device=dev, dtype=torch.float)
  zmap[..., 2] = depth_t
  origin = torch.zeros((N, 4, 4), device=dev, dtype=torch.float)
  raw = torch.empty((N, 6, h, w), device=dev, dtype=torch.float)
  check(lib().fp_crop_observed(ctx.handle, ptr(rgb_t), ptr(zmap), H, W, Kp, ptr(tf), ptr(origin), N, h, w, 0, float(mesh_diameter), 0, 0,
                               ptr(raw), s))
  Ks = torch.as_tensor(np.asarray(K), dtype=torch.float, device=dev).reshape(1, 3, 3).expand(N, 3, 3)
  return BatchPoseData(rgbAs=rgbAs, rgbBs=rgbBs, depthAs=depthAs, depthBs=raw[:, 5:6], xyz_mapAs=xyz_mapAs, xyz_mapBs=xyz_mapBs, poseA=poses,
                       tf_to_crops=tf, Ks=Ks, mesh_diameters=torch.full((N,), float(mesh_diameter), device=dev), net_input=net)


class ScorePredictor:
  run_name_default = "2024-01-11-20-02-45"      # predict_score.py:120

  def __init__(self, amp=True, state_dict=None, cfg=None, weights_root=None, device='cuda'):
    self.amp = amp
    self.run_name = self.run_name_default
    if state_dict is None:
      state_dict, file_cfg = load_run_dir(self.run_name, weights_root)
      cfg = dict(file_cfg, **(cfg or {}))
    self.cfg = Cfg(cfg or {})
    self.cfg['enable_amp'] = True
    ########## Defaults, to be backward compatible (predict_score.py:131-143)
    defaults = dict(use_normal=False, use_BN=False, zfar=np.inf, c_in=4, normalize_xyz=False)
    for k, v in defaults.items():
      if k not in self.cfg:
        self.cfg[k] = v
    if 'crop_ratio' not in self.cfg or self.cfg['crop_ratio'] is None:
      self.cfg['crop_ratio'] = 1.2
    if 'input_resize' not in self.cfg:
      raise KeyError("scorer config has no 'input_resize'")
    check_network_cfg(self.cfg, state_dict, 'encoderA.0.net.0.weight', 'predict_score.py')
    self.device = torch.device(device)
    self.ctx = _lib.Context.get(self.device)
    self.model = _lib.DeviceNet(self.ctx, FP_NET_SCORE, state_dict, use_bn=bool(self.cfg['use_BN']))
    self.dataset = None
    logging.info("init done")

  def to_device(self, device):
    """Move the network to `device` (src/estimater.py:97-100 does `self.scorer.model.to(s)`): the context and the packed weights
    follow; later predict() calls run there."""
    self.device = torch.device(device)
    self.ctx = _lib.Context.get(self.device)
    self.model.to(self.device)
    return self

  @torch.inference_mode()
  def extract_features(self, rgb, depth, K, ob_in_cams, mesh=None, mesh_tensors=None, glctx=None, mesh_diameter=None):
    """Per-hypothesis part of predict (crop batch + ScoreNetMultiPair.extract_feat) -> (N,512) fp32.
    This is the shardable half; `score_tail` couples the hypotheses."""
    ctx = _ctx_of(glctx, self.device) if glctx is not None else self.ctx
    dev = torch.device('cuda', ctx.device_index)
    if mesh_tensors is None:
      mesh_tensors = make_mesh_tensors(mesh, device=dev)
    dm = _lib.device_mesh(ctx, mesh_tensors)
    poses = torch.as_tensor(ob_in_cams, dtype=torch.float, device=dev).reshape(-1, 4, 4).contiguous()
    rgb_t = torch.as_tensor(rgb, device=dev, dtype=torch.float).contiguous()
    depth_t = torch.as_tensor(depth, device=dev, dtype=torch.float).contiguous()
    H, W = depth_t.shape[:2]
    N = len(poses)
    feats = torch.empty((N, 512), dtype=torch.float, device=dev)
    Kd, Kp = k_ptr(K)
    check(lib().fp_score_predict_features(ctx.handle, self.model.handle, dm.handle, ptr(rgb_t), ptr(depth_t), H, W, Kp,
                                          float(mesh_diameter), float(self.cfg['crop_ratio']), 1 if self.cfg['normalize_xyz'] else 0,
                                          ptr(poses), N, ptr(feats), stream_ptr(dev)))
    return feats

  @torch.inference_mode()
  def forward_features(self, pose_data):
    """ScoreNetMultiPair.extract_feat on a batch made by make_crop_data_batch (score_network.py:57-80) -> (N,512)."""
    net = pose_data.net_input
    N = len(net) // 2
    feats = torch.empty((N, 512), dtype=torch.float, device=net.device)
    check(lib().fp_score_features(self.ctx.handle, self.model.handle, ptr(net), N, ptr(feats), stream_ptr(net.device)))
    return feats

  @torch.inference_mode()
  def extract_features_multi(self, objects):
    """`objects`: list of dicts(rgb, depth, K, mesh_tensors, mesh_diameter, ob_in_cams) -> (sum n, 512) features,
    one ScoreNet pass for all objects (see PoseRefinePredictor.predict_multi)."""
    arr, poses, keep = _lib.object_batches(self.ctx, objects, 'depth')
    feats = torch.empty((len(poses), 512), dtype=torch.float, device=poses.device)
    check(lib().fp_score_predict_features_multi(self.ctx.handle, self.model.handle, arr, len(objects), float(self.cfg['crop_ratio']),
                                                1 if self.cfg['normalize_xyz'] else 0, ptr(poses), ptr(feats), stream_ptr(poses.device)))
    return feats

  @torch.inference_mode()
  def extract_rows_multi(self, objects):
    """extract_features_multi writing the hypothesis-parallel job's all-gather records: (sum n, 528) rows [feature 512 | pose 16]
    (dist.py ROW), straight from the library - no concatenation pass.  `score_tail` reads such rows in place."""
    arr, poses, keep = _lib.object_batches(self.ctx, objects, 'depth')
    rows = torch.empty((len(poses), 528), dtype=torch.float, device=poses.device)
    check(lib().fp_score_predict_rows_multi(self.ctx.handle, self.model.handle, arr, len(objects), float(self.cfg['crop_ratio']),
                                            1 if self.cfg['normalize_xyz'] else 0, ptr(poses), ptr(rows), stream_ptr(poses.device)))
    return rows

  @torch.inference_mode()
  def score_tail(self, feats, L=None, score_offset=None):
    """att_cross + linear over groups of L hypotheses (score_network.py:82-88): (groups*L,512) features - or (groups*L,528) [feature |
    pose] rows, read in place - -> (groups,L) logits, (groups,) argmax; with `score_offset` also scores = logits + score_offset from
    the same launch (predict_score.py:209: + 100)."""
    feats = feats if feats.is_contiguous() else feats.contiguous()
    assert feats.shape[1] in (512, 528)
    ld = int(feats.shape[1])
    M = feats.shape[0]
    L = M if L is None else int(L)
    assert M % L == 0
    groups = M // L
    logits = torch.empty((groups, L), dtype=torch.float, device=feats.device)
    argmax = torch.empty((groups,), dtype=torch.int32, device=feats.device)
    if score_offset is None and ld == 512:
      check(lib().fp_score_tail(self.ctx.handle, self.model.handle, ptr(feats), groups, L, ptr(logits), ptr(argmax), stream_ptr(feats.device)))
      return logits, argmax
    scores = None if score_offset is None else torch.empty((groups, L), dtype=torch.float, device=feats.device)
    check(lib().fp_score_tail_scores(self.ctx.handle, self.model.handle, ptr(feats), ld, groups, L, float(score_offset or 0.0), ptr(logits),
                                     ptr(scores), ptr(argmax), stream_ptr(feats.device)))
    return (logits, argmax) if score_offset is None else (logits, argmax, scores)

  @torch.inference_mode()
  def predict(self, rgb, depth, K, ob_in_cams, normal_map=None, get_vis=False, mesh=None, mesh_tensors=None, glctx=None,
              mesh_diameter=None):
    '''
    @rgb: np array (H,W,3)
    returns (scores (N,) float tensor on the device = logits + 100, vis or None)
    '''
    logging.info(f"ob_in_cams:{np.shape(ob_in_cams)}")
    feats = self.extract_features(rgb, depth, K, ob_in_cams, mesh=mesh, mesh_tensors=mesh_tensors, glctx=glctx, mesh_diameter=mesh_diameter)
    # find_best_among_pairs runs ONE forward over all hypotheses (bs == N), so the tournament loop of
    # predict_score.py:206-212 exits in its first round: scores_global = logits + 100
    _, _, scores = self.score_tail(feats, L=len(feats), score_offset=100.0)          # scores = logits + 100, from the tail's own launch
    scores = scores.reshape(-1)
    logging.info('forward done')
    if get_vis:
      # predict_score.py:219-224: one row per hypothesis, best first (vis.py: no cv2 here, the labels come in a bitmap font)
      from .vis import score_canvas
      logging.info("get_vis...")
      pd = make_crop_data_batch(self.cfg['input_resize'], ob_in_cams, mesh, rgb, depth, K, self.cfg['crop_ratio'], mesh_diameter=mesh_diameter,
                                glctx=glctx, mesh_tensors=mesh_tensors, cfg=self.cfg)
      ids = scores.argsort(descending=True)
      return scores, score_canvas(pd, ids.cpu().numpy(), scores.cpu().numpy())
    return scores, None

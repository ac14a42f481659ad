"""PoseRefinePredictor - mirror of learning/training/predict_pose_refine.py:94-296 on the HIP library.

The whole refinement loop (crop window -> render -> observed crop -> RefineNet -> pose update, x
`iteration`) runs on the device inside one C-ABI call (fp_refine_predict); poses never visit the
host between iterations.
"""
import logging

import numpy as np
import torch

from . import _lib
from ._lib import FP_NET_REFINE, FpRefineCfg, byref, check, k_ptr, lib, ptr, stream_ptr
from .Utils import _ctx_of, make_mesh_tensors
from .config import Cfg, check_network_cfg, load_run_dir
from .pose_dataset import BatchPoseData, planar_views


def crop_net_input(ctx, dm, poses, rgb_t, geom_t, K, crop_ratio, mesh_diameter, normalize_xyz, mode, render_size=(160, 160)):
  """Crop window -> render (side A) -> observed crop (side B) into one fp16 net tensor, three launches on the current
  stream: the steps fp_refine_predict / fp_score_predict_features run internally, exposed one at a time.
  mode 0: geom = xyz_map (refiner, invalid below 1 mm); mode 1: geom = depth (scorer, invalid below 0.1 m).
  Returns (net (2N,h,w,8) fp16, tf_to_crops (N,3,3), bbox2d (N,4))."""
  dev = poses.device
  N = len(poses)
  H, W = rgb_t.shape[:2]
  h, w = int(render_size[0]), int(render_size[1])
  tf = torch.empty((N, 3, 3), device=dev, dtype=torch.float)
  bbox = torch.empty((N, 4), device=dev, dtype=torch.float)
  net = torch.empty((2 * N, h, w, 8), device=dev, dtype=torch.float16)
  Kd, Kp = k_ptr(K)
  s = stream_ptr(dev)
  nz = 1 if normalize_xyz else 0
  check(lib().fp_crop_window_tf(ctx.handle, ptr(poses), N, Kp, float(crop_ratio), float(mesh_diameter), w, h, ptr(tf), ptr(bbox), s))
  check(lib().fp_render_net(ctx.handle, dm.handle, ptr(poses), N, Kp, H, W, ptr(bbox), h, w, float(mesh_diameter), nz,
                            0.1 if mode else 0.001, ptr(net), s))
  check(lib().fp_crop_observed(ctx.handle, ptr(rgb_t), ptr(geom_t), H, W, Kp, ptr(tf), ptr(poses), N, h, w, mode, float(mesh_diameter),
                               nz, 1, ptr(net[N:]), s))
  return net, tf, bbox


@torch.inference_mode()
def make_crop_data_batch(render_size, ob_in_cams, mesh, rgb, depth, K, crop_ratio, xyz_map, normal_map=None, mesh_diameter=None, cfg=None,
                         glctx=None, mesh_tensors=None, dataset=None):
  """predict_pose_refine.py:24-89 including dataset.transform_batch (h5_dataset.py:79-127): the batch one refinement
  iteration feeds to RefineNet.  `pose_data.net_input` is what the HIP network reads; rgbAs / xyz_mapAs / rgbBs /
  xyz_mapBs are its float32 planar views (fp16 precision).  `dataset` is accepted and ignored: its transform is fused
  into the kernels."""
  cfg = cfg if cfg is not None else {}
  ctx = _ctx_of(glctx)
  dev = torch.device('cuda', ctx.device_index)
  if mesh_tensors is None:
    mesh_tensors = make_mesh_tensors(mesh, device=dev)
  dm = _lib.device_mesh(ctx, mesh_tensors)
  poses = torch.as_tensor(ob_in_cams, device=dev, dtype=torch.float).reshape(-1, 4, 4).contiguous()
  rgb_t = torch.as_tensor(rgb, device=dev, dtype=torch.float).contiguous()
  xyz_t = torch.as_tensor(xyz_map, device=dev, dtype=torch.float).contiguous()
  assert xyz_t.shape[:2] == rgb_t.shape[:2] == tuple(depth.shape[:2])
  net, tf, bbox = crop_net_input(ctx, dm, poses, rgb_t, xyz_t, K, crop_ratio, mesh_diameter, cfg.get('normalize_xyz', False), 0, render_size)
  rgbAs, xyz_mapAs, rgbBs, xyz_mapBs = planar_views(net)
  N = len(poses)
  normalAs = normalBs = None
  if cfg.get('use_normal', False):
    normalAs, normalBs = crop_normals(ctx, dm, poses, tf, bbox, K, rgb_t.shape[:2], render_size, normal_map)
  Ks = torch.as_tensor(np.asarray(K), dtype=torch.float, device=dev).reshape(1, 3, 3).expand(N, 3, 3)
  return BatchPoseData(rgbAs=rgbAs, rgbBs=rgbBs, normalAs=normalAs, normalBs=normalBs, xyz_mapAs=xyz_mapAs, xyz_mapBs=xyz_mapBs, poseA=poses,
                       tf_to_crops=tf, Ks=Ks, mesh_diameters=torch.full((N,), float(mesh_diameter), device=dev), net_input=net)


def crop_normals(ctx, dm, poses, tf, bbox, K, frame_hw, render_size, normal_map):
  """The use_normal branch of predict_pose_refine.py:49,58,74-76.  normalAs: the camera-frame normals rendered INTO the
  crop (nvdiffrast_render(get_normal=True, output_size=input_resize, bbox2d=...), src/Utils.py:193-197) and then warped
  by tf_to_crops once more, nearest - the reference does that unconditionally for the normals (its rgb / xyz of side A
  are only re-warped when the sizes differ), so normalAs is the crop of a crop; kept as it is.  normalBs: the frame's
  normal map cropped by tf_to_crops, nearest.  Neither is touched by transform_batch (h5_dataset.py:79-127) and neither
  is fed to RefineNet (predict_pose_refine.py:186-187 builds A and B from rgb and xyz_map only): they ride in
  BatchPoseData.  A missing normal_map fails where the reference's torch.as_tensor(None) does."""
  dev = poses.device
  N = len(poses)
  H, W = int(frame_hw[0]), int(frame_hw[1])
  h, w = int(render_size[0]), int(render_size[1])
  Kd, Kp = k_ptr(K)
  s = stream_ptr(dev)
  normal_t = torch.as_tensor(normal_map, dtype=torch.float, device=dev).contiguous()     # (H,W,3); None raises here as in the reference
  assert normal_t.shape == (H, W, 3), f'normal_map is {tuple(normal_t.shape)}, the frame is {(H, W)}'
  rendered = torch.empty((N, h, w, 3), device=dev, dtype=torch.float)
  check(lib().fp_render(ctx.handle, dm.handle, ptr(poses), N, Kp, H, W, ptr(bbox), h, w, 1, 0.8, 0.5, None, None, ptr(rendered), None, s))
  normalAs = torch.empty((N, 3, h, w), device=dev, dtype=torch.float)
  normalBs = torch.empty((N, 3, h, w), device=dev, dtype=torch.float)
  check(lib().fp_warp_nearest(ctx.handle, ptr(rendered), N, h, w, 3, ptr(tf), N, h, w, ptr(normalAs), s))
  check(lib().fp_warp_nearest(ctx.handle, ptr(normal_t), 1, H, W, 3, ptr(tf), N, h, w, ptr(normalBs), s))
  return normalAs, normalBs


class PoseRefinePredictor:
  run_name_default = "2023-10-28-18-33-37"      # predict_pose_refine.py:97

  def __init__(self, state_dict=None, cfg=None, weights_root=None, device='cuda'):
    """With no arguments: loads weights/<run_name>/{model_best.pth,config.yml} like the reference
    (predict_pose_refine.py:94-141).  `state_dict` / `cfg` inject parameters directly (the weights
    are not distributed with the repositories; tests and bench.py use seeded synthetic ones)."""
    logging.info("welcome")
    self.amp = True
    self.run_name = self.run_name_default
    if state_dict is None:
      state_dict, file_cfg = load_run_dir(self.run_name, weights_root)
      cfg = dict(file_cfg, **(cfg or {}))
    self.cfg = Cfg(cfg or {})
    self.cfg['enable_amp'] = True
    ########## Defaults, to be backward compatible (predict_pose_refine.py:107-131)
    defaults = dict(use_normal=False, use_mask=False, use_BN=False, c_in=4, n_view=1, trans_rep='tracknet', rot_rep='axis_angle',
                    zfar=3, normalize_xyz=False, normal_uint8=False)
    for k, v in defaults.items():
      if k not in self.cfg:
        self.cfg[k] = v
    if 'crop_ratio' not in self.cfg or self.cfg['crop_ratio'] is None:
      self.cfg['crop_ratio'] = 1.2
    if isinstance(self.cfg['zfar'], str) and 'inf' in self.cfg['zfar'].lower():
      self.cfg['zfar'] = np.inf
    for k in ('input_resize', 'trans_normalizer', 'rot_normalizer'):
      if k not in self.cfg:
        raise KeyError(f"refiner config has no '{k}' (the reference reads it without a default)")
    check_network_cfg(self.cfg, state_dict, 'encodeA.0.net.0.weight', 'predict_pose_refine.py')
    self.device = torch.device(device)
    self.ctx = _lib.Context.get(self.device)
    self.model = _lib.DeviceNet(self.ctx, FP_NET_REFINE, state_dict, use_bn=bool(self.cfg['use_BN']))
    want = 3 if self.cfg['rot_rep'] == 'axis_angle' else 6
    if self.model.rot_dim != want:
      raise RuntimeError(f"rot_rep={self.cfg['rot_rep']} needs a {want}-d rotation head, state_dict has {self.model.rot_dim}")
    self.dataset = None
    logging.info("init done")
    self.last_trans_update = None
    self.last_rot_update = None

  def to_device(self, device):
    """Move the network to `device` (src/estimater.py:97-100 does `self.refiner.model.to(s)`): the context and the packed weights
    follow; later predict() calls run there."""
    self.device = torch.device(device)
    self.ctx = _lib.Context.get(self.device)
    self.model.to(self.device)
    return self

  def _c_cfg(self):
    """fp_refine_cfg for the C-ABI from the reference's config keys (predict_pose_refine.py:195-231)."""
    c = FpRefineCfg()
    c.crop_ratio = float(self.cfg['crop_ratio'])
    c.normalize_xyz = 1 if self.cfg['normalize_xyz'] else 0
    # 0: raw head output, 1: tanh * trans_normalizer (tracknet without normalize_xyz), 2: deepim (predict_pose_refine.py:195-215)
    c.trans_rep_tanh = 2 if self.cfg['trans_rep'] == 'deepim' else (1 if (self.cfg['trans_rep'] == 'tracknet' and not self.cfg['normalize_xyz']) else 0)
    tn = self.cfg['trans_normalizer']
    tn = [float(tn)] * 3 if isinstance(tn, (float, int)) else [float(x) for x in tn]
    for i in range(3):
      c.trans_normalizer[i] = tn[i]
    c.rot_normalizer = float(self.cfg['rot_normalizer'])
    return c

  @torch.inference_mode()
  def forward(self, pose_data):
    """RefineNet on a batch made by make_crop_data_batch (predict_pose_refine.py:186-193, `self.model(A, B)`):
    returns dict(trans (N,3), rot (N,3|6)) of raw head outputs."""
    net = pose_data.net_input
    N = len(net) // 2
    trans = torch.empty((N, 3), device=net.device, dtype=torch.float)
    rot = torch.empty((N, self.model.rot_dim), device=net.device, dtype=torch.float)
    check(lib().fp_refine_forward(self.ctx.handle, self.model.handle, ptr(net), N, ptr(trans), ptr(rot), stream_ptr(net.device)))
    return dict(trans=trans, rot=rot)

  @torch.inference_mode()
  def update_poses(self, poseA, trans, rot, mesh_diameter, tf_to_crops=None, K=None):
    """predict_pose_refine.py:195-231: head outputs -> B_in_cam (egocentric delta applied to poseA).  trans_rep='deepim'
    (:201-215) also needs the pass's crop transforms (pose_data.tf_to_crops) and the intrinsics."""
    c = self._c_cfg()
    poseA = poseA.contiguous()
    out = torch.empty_like(poseA)
    tn = np.array([c.trans_normalizer[i] for i in range(3)], dtype=np.float32)
    scale = float(mesh_diameter) / 2 if c.normalize_xyz else 1.0
    if c.trans_rep_tanh == 2:
      if tf_to_crops is None or K is None:
        raise ValueError("trans_rep='deepim': update_poses needs tf_to_crops and K")
      tf = torch.as_tensor(tf_to_crops, device=poseA.device, dtype=torch.float).reshape(-1, 9).contiguous()
      Kd, Kp = k_ptr(K)
      check(lib().fp_pose_update_deepim(self.ctx.handle, ptr(poseA), ptr(trans), ptr(rot), len(poseA), self.model.rot_dim, ptr(tf), Kp,
                                        float(self.cfg['input_resize'][0]), c.rot_normalizer, scale, ptr(out), stream_ptr(poseA.device)))
      return out
    check(lib().fp_pose_update(self.ctx.handle, ptr(poseA), ptr(trans), ptr(rot), len(poseA), self.model.rot_dim, c.trans_rep_tanh,
                               tn.ctypes.data, c.rot_normalizer, scale, ptr(out), stream_ptr(poseA.device)))
    return out

  @staticmethod
  def _shares_translation(ob_in_cams, shared_translation):
    """True when every hypothesis has the translation of the first one, which makes the observed side of the FIRST iteration one crop
    (FP_REFINE_SHARED_TRANSLATION in include/foundationpose_amd.h).  `shared_translation` True / False: the caller's word (the estimator
    knows: its hypotheses are a rotation grid around one guessed centre, src/estimater.py:126-135); None: host arrays - what the reference
    passes - are compared here, device tensors are not (that would be a synchronisation per call)."""
    if shared_translation is not None:
      return bool(shared_translation)
    if torch.is_tensor(ob_in_cams) and ob_in_cams.is_cuda:
      return False
    t = np.asarray(ob_in_cams, dtype=np.float32).reshape(-1, 4, 4)[:, :3, 3]
    return len(t) > 1 and bool((t == t[0]).all())

  @torch.inference_mode()
  def predict_multi(self, objects, iteration=5):
    """Several objects in one pass (BASELINE configs[3]; a rank's slices in the sharded job): `objects` is a list of
    dicts(rgb, xyz_map, K, mesh_tensors, mesh_diameter, ob_in_cams [, shared_translation]).  Render / crop run per object, RefineNet once
    on the concatenated hypotheses.  Returns the refined poses concatenated in object order."""
    arr, poses, keep = _lib.object_batches(self.ctx, objects, 'xyz_map')
    c = self._c_cfg()
    N = len(poses)
    trans = torch.empty((N, 3), device=poses.device, dtype=torch.float)
    rot = torch.empty((N, self.model.rot_dim), device=poses.device, dtype=torch.float)
    flags = _lib.FP_REFINE_SHARED_TRANSLATION if all(self._shares_translation(ob['ob_in_cams'], ob.get('shared_translation')) for ob in objects) else 0
    check(lib().fp_refine_predict_multi_flags(self.ctx.handle, self.model.handle, arr, len(objects), byref(c), ptr(poses), int(iteration),
                                              ptr(trans), ptr(rot), flags, stream_ptr(poses.device)))
    self.last_trans_update, self.last_rot_update = trans, rot
    return poses

  @torch.inference_mode()
  def predict(self, rgb, depth, K, ob_in_cams, xyz_map, normal_map=None, get_vis=False, mesh=None, mesh_tensors=None, glctx=None,
              mesh_diameter=None, iteration=5, shared_translation=None):
    '''
    @rgb: np array (H,W,3)
    @ob_in_cams: np array (N,4,4)
    @shared_translation: see _shares_translation (an extension; the results do not depend on it)
    returns (B_in_cams (N,4,4) float tensor on the device, vis or None)
    '''
    logging.info(f'ob_in_cams:{np.shape(ob_in_cams)}')
    if self.cfg['rot_rep'] not in ('axis_angle', '6d'):
      raise RuntimeError
    # predict_pose_refine.py:161-163: the normal map only exists under use_normal.  There the reference warps it in every iteration
    # into pose_data.normalAs / normalBs, which predict() itself never reads (A and B are rgb + xyz_map, :186-187): the refined poses do
    # not depend on it, so the fused passes leave those warps out; make_crop_data_batch (the stepping API, get_vis) makes them.
    if not self.cfg['use_normal']:
      normal_map = None
    elif normal_map is None:
      raise RuntimeError('Could not infer dtype of NoneType (use_normal=True needs normal_map: predict_pose_refine.py:75 hands it to torch.as_tensor)')
    elif tuple(np.shape(normal_map)) != tuple(np.shape(rgb)):
      raise RuntimeError(f'normal_map {np.shape(normal_map)} does not match the frame {np.shape(rgb)}')
    ctx = _ctx_of(glctx, self.device) if glctx is not None else self.ctx
    dev = torch.device('cuda', ctx.device_index)
    if mesh_tensors is None:
      mesh_tensors = make_mesh_tensors(mesh, device=dev)
    dm = _lib.device_mesh(ctx, mesh_tensors)
    poses = torch.as_tensor(ob_in_cams, device=dev, dtype=torch.float).reshape(-1, 4, 4).contiguous().clone()
    N = len(poses)
    rgb_t = torch.as_tensor(rgb, device=dev, dtype=torch.float).contiguous()
    xyz_t = torch.as_tensor(xyz_map, device=dev, dtype=torch.float).contiguous()
    H, W = rgb_t.shape[:2]
    assert xyz_t.shape[:2] == (H, W)
    c = self._c_cfg()
    trans = torch.empty((N, 3), device=dev, dtype=torch.float)
    rot = torch.empty((N, self.model.rot_dim), device=dev, dtype=torch.float)
    Kd, Kp = k_ptr(K)
    ob = _lib.FpObjectBatch()
    ob.mesh, ob.d_rgb, ob.d_geom, ob.H, ob.W, ob.K, ob.mesh_diameter, ob.n = dm.handle, rgb_t.data_ptr(), xyz_t.data_ptr(), H, W, Kd.ctypes.data, float(mesh_diameter), N
    flags = _lib.FP_REFINE_SHARED_TRANSLATION if self._shares_translation(ob_in_cams, shared_translation) else 0
    check(lib().fp_refine_predict_multi_flags(ctx.handle, self.model.handle, byref(ob), 1, byref(c), ptr(poses), int(iteration), ptr(trans), ptr(rot),
                                              flags, stream_ptr(dev)))
    self.last_trans_update = trans
    self.last_rot_update = rot
    if get_vis:
      # predict_pose_refine.py:241-293: crops at the start poses and at the refined ones, side by side (vis.py: no cv2 here, the labels come in a bitmap font)
      from .vis import refine_canvas
      logging.info("get_vis...")
      kw = dict(mesh_diameter=mesh_diameter, cfg=self.cfg, glctx=glctx, mesh_tensors=mesh_tensors, normal_map=normal_map)
      before = make_crop_data_batch(self.cfg['input_resize'], ob_in_cams, mesh, rgb, depth, K, self.cfg['crop_ratio'], xyz_map, **kw)
      after = make_crop_data_batch(self.cfg['input_resize'], poses, mesh, rgb, depth, K, self.cfg['crop_ratio'], xyz_map, **kw)
      return poses, refine_canvas(before, after)
    return poses, None

"""Generate tests/golden/reference_outputs.npz by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  The
reference's modules import a dozen third-party packages at module scope that are absent here
(trimesh, open3d, nvdiffrast, ...); inert empty modules are registered for those names so that the
import statements succeed - none of their attributes is used by the code exercised below (the two
nn.Modules, their building blocks and a handful of pure numpy/torch helpers).  Everything that
*would* need those packages (rasteriser, kornia warps, pytorch3d so3, Warp kernels) is NOT covered
by these vectors and stays "parity unpinned" (oracle/__init__.py).

Inputs are not stored: tests regenerate them from the same seeds (numpy RandomState streams are
frozen by numpy's compatibility policy) and the same foundationpose_amd.synthetic state_dicts.

usage:  python tests/golden/gen_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
sys.path.insert(0, REPO)

ABSENT = ['trimesh', 'imageio', 'pytorch3d', 'pytorch3d.transforms', 'pytorch3d.renderer', 'pytorch3d.renderer.mesh',
          'pytorch3d.renderer.mesh.rasterize_meshes', 'pytorch3d.renderer.mesh.shader', 'pytorch3d.renderer.mesh.textures',
          'pytorch3d.structures', 'nvdiffrast', 'nvdiffrast.torch', 'torchvision', 'open3d', 'cv2', 'transformations',
          'ruamel', 'ruamel.yaml', 'kornia', 'warp', 'omegaconf', 'h5py', 'joblib', 'PIL', 'matplotlib', 'matplotlib.pyplot']


class _Inert(types.ModuleType):
  def __getattr__(self, k):
    if k.startswith('__'):
      raise AttributeError(k)
    return _Inert(self.__name__ + '.' + k)

  def __call__(self, *a, **k):
    return _Inert('call')


def import_reference():
  for n in ABSENT:
    try:
      importlib.import_module(n)
    except Exception:
      m = _Inert(n)
      m.__path__ = []
      m.__all__ = []
      sys.modules[n] = m
  sys.path.insert(0, REF)
  sys.path.insert(0, os.path.join(REF, 'src'))
  sys.path.insert(0, os.path.join(REF, 'learning', 'models'))
  import src.Utils as U                                        # noqa
  from learning.models.refine_network import RefineNet          # noqa
  from learning.models.score_network import ScoreNetMultiPair   # noqa
  from learning.models.network_modules import PositionalEmbedding  # noqa
  from learning.datasets.pose_dataset import BatchPoseData      # noqa
  import src.estimater as E                                     # noqa
  return U, RefineNet, ScoreNetMultiPair, PositionalEmbedding, BatchPoseData, E


class Cfg(dict):
  __getattr__ = dict.__getitem__


def main():
  from foundationpose_amd import synthetic as S
  from tests.util import net_inputs
  U, RefineNet, ScoreNetMultiPair, PositionalEmbedding, BatchPoseData, E = import_reference()
  torch.set_num_threads(8)
  out = {}

  # ---- RefineNet (use_BN, c_in 6, axis_angle) ------------------------------------------------
  taps = {}

  def hook(name):
    def f(mod, inp, o):
      taps[name] = o.detach()
    return f
  cfg = Cfg(use_BN=True, rot_rep='axis_angle')
  net = RefineNet(cfg=cfg, c_in=6).eval()
  sd = S.make_refine_state_dict(seed=0)
  missing = net.load_state_dict(sd, strict=True)
  out['refine_keys_ok'] = np.array(1)
  net.encodeA[3].register_forward_hook(hook('encA3'))
  net.encodeAB[4].register_forward_hook(hook('encAB4'))
  A, B = net_inputs(11, 8)
  with torch.no_grad():
    o = net(A, B)
  out['refine_trans'] = o['trans'].numpy()
  out['refine_rot'] = o['rot'].numpy()
  out['refine_encA3_sub'] = taps['encA3'][:, ::16, ::8, ::8].numpy()
  out['refine_encAB4_sub'] = taps['encAB4'][:, ::64, ::4, ::4].numpy()
  # the same module under the reference's own inference precision (torch.cuda.amp.autocast, predict_pose_refine.py:190): fp16 autocast,
  # here on the CPU backend.  What the HIP path is allowed to lose against the fp32 outputs is measured against what the reference's
  # fp16 path loses (tests/test_gpu_pipeline.py::test_no_narrower_than_the_reference_autocast).
  with torch.no_grad(), torch.autocast('cpu', dtype=torch.float16):
    o16 = net(A, B)
  out['refine_trans_ac16'] = o16['trans'].float().numpy()
  out['refine_rot_ac16'] = o16['rot'].float().numpy()
  out['refine_encA3_sub_ac16'] = taps['encA3'][:, ::16, ::8, ::8].float().numpy()
  out['refine_encAB4_sub_ac16'] = taps['encAB4'][:, ::64, ::4, ::4].float().numpy()

  # ---- RefineNet without BN, 6d rotation head (non-default branches) --------------------------
  cfg2 = Cfg(use_BN=False, rot_rep='6d')
  net2 = RefineNet(cfg=cfg2, c_in=6).eval()
  sd2 = S.make_refine_state_dict(seed=2, use_bn=False, rot_out_dim=6)
  net2.load_state_dict(sd2, strict=True)
  A2, B2 = net_inputs(12, 4)
  with torch.no_grad():
    o2 = net2(A2, B2)
  out['refine_nobn_trans'] = o2['trans'].numpy()
  out['refine_nobn_rot'] = o2['rot'].numpy()

  # ---- ScoreNetMultiPair ------------------------------------------------------------------------
  snet = ScoreNetMultiPair(cfg=Cfg(use_BN=True), c_in=6).eval()
  ssd = S.make_score_state_dict(seed=1)
  snet.load_state_dict(ssd, strict=True)
  A3, B3 = net_inputs(13, 8)
  snet.encoderAB[4].register_forward_hook(hook('s_encAB4'))
  with torch.no_grad():
    feats = snet.extract_feat(A3, B3)
    so = snet(A3, B3, L=8)
    so2 = snet(A3, B3, L=4)        # two "objects" of four hypotheses each (bs=2 groups)
  out['score_feats'] = feats.numpy()
  out['score_logit_L8'] = so['score_logit'].numpy()
  out['score_logit_L4'] = so2['score_logit'].numpy()
  out['score_encAB4_sub'] = taps['s_encAB4'][:, ::64, ::4, ::4].numpy()
  with torch.no_grad(), torch.autocast('cpu', dtype=torch.float16):          # predict_score.py:193
    out['score_feats_ac16'] = snet.extract_feat(A3, B3).float().numpy()
  out['score_encAB4_sub_ac16'] = taps['s_encAB4'][:, ::64, ::4, ::4].float().numpy()

  # ---- PositionalEmbedding buffer -----------------------------------------------------------------
  pe = PositionalEmbedding(d_model=512, max_len=400).pe
  out['pe_sub'] = pe[0, ::37, ::61].numpy()
  out['pe_shape'] = np.array(pe.shape)

  # ---- pure helpers -------------------------------------------------------------------------------
  K = S.YCB_K
  out['proj_y_down'] = U.projection_matrix_from_intrinsics(K, height=480, width=640, znear=0.001, zfar=100)
  out['proj_y_up'] = U.projection_matrix_from_intrinsics(K, height=480, width=640, znear=0.001, zfar=100, window_coords='y_up')
  rs = np.random.RandomState(21)
  depth = rs.uniform(0.3, 1.5, (12, 16)).astype(np.float32)
  depth[rs.uniform(size=depth.shape) < 0.2] = 0
  out['d2x_depth'] = depth
  out['d2x_xyz'] = U.depth2xyzmap(depth, K)
  pts = torch.from_numpy(rs.randn(7, 3).astype(np.float32))
  tf = torch.from_numpy(rs.randn(5, 4, 4).astype(np.float32))
  out['tp_pts'] = pts.numpy(); out['tp_tf'] = tf.numpy()
  out['tp_out'] = U.transform_pts(pts, tf).numpy()
  out['td_out'] = U.transform_dirs(pts, tf).numpy()
  out['homo_out'] = U.to_homo_torch(pts).numpy()
  Ain = torch.from_numpy(rs.randn(3, 4, 4).astype(np.float32))
  td = torch.from_numpy(rs.randn(3, 3).astype(np.float32))
  rd = torch.from_numpy(rs.randn(3, 3, 3).astype(np.float32))
  out['ego_A'] = Ain.numpy(); out['ego_td'] = td.numpy(); out['ego_rd'] = rd.numpy()
  out['ego_out'] = U.egocentric_delta_pose_to_pose(Ain, td, rd).numpy()
  t2, r2 = U.pose_to_egocentric_delta_pose(Ain, torch.from_numpy(out['ego_out']))
  out['ego_back_t'] = t2.numpy(); out['ego_back_r'] = r2.numpy()
  # guess_translation (src/estimater.py:137-156) called unbound with a dummy self
  mask = np.zeros((12, 16), bool); mask[3:9, 4:13] = True
  dummy = types.SimpleNamespace(debug=0)
  out['gt_mask'] = mask
  out['gt_center'] = E.FoundationPose.guess_translation(dummy, depth=depth, mask=mask, K=K)
  out['gt_center_empty'] = E.FoundationPose.guess_translation(dummy, depth=depth, mask=np.zeros_like(mask), K=K)
  out['glcam_in_cvcam'] = U.glcam_in_cvcam
  # BatchPoseData.select_by_indices
  bp = BatchPoseData(rgbAs=torch.arange(24.).reshape(4, 6), poseA=torch.arange(8.).reshape(4, 2))
  sel = bp.select_by_indices(torch.tensor([2, 0]))
  out['bpd_rgbAs'] = sel.rgbAs.numpy(); out['bpd_poseA'] = sel.poseA.numpy()

  # ---- src/transform.py PoseTransformer (pure python) ---------------------------------------------
  sys.path.insert(0, os.path.join(REF, 'src'))
  import contextlib, io
  from transform import PoseTransformer as RefPT
  mats = []
  for i in range(6):
    M = np.eye(4); M[:3, :3] = S.random_rotation(rs); M[:3, 3] = rs.randn(3)
    mats.append(M)
  for sgn in (1.0, -1.0):     # gimbal lock: R[2,0] = -+1
    c, s_ = np.cos(0.7), np.sin(0.7)
    M = np.eye(4); M[:3, :3] = np.array([[0, -s_, c * sgn], [0, c, s_ * sgn], [-sgn, 0, 0]]); M[:3, 3] = [0.1, -0.2, 0.9]
    mats.append(M)
  out['pt_mats'] = np.asarray(mats)
  with contextlib.redirect_stdout(io.StringIO()):
    out['pt_inch_deg'] = np.asarray([RefPT().transform_pose(M) for M in mats])
    out['pt_m_rad'] = np.asarray([RefPT(to_inches=False, to_degrees=False).transform_pose(M) for M in mats])

  path = os.path.join(HERE, 'reference_outputs.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == '__main__':
  main()

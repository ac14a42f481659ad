"""The CPU oracle against vectors produced by the reference's own code (tests/golden/gen_golden.py).
Inputs are regenerated from the seeds the generator used."""
import numpy as np
import torch

from foundationpose_amd import synthetic as S
from oracle import geometry as G
from oracle import nets
from tests.util import net_inputs


def test_refine_net_matches_reference_module(golden):
  sd = S.make_refine_state_dict(seed=0)
  A, B = net_inputs(11, 8)
  taps = {}
  o = nets.refine_forward(sd, A, B, use_bn=True, taps=taps)
  # fp32 CPU, different op grouping (functional vs nn.Module): 1e-5 absolute on O(0.1) outputs
  np.testing.assert_allclose(o['trans'].numpy(), golden['refine_trans'], atol=2e-5, rtol=1e-4)
  np.testing.assert_allclose(o['rot'].numpy(), golden['refine_rot'], atol=2e-5, rtol=1e-4)
  np.testing.assert_allclose(taps['encA3'][:, ::16, ::8, ::8].numpy(), golden['refine_encA3_sub'], atol=1e-4, rtol=1e-4)
  np.testing.assert_allclose(taps['encAB4'][:, ::64, ::4, ::4].numpy(), golden['refine_encAB4_sub'], atol=1e-4, rtol=1e-4)
  # the vectors are able to fail: the reference's outputs differ from sample to sample by far more than the tolerance
  assert golden['refine_trans'].std(0).min() > 100 * 2e-5 and golden['refine_rot'].std(0).min() > 100 * 2e-5


def test_refine_net_no_bn_6d(golden):
  sd = S.make_refine_state_dict(seed=2, use_bn=False, rot_out_dim=6)
  A, B = net_inputs(12, 4)
  o = nets.refine_forward(sd, A, B, use_bn=False)
  np.testing.assert_allclose(o['trans'].numpy(), golden['refine_nobn_trans'], atol=2e-5, rtol=1e-4)
  np.testing.assert_allclose(o['rot'].numpy(), golden['refine_nobn_rot'], atol=2e-5, rtol=1e-4)


def test_score_net_matches_reference_module(golden):
  sd = S.make_score_state_dict(seed=1)
  A, B = net_inputs(13, 8)
  feats = nets.score_extract_feat(sd, A, B, use_bn=True)
  np.testing.assert_allclose(feats.numpy(), golden['score_feats'], atol=1e-4, rtol=1e-4)
  np.testing.assert_allclose(nets.score_tail(sd, feats, 8).numpy(), golden['score_logit_L8'], atol=1e-4, rtol=1e-4)
  np.testing.assert_allclose(nets.score_tail(sd, feats, 4).numpy(), golden['score_logit_L4'], atol=1e-4, rtol=1e-4)
  assert int(nets.score_tail(sd, feats, 8).argmax()) == int(golden['score_logit_L8'].argmax())
  assert (nets.score_tail(sd, feats, 4).argmax(-1).numpy() == golden['score_logit_L4'].argmax(-1)).all()


def test_positional_embedding(golden):
  pe = S.positional_embedding()
  assert tuple(pe.shape) == tuple(golden['pe_shape'])
  np.testing.assert_array_equal(pe[0, ::37, ::61].numpy(), golden['pe_sub'])


def test_projection_matrix(golden):
  for mode, key in (('y_down', 'proj_y_down'), ('y_up', 'proj_y_up')):
    P = G.projection_matrix_from_intrinsics(S.YCB_K, 480, 640, 0.001, 100, window_coords=mode)
    np.testing.assert_allclose(P, golden[key], rtol=0, atol=1e-15)
  np.testing.assert_array_equal(G.glcam_in_cvcam, golden['glcam_in_cvcam'])


def test_depth2xyzmap(golden):
  xyz = G.depth2xyzmap(golden['d2x_depth'], S.YCB_K)
  np.testing.assert_array_equal(xyz, golden['d2x_xyz'])
  # batch form agrees on valid pixels (src/Utils.py:420-438)
  xb = G.depth2xyzmap_batch(torch.from_numpy(golden['d2x_depth'])[None], torch.as_tensor(S.YCB_K, dtype=torch.float32)[None], zfar=np.inf)[0]
  np.testing.assert_allclose(xb.numpy(), golden['d2x_xyz'], atol=1e-6)


def test_pose_algebra(golden):
  pts, tf = torch.from_numpy(golden['tp_pts']), torch.from_numpy(golden['tp_tf'])
  np.testing.assert_allclose(G.transform_pts(pts, tf).numpy(), golden['tp_out'], atol=1e-6)
  np.testing.assert_allclose(G.transform_dirs(pts, tf).numpy(), golden['td_out'], atol=1e-6)
  np.testing.assert_array_equal(G.to_homo_torch(pts).numpy(), golden['homo_out'])
  A, td, rd = (torch.from_numpy(golden[k]) for k in ('ego_A', 'ego_td', 'ego_rd'))
  out = G.egocentric_delta_pose_to_pose(A, td, rd)
  np.testing.assert_allclose(out.numpy(), golden['ego_out'], atol=1e-6)
  t2, r2 = G.pose_to_egocentric_delta_pose(A, out)
  np.testing.assert_allclose(t2.numpy(), golden['ego_back_t'], atol=1e-6)
  np.testing.assert_allclose(r2.numpy(), golden['ego_back_r'], atol=1e-5)


def test_guess_translation(golden):
  c = G.guess_translation(golden['d2x_depth'], golden['gt_mask'], S.YCB_K)
  np.testing.assert_allclose(c, golden['gt_center'], rtol=1e-12)
  c0 = G.guess_translation(golden['d2x_depth'], np.zeros_like(golden['gt_mask']), S.YCB_K)
  np.testing.assert_array_equal(c0, golden['gt_center_empty'])

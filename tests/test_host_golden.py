"""CPU: the host-side helpers of the product package against vectors produced by the
reference's own functions (tests/golden/gen_golden.py).  No kernel is called; the numpy / torch-CPU branches run."""
import types

import numpy as np
import torch

from foundationpose_amd import Utils as U
from foundationpose_amd import synthetic as S


def test_projection_and_camera_convention(golden):
  for mode, key in (('y_down', 'proj_y_down'), ('y_up', 'proj_y_up')):
    P = U.projection_matrix_from_intrinsics(S.YCB_K, height=480, width=640, znear=0.001, zfar=100, window_coords=mode)
    np.testing.assert_allclose(P, golden[key], rtol=0, atol=1e-15)
  np.testing.assert_array_equal(U.glcam_in_cvcam, golden['glcam_in_cvcam'])


def test_depth2xyzmap_numpy_branch(golden):
  np.testing.assert_array_equal(U.depth2xyzmap(golden['d2x_depth'], S.YCB_K), golden['d2x_xyz'])


def test_pose_algebra(golden):
  pts, tf = torch.from_numpy(golden['tp_pts']), torch.from_numpy(golden['tp_tf'])
  np.testing.assert_allclose(U.transform_pts(pts, tf).numpy(), golden['tp_out'], atol=1e-6)
  np.testing.assert_allclose(U.transform_dirs(pts, tf).numpy(), golden['td_out'], atol=1e-6)
  np.testing.assert_array_equal(U.to_homo_torch(pts).numpy(), golden['homo_out'])
  A, td, rd = (torch.from_numpy(golden[k]) for k in ('ego_A', 'ego_td', 'ego_rd'))
  out = U.egocentric_delta_pose_to_pose(A, td, rd)
  np.testing.assert_allclose(out.numpy(), golden['ego_out'], atol=1e-6)
  t2, r2 = U.pose_to_egocentric_delta_pose(A, out)
  np.testing.assert_allclose(t2.numpy(), golden['ego_back_t'], atol=1e-6)
  np.testing.assert_allclose(r2.numpy(), golden['ego_back_r'], atol=1e-5)


def test_guess_translation_host_branch(golden):
  """src/estimater.py:137-156 called the way the golden generator called the reference: unbound, dummy self."""
  from foundationpose_amd.estimater import FoundationPose
  me = types.SimpleNamespace(debug=0)
  c = FoundationPose.guess_translation(me, depth=golden['d2x_depth'], mask=golden['gt_mask'], K=S.YCB_K)
  np.testing.assert_allclose(c, golden['gt_center'], rtol=1e-12)
  c0 = FoundationPose.guess_translation(me, depth=golden['d2x_depth'], mask=np.zeros_like(golden['gt_mask']), K=S.YCB_K)
  np.testing.assert_array_equal(c0, golden['gt_center_empty'])
  # a mask over pixels with no usable depth
  c1 = FoundationPose.guess_translation(me, depth=np.zeros_like(golden['d2x_depth']), mask=golden['gt_mask'], K=S.YCB_K)
  np.testing.assert_array_equal(c1, np.zeros(3))

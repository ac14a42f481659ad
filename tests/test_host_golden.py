"""CPU: the host-side helpers of the product package against vectors produced by the
reference's own functions (tests/golden/gen_golden.py).  No kernel is called; the numpy / torch-CPU branches run
(depth2xyzmap runs the device kernel for numpy input too: its golden check is tests/test_gpu_kernels.py)."""
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


def _rot_z(deg):
  a = np.deg2rad(deg)
  M = np.eye(4)
  M[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
  return M


def _geodesic_deg(Ra, Rb):
  return np.degrees(np.arccos(np.clip((np.trace(Ra @ Rb.T) - 1) / 2, -1, 1)))


def test_cluster_poses_native_with_symmetries():
  """mycpp.cluster_poses (pybind_api.cpp:24-68) as native host code in the library (no kernel runs)
  against the oracle's restatement, for the identity, a 4-fold and a 12-fold z symmetry, with and without the translation
  gate; plus the algorithm's own invariants checked in float64 (kept poses are a subsequence starting with pose 0,
  every later kept pose is >= the threshold from all earlier kept ones under every symmetry, every dropped pose is
  within it of an earlier kept one).  The reference has no test for this routine: parity unpinned beyond the restatement."""
  from oracle import geometry as G
  rs = np.random.RandomState(4)
  poses = []
  for i in range(120):
    M = np.eye(4)
    M[:3, :3] = S.random_rotation(rs)
    M[:3, 3] = rs.uniform(-0.05, 0.05, 3)
    poses.append(M.astype(np.float32))
  poses[7] = poses[3].copy()                                       # an exact duplicate
  poses[9][:3, :3] = (poses[2] @ _rot_z(90).astype(np.float32))[:3, :3]; poses[9][:3, 3] = poses[2][:3, 3]   # equal to pose 2 under the 4-fold symmetry
  for sym_deg, angle, dist in ((None, 30, 99999), ((0, 90, 180, 270), 30, 99999), (tuple(range(0, 360, 30)), 20, 99999), ((0, 180), 45, 0.04)):
    sym = np.eye(4)[None] if sym_deg is None else np.stack([_rot_z(d) for d in sym_deg])
    got = U.cluster_poses(angle, dist, poses, sym.astype(np.float32))
    ref = G.cluster_poses(angle, dist, poses, sym)
    assert len(got) == len(ref) and len(got) < len(poses)
    for a, b in zip(got, ref):
      np.testing.assert_array_equal(a, b)
    idx = [next(i for i, p in enumerate(poses) if np.array_equal(p, g)) for g in got]
    assert idx[0] == 0 and idx == sorted(idx)
    if sym_deg is not None and 90 in sym_deg and dist > 1:
      assert 9 not in idx and 7 not in idx
    margin = 1e-3                                                  # degrees: float32 acos near the threshold
    def near(i, j):
      if np.linalg.norm(poses[i][:3, 3] - poses[j][:3, 3]) >= dist:
        return None
      return min(_geodesic_deg((poses[i].astype(np.float64) @ t)[:3, :3], poses[j][:3, :3].astype(np.float64)) for t in sym)
    for n, i in enumerate(idx):
      for j in idx[:n]:
        d = near(i, j)
        assert d is None or d >= angle - margin
    for i in set(range(len(poses))) - set(idx):
      ds = [near(i, j) for j in idx if j < i]
      assert any(d is not None and d < angle + margin for d in ds)


def test_rotation_grid_shrinks_under_symmetry():
  """src/estimater.py:106-124 with symmetry_tfs: the 252-pose grid thins out when the object repeats every 90 degrees about
  z; same poses from the package's native path and the oracle."""
  from oracle import geometry as G
  sym = np.stack([_rot_z(d) for d in (0, 90, 180, 270)])
  ref = G.make_rotation_grid(symmetry_tfs=sym)
  full = G.make_rotation_grid()
  assert len(full) == 252 and 40 <= len(ref) < 252
  views = U.sample_views_icosphere(n_views=40)
  grid = [np.linalg.inv(v @ U.euler_matrix(0, 0, a)) for v in views for a in np.deg2rad(np.arange(0, 360, 60))]
  got = np.asarray(U.cluster_poses(30, 99999, np.asarray(grid), sym.astype(np.float32)))
  assert got.shape == ref.shape
  np.testing.assert_allclose(got, ref, atol=1e-6)

"""Oracle: geometry, camera model, pose algebra, depth pre-processing (CPU, numpy / torch-CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Citations are reference file:line.
"""
import math
import numpy as np
import torch

# src/Utils.py:68-71
glcam_in_cvcam = np.array([[1, 0, 0, 0],
                           [0, -1, 0, 0],
                           [0, 0, -1, 0],
                           [0, 0, 0, 1]], dtype=np.float64)


def projection_matrix_from_intrinsics(K, height, width, znear, zfar, window_coords='y_down'):
  """src/Utils.py:752-802 (x0=y0=0)."""
  K = np.asarray(K, dtype=np.float64)
  w, h = float(width), float(height)
  depth = float(zfar - znear)
  q = -(zfar + znear) / depth
  qn = -2.0 * (zfar * znear) / depth
  if window_coords == 'y_up':
    row1 = [0, -2 * K[1, 1] / h, (-2 * K[1, 2] + h) / h, 0]
  elif window_coords == 'y_down':
    row1 = [0, 2 * K[1, 1] / h, (2 * K[1, 2] - h) / h, 0]
  else:
    raise NotImplementedError
  return np.array([[2 * K[0, 0] / w, -2 * K[0, 1] / w, (-2 * K[0, 2] + w) / w, 0],
                   row1,
                   [0, 0, q, qn],
                   [0, 0, -1, 0]], dtype=np.float64)


def to_homo_torch(pts):
  """src/Utils.py:520-526."""
  ones = torch.ones((*pts.shape[:-1], 1), dtype=torch.float, device=pts.device)
  return torch.cat((pts, ones), dim=-1)


def transform_pts(pts, tf):
  """src/Utils.py:529-536 (torch or numpy)."""
  if len(tf.shape) >= 3 and tf.shape[-3] != pts.shape[-2]:
    tf = tf[..., None, :, :]
  return (tf[..., :-1, :-1] @ pts[..., None] + tf[..., :-1, -1:])[..., 0]


def transform_dirs(dirs, tf):
  """src/Utils.py:539-546."""
  if len(tf.shape) >= 3 and tf.shape[-3] != dirs.shape[-2]:
    tf = tf[..., None, :, :]
  return (tf[..., :3, :3] @ dirs[..., None])[..., 0]


def depth2xyzmap(depth, K):
  """src/Utils.py:399-417 (uvs=None branch). numpy in -> (H,W,3) float32."""
  depth = np.asarray(depth)
  K = np.asarray(K)
  invalid_mask = depth < 0.001
  H, W = depth.shape[:2]
  vs, us = np.meshgrid(np.arange(0, H), np.arange(0, W), sparse=False, indexing='ij')
  vs = vs.reshape(-1)
  us = us.reshape(-1)
  zs = depth[vs, us]
  xs = (us - K[0, 2]) * zs / K[0, 0]
  ys = (vs - K[1, 2]) * zs / K[1, 1]
  pts = np.stack((xs.reshape(-1), ys.reshape(-1), zs.reshape(-1)), 1)
  xyz_map = np.zeros((H, W, 3), dtype=np.float32)
  xyz_map[vs, us] = pts
  xyz_map[invalid_mask] = 0
  return xyz_map


def depth2xyzmap_batch(depths, Ks, zfar):
  """src/Utils.py:420-438.  depths (B,H,W) float32 tensor, Ks (B,3,3) float32 -> (B,H,W,3)."""
  bs = depths.shape[0]
  invalid_mask = (depths < 0.001) | (depths > zfar)
  H, W = depths.shape[-2:]
  vs, us = torch.meshgrid(torch.arange(0, H), torch.arange(0, W), indexing='ij')
  vs = vs.reshape(-1).float()[None].expand(bs, -1)
  us = us.reshape(-1).float()[None].expand(bs, -1)
  zs = depths.reshape(bs, -1)
  Ks = Ks[:, None].expand(bs, zs.shape[-1], 3, 3)
  xs = (us - Ks[..., 0, 2]) * zs / Ks[..., 0, 0]
  ys = (vs - Ks[..., 1, 2]) * zs / Ks[..., 1, 1]
  pts = torch.stack([xs, ys, zs], dim=-1)
  xyz_maps = pts.reshape(bs, H, W, 3).clone()
  xyz_maps[invalid_mask] = 0
  return xyz_maps


def compute_crop_window_tf_batch(poses, K, crop_ratio, out_size, mesh_diameter):
  """src/Utils.py:577-621, method='box_3d'.  All arithmetic in float32, evaluated
  left-to-right without fused multiply-add (this fixes the order the HIP kernel
  mirrors; the reference does the same sums inside a cuBLAS 3x3 matmul).

  poses: (B,4,4) float32 tensor.  Returns tf_to_crops (B,3,3) float32 tensor."""
  poses = torch.as_tensor(poses, dtype=torch.float32)
  B = len(poses)
  Kf = torch.as_tensor(np.asarray(K, dtype=np.float64), dtype=torch.float32)
  radius = torch.tensor(float(mesh_diameter) * float(crop_ratio) / 2, dtype=torch.float32)
  zero = torch.zeros((), dtype=torch.float32)
  offsets = torch.stack([zero, zero, zero,
                         radius, zero, zero,
                         -radius, zero, zero,
                         zero, radius, zero,
                         zero, -radius, zero]).reshape(-1, 3)
  pts = poses[:, :3, 3].reshape(-1, 1, 3) + offsets.reshape(1, -1, 3)   # (B,5,3)
  x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]

  def row(i):
    return (Kf[i, 0] * x + Kf[i, 1] * y) + Kf[i, 2] * z
  pu, pv, pw = row(0), row(1), row(2)
  u = pu / pw
  v = pv / pw
  uvs = torch.stack([u, v], dim=-1)            # (B,5,2)
  center = uvs[:, 0]
  radius_px = torch.abs(uvs - center.reshape(-1, 1, 2)).reshape(B, -1).max(dim=-1)[0].reshape(-1)
  left = (center[:, 0] - radius_px).round()
  right = (center[:, 0] + radius_px).round()
  top = (center[:, 1] - radius_px).round()
  bottom = (center[:, 1] + radius_px).round()
  tf = torch.zeros((B, 3, 3), dtype=torch.float32)
  sx = out_size[0] / (right - left)
  sy = out_size[1] / (bottom - top)
  # new_tf @ tf with tf = [[1,0,-l],[0,1,-t],[0,0,1]], new_tf = diag(sx,sy,1)
  tf[:, 0, 0] = sx
  tf[:, 1, 1] = sy
  tf[:, 0, 2] = sx * (-left)
  tf[:, 1, 2] = sy * (-top)
  tf[:, 2, 2] = 1
  return tf


def crop_bbox2d_ori(tf_to_crops, input_resize):
  """learning/training/predict_pose_refine.py:44-45 / predict_score.py:71-72:
  bbox2d_crop = [(0,0),(W-1,H-1)] mapped through tf_to_crops.inverse() -> (B,4) umin,vmin,umax,vmax."""
  bbox2d_crop = torch.tensor([0, 0, input_resize[0] - 1, input_resize[1] - 1], dtype=torch.float32).reshape(2, 2)
  inv = torch.linalg.inv(tf_to_crops)
  return transform_pts(bbox2d_crop, inv).reshape(-1, 4)


def so3_exp_map(log_rot, eps=1e-4):
  """pytorch3d.transforms.so3_exp_map (branch `stable`; used at predict_pose_refine.py:222).
  PARITY UNPINNED (pytorch3d not in the reference checkout)."""
  log_rot = torch.as_tensor(log_rot, dtype=torch.float32)
  nrms = (log_rot * log_rot).sum(1)
  rot_angles = torch.clamp(nrms, eps).sqrt()
  rot_angles_inv = 1.0 / rot_angles
  fac1 = rot_angles_inv * rot_angles.sin()
  fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
  x, y, z = log_rot[:, 0], log_rot[:, 1], log_rot[:, 2]
  zeros = torch.zeros_like(x)
  skews = torch.stack([zeros, -z, y, z, zeros, -x, -y, x, zeros], dim=1).reshape(-1, 3, 3)
  skews_square = torch.bmm(skews, skews)
  return fac1[:, None, None] * skews + fac2[:, None, None] * skews_square + torch.eye(3, dtype=torch.float32)[None]


def rotation_6d_to_matrix(d6):
  """pytorch3d.transforms.rotation_6d_to_matrix (non-default branch, predict_pose_refine.py:224)."""
  a1, a2 = d6[..., :3], d6[..., 3:]
  b1 = torch.nn.functional.normalize(a1, dim=-1)
  b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
  b2 = torch.nn.functional.normalize(b2, dim=-1)
  b3 = torch.cross(b1, b2, dim=-1)
  return torch.stack((b1, b2, b3), dim=-2)


def egocentric_delta_pose_to_pose(A_in_cam, trans_delta, rot_mat_delta):
  """src/Utils.py:848-855."""
  B_in_cam = torch.eye(4, dtype=torch.float)[None].expand(len(A_in_cam), -1, -1).contiguous()
  B_in_cam[:, :3, 3] = A_in_cam[:, :3, 3] + trans_delta
  B_in_cam[:, :3, :3] = rot_mat_delta @ A_in_cam[:, :3, :3]
  return B_in_cam


def pose_to_egocentric_delta_pose(A_in_cam, B_in_cam):
  """src/Utils.py:836-843."""
  trans_delta = B_in_cam[:, :3, 3] - A_in_cam[:, :3, 3]
  rot_mat_delta = B_in_cam[:, :3, :3] @ A_in_cam[:, :3, :3].permute(0, 2, 1)
  return trans_delta, rot_mat_delta


def guess_translation(depth, mask, K):
  """src/estimater.py:137-156 (debug branch omitted)."""
  vs, us = np.where(mask > 0)
  if len(us) == 0:
    return np.zeros((3))
  uc = (us.min() + us.max()) / 2.0
  vc = (vs.min() + vs.max()) / 2.0
  valid = mask.astype(bool) & (depth >= 0.001)
  if not valid.any():
    return np.zeros((3))
  zc = np.median(depth[valid])
  center = (np.linalg.inv(K) @ np.asarray([uc, vc, 1]).reshape(3, 1)) * zc
  return center.reshape(3)


def compute_mesh_diameter(model_pts, n_sample=10000, rng=None):
  """src/Utils.py:559-574 (model_pts branch).  `rng` replaces the global np.random state."""
  model_pts = np.asarray(model_pts)
  if n_sample is None or n_sample >= len(model_pts):
    pts = model_pts if n_sample is None else model_pts[(rng or np.random).choice(len(model_pts), size=len(model_pts), replace=False)]
  else:
    ids = (rng or np.random).choice(len(model_pts), size=n_sample, replace=False)
    pts = model_pts[ids]
  best = 0.0
  for s in range(0, len(pts), 1024):    # blocked so the transient stays small
    d = np.linalg.norm(pts[None] - pts[s:s + 1024, None], axis=-1)
    best = max(best, float(d.max()))
  return best


# ----------------------------------------------------------------------------------------------
# depth pre-processing: NVIDIA Warp kernels restated (src/Utils.py:303-395), float32 arithmetic
# ----------------------------------------------------------------------------------------------
def _shifted(depth, du, dv, fill):
  """value of depth[v+dv, u+du] with `fill` outside the image, plus an in-bounds mask."""
  H, W = depth.shape
  out = np.full((H, W), fill, dtype=depth.dtype)
  inb = np.zeros((H, W), dtype=bool)
  v0, v1 = max(0, -dv), min(H, H - dv)
  u0, u1 = max(0, -du), min(W, W - du)
  if v1 > v0 and u1 > u0:
    out[v0:v1, u0:u1] = depth[v0 + dv:v1 + dv, u0 + du:u1 + du]
    inb[v0:v1, u0:u1] = True
  return out, inb


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100):
  """src/Utils.py:359-395.  Note :367-368 writes 0 for an invalid centre but does not return;
  the final ratio test decides (an invalid centre makes every |cur-d_ori| comparison use d_ori)."""
  depth = np.asarray(depth, dtype=np.float32)
  f32 = np.float32
  bad = np.zeros(depth.shape, dtype=np.float32)
  total = np.zeros(depth.shape, dtype=np.float32)
  for du in range(-radius, radius + 1):
    for dv in range(-radius, radius + 1):
      cur, inb = _shifted(depth, du, dv, f32(0))
      total += inb.astype(np.float32)
      isbad = (cur < f32(0.001)) | (cur >= f32(zfar)) | (np.abs(cur - depth) > f32(depth_diff_thres))
      bad += (isbad & inb).astype(np.float32)
  out = np.where(bad / total > f32(ratio_thres), f32(0), depth).astype(np.float32)
  return out


def bilateral_filter_depth(depth, radius=2, zfar=100, sigmaD=2, sigmaR=100000):
  """src/Utils.py:304-356.  Accumulation order = u outer, v inner, exactly as the Warp kernel."""
  depth = np.asarray(depth, dtype=np.float32)
  f32 = np.float32
  mean = np.zeros(depth.shape, dtype=np.float32)
  num_valid = np.zeros(depth.shape, dtype=np.int32)
  for du in range(-radius, radius + 1):
    for dv in range(-radius, radius + 1):
      cur, inb = _shifted(depth, du, dv, f32(0))
      valid = inb & (cur >= f32(0.001)) & (cur < f32(zfar))
      num_valid += valid
      mean = mean + np.where(valid, cur, f32(0))
  with np.errstate(divide='ignore', invalid='ignore'):
    mean = mean / num_valid.astype(np.float32)
  sum_w = np.zeros(depth.shape, dtype=np.float32)
  acc = np.zeros(depth.shape, dtype=np.float32)
  for du in range(-radius, radius + 1):
    for dv in range(-radius, radius + 1):
      cur, inb = _shifted(depth, du, dv, f32(0))
      with np.errstate(invalid='ignore'):
        ok = inb & (cur >= f32(0.001)) & (cur < f32(zfar)) & (np.abs(cur - mean) < f32(0.01))
      spatial = f32(-float(du * du + dv * dv)) / (f32(2.0) * f32(sigmaD) * f32(sigmaD))
      diff = depth - cur
      rng = (diff * diff) / (f32(2.0) * f32(sigmaR) * f32(sigmaR))
      w = np.exp((spatial - rng).astype(np.float32)).astype(np.float32)
      w = np.where(ok, w, f32(0))
      sum_w = sum_w + w
      acc = acc + w * cur
  with np.errstate(divide='ignore', invalid='ignore'):
    out = np.where((sum_w > 0) & (num_valid > 0), acc / sum_w, f32(0)).astype(np.float32)
  return out


# ----------------------------------------------------------------------------------------------
# hypothesis set: icosphere viewpoints x in-plane rotations, symmetry clustering
# ----------------------------------------------------------------------------------------------
def icosphere_vertices(subdivisions=1):
  """Unit icosphere vertices: 12 icosahedron vertices followed by the midpoints of its 30 edges
  (one subdivision), normalised.  trimesh.creation.icosphere (src/Utils.py:485-489) is absent
  here; its vertex ORDER is PARITY UNPINNED, so the rotation grid is an input fixture shared by
  the oracle and the HIP path."""
  t = (1.0 + 5.0 ** 0.5) / 2.0
  v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
  f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
  v /= np.linalg.norm(v, axis=1, keepdims=True)
  for _ in range(subdivisions):
    edges = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0), axis=1)
    uniq, inv = np.unique(edges, axis=0, return_inverse=True)
    mid = v[uniq].mean(axis=1)
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    inv = inv.reshape(3, -1).T + len(v)
    f = np.concatenate([np.stack([f[:, 0], inv[:, 0], inv[:, 2]], 1),
                        np.stack([f[:, 1], inv[:, 1], inv[:, 0]], 1),
                        np.stack([f[:, 2], inv[:, 2], inv[:, 1]], 1),
                        np.stack([inv[:, 0], inv[:, 1], inv[:, 2]], 1)], 0)
    v = np.concatenate([v, mid], 0)
  return v


def sample_views_icosphere(n_views, subdivisions=None, radius=1):
  """src/Utils.py:483-507."""
  if subdivisions is not None:
    verts = icosphere_vertices(subdivisions) * radius
  else:
    subdivision = 1
    while 1:
      verts = icosphere_vertices(subdivision) * radius
      if verts.shape[0] >= n_views:
        break
      subdivision += 1
  cam_in_obs = np.tile(np.eye(4)[None], (len(verts), 1, 1))
  cam_in_obs[:, :3, 3] = verts
  up = np.array([0, 0, 1])
  z_axis = -cam_in_obs[:, :3, 3]
  z_axis /= np.linalg.norm(z_axis, axis=-1).reshape(-1, 1)
  x_axis = np.cross(up.reshape(1, 3), z_axis)
  invalid = (x_axis == 0).all(axis=-1)
  x_axis[invalid] = [1, 0, 0]
  x_axis /= np.linalg.norm(x_axis, axis=-1).reshape(-1, 1)
  y_axis = np.cross(z_axis, x_axis)
  y_axis /= np.linalg.norm(y_axis, axis=-1).reshape(-1, 1)
  cam_in_obs[:, :3, 0] = x_axis
  cam_in_obs[:, :3, 1] = y_axis
  cam_in_obs[:, :3, 2] = z_axis
  return cam_in_obs


def euler_matrix_z(a):
  """transformations.euler_matrix(0,0,a) (axes 'sxyz') = rotation about z (src/estimater.py:113)."""
  c, s = math.cos(a), math.sin(a)
  return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)


def rotation_geodesic_distance(R1, R2):
  """mycpp/src/Utils.cpp:21-26 (float32)."""
  c = (np.float32(np.trace(R1.astype(np.float32) @ R2.astype(np.float32).T)) - np.float32(1)) / np.float32(2.0)
  c = max(min(float(c), 1.0), -1.0)
  return math.acos(c)


def cluster_poses(angle_diff_deg, dist_diff, poses_in, symmetry_tfs):
  """mycpp/src/app/pybind_api.cpp:24-68: greedy keep-if-new, float32 Eigen arithmetic."""
  poses_in = [np.asarray(p, dtype=np.float32) for p in poses_in]
  symmetry_tfs = [np.asarray(t, dtype=np.float32) for t in symmetry_tfs]
  poses_out = [poses_in[0]]
  radian_thres = np.float32(angle_diff_deg) / np.float32(180.0) * np.float32(math.pi)
  for i in range(1, len(poses_in)):
    isnew = True
    cur = poses_in[i]
    for cluster in poses_out:
      if np.linalg.norm(cluster[:3, 3] - cur[:3, 3]) >= dist_diff:
        continue
      for tf in symmetry_tfs:
        tmp = cur @ tf
        if rotation_geodesic_distance(tmp[:3, :3], cluster[:3, :3]) < radian_thres:
          isnew = False
          break
      if not isnew:
        break
    if isnew:
      poses_out.append(cur)
  return poses_out


def make_rotation_grid(min_n_views=40, inplane_step=60, symmetry_tfs=None):
  """src/estimater.py:106-124."""
  cam_in_obs = sample_views_icosphere(n_views=min_n_views)
  rot_grid = []
  for i in range(len(cam_in_obs)):
    for inplane_rot in np.deg2rad(np.arange(0, 360, inplane_step)):
      cam_in_ob = cam_in_obs[i] @ euler_matrix_z(inplane_rot)
      rot_grid.append(np.linalg.inv(cam_in_ob))
  rot_grid = np.asarray(rot_grid)
  if symmetry_tfs is None:
    symmetry_tfs = np.eye(4)[None]
  rot_grid = cluster_poses(30, 99999, rot_grid, symmetry_tfs)
  return np.asarray(rot_grid, dtype=np.float32)

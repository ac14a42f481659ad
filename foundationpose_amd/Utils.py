"""Host-side mirror of the hot-path subset of the reference's `src/Utils.py`, backed by the HIP library.

Same names, argument meaning and error behaviour as the reference functions cited in each
docstring; tensors live on the HIP device ('cuda' in torch-ROCm).  The rasteriser context is
`RasterizeContext` (the reference's `dr.RasterizeCudaContext`).
"""
import logging
import math
import random

import ctypes
import numpy as np
import torch

from . import _lib
from ._lib import check, k_ptr, lib, ptr, stream_ptr
from .mesh_tensors import make_mesh_tensors  # noqa: F401  (src/Utils.py:104-130)
from .vis import cv_draw_text, depth_to_vis, make_grid_image  # noqa: F401  (src/Utils.py:630-653, 456-478, 293-300)

glcam_in_cvcam = np.array([[1, 0, 0, 0],
                           [0, -1, 0, 0],
                           [0, 0, -1, 0],
                           [0, 0, 0, 1]]).astype(float)


def set_logging_format(level=logging.INFO):
  """src/Utils.py:94-99"""
  logging.basicConfig(level=level, format='[%(funcName)s()] %(message)s')


def set_seed(random_seed):
  """src/Utils.py:222-229"""
  np.random.seed(random_seed)
  random.seed(random_seed)
  torch.manual_seed(random_seed)
  if torch.cuda.is_available():
    torch.cuda.manual_seed_all(random_seed)


class RasterizeContext:
  """Opaque, reusable per-device handle; replaces dr.RasterizeCudaContext(device)
  (main.py:42; src/estimater.py:102,168; src/Utils.py:147)."""

  def __init__(self, device='cuda'):
    self.device = torch.device(device)
    self.ctx = _lib.Context.get(self.device)


RasterizeCudaContext = RasterizeContext   # the reference's spelling


def _ctx_of(glctx, device=None):
  if glctx is None:
    return _lib.Context.get(device)
  if isinstance(glctx, RasterizeContext):
    return glctx.ctx
  if isinstance(glctx, _lib.Context):
    return glctx
  raise TypeError(f'glctx must be a foundationpose_amd RasterizeContext, got {type(glctx)}')


def nvdiffrast_render(K=None, H=None, W=None, ob_in_cams=None, glctx=None, context='cuda', get_normal=False, mesh_tensors=None,
                      mesh=None, projection_mat=None, bbox2d=None, output_size=None, use_light=False, light_color=None,
                      light_dir=np.array([0, 0, 1]), light_pos=np.array([0, 0, 0]), w_ambient=0.8, w_diffuse=0.5, extra={}):
  """src/Utils.py:133-219.  Returns (color (N,h,w,3), depth (N,h,w), normal_map (N,h,w,3)|None);
  extra['xyz_map'] (N,h,w,3).  Rendered by the hand-written HIP rasteriser (csrc/raster.hip)."""
  if glctx is None:
    if context == 'gl' or context == 'cuda':
      glctx = RasterizeContext()
      logging.info("created context")
    else:
      raise NotImplementedError
  if mesh_tensors is None:
    mesh_tensors = make_mesh_tensors(mesh)
  ctx = _ctx_of(glctx)
  dev = torch.device('cuda', ctx.device_index)
  poses = torch.as_tensor(ob_in_cams, dtype=torch.float, device=dev).reshape(-1, 4, 4).contiguous()
  N = len(poses)
  if output_size is None:
    output_size = np.asarray([H, W])
  h, w = int(output_size[0]), int(output_size[1])
  bb = None
  if bbox2d is not None:
    bb = torch.as_tensor(bbox2d, dtype=torch.float, device=dev).reshape(-1, 4).contiguous()
    assert len(bb) == N
  if use_light:
    get_normal = True
  dm = _lib.device_mesh(ctx, mesh_tensors)
  color = torch.empty((N, h, w, 3), dtype=torch.float, device=dev)
  depth = torch.empty((N, h, w), dtype=torch.float, device=dev)
  xyz = torch.empty((N, h, w, 3), dtype=torch.float, device=dev)
  normal = torch.empty((N, h, w, 3), dtype=torch.float, device=dev) if get_normal else None
  default_light = light_color is None and light_dir is not None and np.array_equal(np.asarray(light_dir, dtype=float).reshape(-1), [0, 0, 1])
  # extra={'rast': ...} asks for dr.rasterize's own output as well (u, v, z/w, triangle id + 1 per pixel: src/Utils.py:182's rast_out, rows
  # flipped like the images) - the reference keeps it internal; the parity tests compare coverage and the winning face on it
  rast = torch.empty((N, h, w, 4), dtype=torch.float, device=dev) if 'rast' in extra else None
  if projection_mat is None and (default_light or not use_light) and rast is None:
    Kd, Kp = k_ptr(K)
    check(lib().fp_render(ctx.handle, dm.handle, ptr(poses), N, Kp, int(H), int(W), ptr(bb), h, w, 1 if use_light else 0,
                          float(w_ambient), float(w_diffuse), ptr(color), ptr(depth), ptr(normal), ptr(xyz), stream_ptr(dev)))
  else:
    # light_dir / light_pos / light_color / projection_mat (src/Utils.py:159-161,200-211) travel in fp_render_opts
    o = _lib.FpRenderOpts()
    o.struct_size = ctypes.sizeof(o)
    o.use_light, o.w_ambient, o.w_diffuse = (1 if use_light else 0), float(w_ambient), float(w_diffuse)
    if default_light:
      o.light_mode = 0
    elif light_dir is not None:
      o.light_mode = 1
      o.light_vec[:] = [-float(x) for x in np.asarray(light_dir, dtype=np.float32).reshape(3)]      # light_dir_neg
    else:
      o.light_mode = 2
      o.light_vec[:] = [float(x) for x in np.asarray(light_pos, dtype=np.float32).reshape(3)]
    if light_color is not None:
      o.has_light_color = 1
      o.light_color[:] = [float(x) for x in np.asarray(torch.as_tensor(light_color).cpu(), dtype=np.float32).reshape(3)]
    if projection_mat is not None:
      pm = np.asarray(torch.as_tensor(projection_mat).cpu(), dtype=np.float64).reshape(-1, 4, 4)
      if len(pm) != 1:
        raise NotImplementedError('one projection_mat per call (the reference broadcasts a single matrix in every in-repo call)')
      o.has_projection = 1
      o.projection[:] = [float(x) for x in pm[0].reshape(16)]
    Kp = None
    if K is not None:
      Kd, Kp = k_ptr(K)
    o.d_rast = ptr(rast)
    check(lib().fp_render_ex(ctx.handle, dm.handle, ptr(poses), N, Kp, int(H), int(W), ptr(bb), h, w, ctypes.byref(o), ptr(color), ptr(depth),
                             ptr(normal), ptr(xyz), stream_ptr(dev)))
  extra['xyz_map'] = xyz
  if rast is not None:
    extra['rast'] = rast
  return color, depth, normal


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100, device='cuda'):
  """src/Utils.py:387-395 (numpy in -> numpy out, tensor in -> tensor out)."""
  d = torch.as_tensor(depth, dtype=torch.float, device=device).contiguous()
  ctx = _lib.Context.get(d.device)
  out = torch.empty_like(d)
  check(lib().fp_erode_depth(ctx.handle, ptr(d), d.shape[0], d.shape[1], int(radius), float(depth_diff_thres), float(ratio_thres),
                             float(zfar), ptr(out), stream_ptr(d.device)))
  if isinstance(depth, np.ndarray):
    out = out.data.cpu().numpy()
  return out


def bilateral_filter_depth(depth, radius=2, zfar=100, sigmaD=2, sigmaR=100000, device='cuda'):
  """src/Utils.py:345-356"""
  d = torch.as_tensor(depth, dtype=torch.float, device=device).contiguous()
  ctx = _lib.Context.get(d.device)
  out = torch.empty_like(d)
  check(lib().fp_bilateral_filter_depth(ctx.handle, ptr(d), d.shape[0], d.shape[1], int(radius), float(zfar), float(sigmaD),
                                        float(sigmaR), ptr(out), stream_ptr(d.device)))
  if isinstance(depth, np.ndarray):
    out = out.data.cpu().numpy()
  return out


def depth_prefilter(depth, K, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100, sigmaD=2, sigmaR=100000, zfar_xyz=np.inf, rgb_u8=None):
  """The depth prelude of a tracking frame (src/estimater.py:256-260) in one launch (a build extension; not in the reference):
  bilateral_filter_depth(erode_depth(depth, radius), radius) and depth2xyzmap_batch of the result with the float32 camera matrix.
  Returns (depth (H,W), xyz_map (H,W,3)) on the device, bit-identical to the three calls chained; with `rgb_u8` (H,W,3) uint8 on the
  device also its float copy (rgb_u8.to(torch.float)) as a third value, from the same launch."""
  d = torch.as_tensor(depth, dtype=torch.float, device='cuda').contiguous()
  ctx = _lib.Context.get(d.device)
  H, W = d.shape
  out = torch.empty_like(d)
  xyz = torch.empty((H, W, 3), dtype=torch.float, device=d.device)
  Kd, Kp = k_ptr(np.asarray(K.detach().cpu().numpy() if torch.is_tensor(K) else K, dtype=np.float32))
  zf = float(zfar_xyz) if np.isfinite(zfar_xyz) else 3.0e38
  rgb_f = None
  if rgb_u8 is not None:
    if not (torch.is_tensor(rgb_u8) and rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and tuple(rgb_u8.shape) == (H, W, 3) and rgb_u8.is_contiguous()):
      raise ValueError('depth_prefilter: rgb_u8 must be a contiguous (H,W,3) uint8 tensor on the device')
    rgb_f = torch.empty((H, W, 3), dtype=torch.float, device=d.device)
  check(lib().fp_depth_prefilter(ctx.handle, ptr(d), H, W, int(radius), float(depth_diff_thres), float(ratio_thres), float(zfar), float(zfar),
                                 float(sigmaD), float(sigmaR), Kp, zf, ptr(out), ptr(xyz), ptr(rgb_u8), ptr(rgb_f), stream_ptr(d.device)))
  return (out, xyz) if rgb_u8 is None else (out, xyz, rgb_f)


def depth2xyzmap(depth, K, uvs=None):
  """src/Utils.py:399-417: back-projection in float64 arithmetic, one rounding to float32, depth < 1 mm -> 0.
  Both input kinds run fp_depth2xyzmap_f64 on the device; a numpy image comes back as numpy (H,W,3) float32 like the
  reference's, a device tensor stays on the device.  `uvs` (n,2) keeps only the listed (rounded) pixels, zeros elsewhere."""
  as_numpy = not torch.is_tensor(depth)
  d = torch.as_tensor(np.ascontiguousarray(depth) if as_numpy else depth, device='cuda').to(torch.float).contiguous()
  ctx = _lib.Context.get(d.device)
  H, W = d.shape[:2]
  xyz = torch.empty((H, W, 3), dtype=torch.float, device=d.device)
  Kd, Kp = k_ptr(K)
  check(lib().fp_depth2xyzmap_f64(ctx.handle, ptr(d), H, W, Kp, ptr(xyz), stream_ptr(d.device)))
  if uvs is not None:
    px = torch.as_tensor(np.asarray(uvs).round().astype(np.int64), device=d.device)
    keep = torch.zeros((H, W, 1), dtype=torch.bool, device=d.device)
    keep[px[:, 1], px[:, 0]] = True
    xyz = xyz * keep
  return xyz.cpu().numpy() if as_numpy else xyz


def mask_depth_stats(depth, mask, min_depth=0.001):
  """Reductions of guess_translation / register()'s validity test on the device (src/estimater.py:137-156,173-177):
  dict(cmin, cmax, rmin, rmax, n_mask, n_usable, median) for a device depth image and a mask (anything non-zero = object)."""
  d = torch.as_tensor(depth, dtype=torch.float, device='cuda').contiguous()
  m = torch.as_tensor(np.ascontiguousarray(mask) if isinstance(mask, np.ndarray) else mask, device=d.device)
  m = (m != 0).to(torch.uint8).contiguous()
  assert m.shape == d.shape, 'mask and depth shapes differ'
  ctx = _lib.Context.get(d.device)
  st = (ctypes.c_int32 * 6)()
  med = ctypes.c_float()
  check(lib().fp_mask_depth_stats(ctx.handle, ptr(d), ptr(m), d.shape[0], d.shape[1], float(min_depth), st, ctypes.byref(med),
                                  stream_ptr(d.device)))
  return dict(cmin=st[0], cmax=st[1], rmin=st[2], rmax=st[3], n_mask=st[4], n_usable=st[5], median=np.float32(med.value))


def depth2xyzmap_batch(depths, Ks, zfar):
  """src/Utils.py:420-438: (B,H,W) device tensor + (B,3,3) -> (B,H,W,3), float32 on the device."""
  depths = torch.as_tensor(depths, dtype=torch.float, device='cuda').contiguous()
  ctx = _lib.Context.get(depths.device)
  B, H, W = depths.shape
  out = torch.empty((B, H, W, 3), dtype=torch.float, device=depths.device)
  Ks = torch.as_tensor(Ks).reshape(-1, 3, 3)
  zf = float(zfar) if np.isfinite(zfar) else 3.0e38
  for b in range(B):
    Kd, Kp = k_ptr(Ks[b if len(Ks) > 1 else 0])
    check(lib().fp_depth2xyzmap(ctx.handle, ptr(depths[b]), H, W, Kp, zf, ptr(out[b]), stream_ptr(depths.device)))
  return out


def compute_crop_window_tf_batch(pts=None, H=None, W=None, poses=None, K=None, crop_ratio=1.2, out_size=None, rgb=None, uvs=None,
                                 method='min_box', mesh_diameter=None):
  """src/Utils.py:577-621.  Only method='box_3d' exists in the reference's hot path; anything else
  raises RuntimeError exactly as the reference does."""
  if method != 'box_3d':
    raise RuntimeError
  poses = torch.as_tensor(poses, dtype=torch.float, device='cuda').reshape(-1, 4, 4).contiguous()
  ctx = _lib.Context.get(poses.device)
  tf = torch.empty((len(poses), 3, 3), dtype=torch.float, device=poses.device)
  Kd, Kp = k_ptr(K)
  check(lib().fp_crop_window_tf(ctx.handle, ptr(poses), len(poses), Kp, float(crop_ratio), float(mesh_diameter), int(out_size[0]),
                                int(out_size[1]), ptr(tf), None, stream_ptr(poses.device)))
  return tf


def projection_matrix_from_intrinsics(K, height, width, znear, zfar, window_coords='y_down'):
  """src/Utils.py:752-802"""
  depth = float(zfar - znear)
  q = -(zfar + znear) / depth
  qn = -2 * (zfar * znear) / depth
  w, h = width, height
  if window_coords == 'y_up':
    r1 = [0, -2 * K[1, 1] / h, (-2 * K[1, 2] + h) / h, 0]
  elif window_coords == 'y_down':
    r1 = [0, 2 * K[1, 1] / h, (2 * K[1, 2] - h) / h, 0]
  else:
    raise NotImplementedError
  return np.array([[2 * K[0, 0] / w, -2 * K[0, 1] / w, (-2 * K[0, 2] + w) / w, 0], r1, [0, 0, q, qn], [0, 0, -1, 0]])


def to_homo_torch(pts):
  """src/Utils.py:520-526: append w = 1."""
  return torch.nn.functional.pad(pts.to(torch.float), (0, 1), value=1.0)


def _apply_linear(vecs, mats, offset=None):
  """Row-vector form of the reference's broadcasting rule (src/Utils.py:529-546): a stack of B transforms whose B is not
  the number of vectors maps EVERY vector (result (B,N,3)); B equal to the vector count maps them one to one."""
  one_to_one = mats.ndim >= 3 and mats.shape[-3] == vecs.shape[-2]
  if one_to_one:
    out = (vecs[..., None, :] @ mats.swapaxes(-1, -2))[..., 0, :]
    return out if offset is None else out + offset
  out = vecs @ mats.swapaxes(-1, -2)
  return out if offset is None else out + offset[..., None, :]


def transform_pts(pts, tf):
  """src/Utils.py:529-536: R p + t."""
  return _apply_linear(pts, tf[..., :-1, :-1], tf[..., :-1, -1])


def transform_dirs(dirs, tf):
  """src/Utils.py:539-546: R d."""
  return _apply_linear(dirs, tf[..., :3, :3])


def pose_to_egocentric_delta_pose(A_in_cam, B_in_cam):
  """src/Utils.py:838-844: the inverse of egocentric_delta_pose_to_pose -> (trans_delta (B,3), rot_mat_delta (B,3,3))."""
  return B_in_cam[:, :3, 3] - A_in_cam[:, :3, 3], B_in_cam[:, :3, :3] @ A_in_cam[:, :3, :3].transpose(1, 2)


def egocentric_delta_pose_to_pose(A_in_cam, trans_delta, rot_mat_delta):
  """src/Utils.py:848-855: t' = t + dt, R' = dR R (the camera-frame update of the refiner)."""
  n = len(A_in_cam)
  top = torch.cat([rot_mat_delta @ A_in_cam[:, :3, :3], (A_in_cam[:, :3, 3] + trans_delta)[..., None]], dim=-1)
  bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float, device=A_in_cam.device).expand(n, 1, 4)
  return torch.cat([top.to(torch.float), bottom], dim=1)


def compute_mesh_diameter(model_pts=None, mesh=None, n_sample=1000):
  """src/Utils.py:559-574 (model_pts branch; the O(n^2) distance matrix is evaluated in blocks)."""
  if mesh is not None:
    import scipy.linalg
    u, s, vh = scipy.linalg.svd(mesh.vertices, full_matrices=False)
    pts = u @ s
    return float(np.linalg.norm(pts.max(axis=0) - pts.min(axis=0)))
  model_pts = np.asarray(model_pts)
  if n_sample is None:
    pts = model_pts
  else:
    ids = np.random.choice(len(model_pts), size=min(n_sample, len(model_pts)), replace=False)
    pts = model_pts[ids]
  best = 0.0
  for s0 in range(0, len(pts), 1024):
    best = max(best, float(np.linalg.norm(pts[None] - pts[s0:s0 + 1024, None], axis=-1).max()))
  return best


def _icosphere_vertices(subdivisions):
  """Unit icosphere: the 12 icosahedron vertices followed, per subdivision, by the normalised
  midpoints of the unique edges (sorted by vertex pair).  trimesh.creation.icosphere
  (src/Utils.py:485-489) is not available offline; its vertex order is unpinned."""
  t = (1.0 + 5.0 ** 0.5) / 2.0
  v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
  f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
  v /= np.linalg.norm(v, axis=1, keepdims=True)
  for _ in range(subdivisions):
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0), axis=1)
    uniq, inv = np.unique(e, axis=0, return_inverse=True)
    mid = v[uniq].mean(axis=1)
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    inv = inv.reshape(3, -1).T + len(v)
    f = np.concatenate([np.stack([f[:, 0], inv[:, 0], inv[:, 2]], 1), np.stack([f[:, 1], inv[:, 1], inv[:, 0]], 1),
                        np.stack([f[:, 2], inv[:, 2], inv[:, 1]], 1), np.stack([inv[:, 0], inv[:, 1], inv[:, 2]], 1)], 0)
    v = np.concatenate([v, mid], 0)
  return v


def sample_views_icosphere(n_views, subdivisions=None, radius=1):
  """src/Utils.py:483-507"""
  if subdivisions is not None:
    verts = _icosphere_vertices(subdivisions) * radius
  else:
    subdivision = 1
    while 1:
      verts = _icosphere_vertices(subdivision) * radius
      if verts.shape[0] >= n_views:
        break
      subdivision += 1
  return _look_at_origin(verts)


def _look_at_origin(eyes):
  """Camera frames at `eyes` looking at the origin (src/Utils.py:491-507): z towards the origin, x = up x z with up = +z
  of the object (x = +x of the object where that product vanishes, i.e. at the poles), y = z x x."""
  eyes = np.asarray(eyes, dtype=np.float64)
  fwd = -eyes / np.linalg.norm(eyes, axis=1, keepdims=True)
  right = np.cross(np.broadcast_to([0.0, 0.0, 1.0], fwd.shape), fwd)
  right[~right.any(axis=1)] = (1.0, 0.0, 0.0)
  right /= np.linalg.norm(right, axis=1, keepdims=True)
  down = np.cross(fwd, right)
  down /= np.linalg.norm(down, axis=1, keepdims=True)
  frames = np.zeros((len(eyes), 4, 4))
  frames[:, :3, :] = np.stack([right, down, fwd, eyes], axis=2)
  frames[:, 3, 3] = 1.0
  return frames


def euler_matrix(ai, aj, ak):
  """transformations.euler_matrix(..., axes='sxyz') restricted to what src/estimater.py:113 uses:
  R = Rz(ak) Ry(aj) Rx(ai)."""
  ci, si, cj, sj, ck, sk = math.cos(ai), math.sin(ai), math.cos(aj), math.sin(aj), math.cos(ak), math.sin(ak)
  M = np.eye(4)
  M[:3, :3] = np.array([[cj * ck, sj * si * ck - ci * sk, sj * ci * ck + si * sk],
                        [cj * sk, sj * si * sk + ci * ck, sj * ci * sk - si * ck],
                        [-sj, cj * si, cj * ci]])
  return M


def cluster_poses(angle_diff, dist_diff, poses_in, symmetry_tfs):
  """mycpp.cluster_poses (mycpp/src/app/pybind_api.cpp:24-68) - native host code in the HIP library."""
  pin = np.ascontiguousarray(np.asarray(poses_in, dtype=np.float32).reshape(-1, 4, 4))
  sym = np.ascontiguousarray(np.asarray(symmetry_tfs, dtype=np.float32).reshape(-1, 4, 4))
  logging.info(f'num original candidates = {len(pin)}')      # the reference's C++ prints these two lines to stdout
  out = np.zeros_like(pin)
  n = lib().fp_cluster_poses(float(angle_diff), float(dist_diff), ptr(pin), len(pin), ptr(sym), len(sym), ptr(out))
  if n < 0:
    check(n)
  logging.info(f'num of pose after clustering: {n}')
  return [out[i] for i in range(n)]

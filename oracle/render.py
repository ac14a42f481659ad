"""Oracle: `nvdiffrast_render` restated (src/Utils.py:133-219) on top of oracle/raster_c.c.

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import ctypes
import os
import subprocess
import numpy as np
import torch

from .geometry import projection_matrix_from_intrinsics, glcam_in_cvcam

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, '_build')
_LIB = None


class _Mesh(ctypes.Structure):
  _fields_ = [('pos', ctypes.c_void_p), ('faces', ctypes.c_void_p), ('vnormals', ctypes.c_void_p),
              ('vcolor', ctypes.c_void_p), ('uv', ctypes.c_void_p), ('uv_idx', ctypes.c_void_p),
              ('tex', ctypes.c_void_p), ('V', ctypes.c_int), ('F', ctypes.c_int),
              ('texH', ctypes.c_int), ('texW', ctypes.c_int)]


def build(force=False):
  """gcc-compile the C oracle into oracle/_build/ (git-ignored; travels with gpurun)."""
  os.makedirs(_BUILD, exist_ok=True)
  so = os.path.join(_BUILD, 'liboracle_raster.so')
  src = os.path.join(_HERE, 'raster_c.c')
  if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-fopenmp', '-ffp-contract=off', '-mfma',
                           '-o', so, src, '-lm'])
  return so


def _lib():
  global _LIB
  if _LIB is None:
    _LIB = ctypes.CDLL(build())
    _LIB.oracle_render.restype = ctypes.c_int
  return _LIB


def _np(x, dtype):
  if torch.is_tensor(x):
    x = x.detach().cpu().numpy()
  return np.ascontiguousarray(x, dtype=dtype)


def clip_matrices(K, H, W, ob_in_cams, bbox2d=None, projection_mat=None):
  """src/Utils.py:155-181: mtx = proj @ glcam_in_cvcam @ ob_in_cam, then the bbox window transform
  (`pos_clip @ tf`) folded into rows 0/1.  Evaluated in float64, rounded once to float32."""
  ob = _np(ob_in_cams, np.float64).reshape(-1, 4, 4)
  if projection_mat is None:
    projection_mat = projection_matrix_from_intrinsics(K, height=H, width=W, znear=0.001, zfar=100)
  P = np.asarray(projection_mat, dtype=np.float64).reshape(4, 4)
  mtx = P[None] @ (glcam_in_cvcam[None] @ ob)
  if bbox2d is not None:
    bb = _np(bbox2d, np.float64).reshape(-1, 4)
    l, t, r, b = bb[:, 0], H - bb[:, 1], bb[:, 2], H - bb[:, 3]
    t00, t11 = W / (r - l), H / (t - b)
    t30, t31 = (W - r - l) / (r - l), (H - t - b) / (t - b)
    row3 = mtx[:, 3].copy()
    mtx[:, 0] = t00[:, None] * mtx[:, 0] + t30[:, None] * row3
    mtx[:, 1] = t11[:, None] * mtx[:, 1] + t31[:, None] * row3
  return mtx.astype(np.float32)


def nvdiffrast_render(K=None, H=None, W=None, ob_in_cams=None, get_normal=False, mesh_tensors=None,
                      projection_mat=None, bbox2d=None, output_size=None, use_light=False, light_color=None,
                      light_dir=(0, 0, 1), light_pos=(0, 0, 0), w_ambient=0.8, w_diffuse=0.5, extra=None):
  """Returns (color (B,h,w,3), depth (B,h,w), normal_map (B,h,w,3) or None) float32 torch-CPU
  tensors; extra['xyz_map'] (B,h,w,3), extra['rast'] (B,h,w,4) - same contract as the reference
  (src/Utils.py:133-219 incl. the light_dir / light_pos / light_color branches of :200-211)."""
  if extra is None:
    extra = {}
  if output_size is None:
    output_size = (H, W)
  Ho, Wo = int(output_size[0]), int(output_size[1])
  pose = _np(ob_in_cams, np.float32).reshape(-1, 4, 4)
  B = len(pose)
  M = clip_matrices(K, H, W, pose, bbox2d, projection_mat)
  pos = _np(mesh_tensors['pos'], np.float32)
  faces = _np(mesh_tensors['faces'], np.int32)
  vn = _np(mesh_tensors['vnormals'], np.float32)
  keep = [pos, faces, vn]
  m = _Mesh()
  m.pos, m.faces, m.vnormals = pos.ctypes.data, faces.ctypes.data, vn.ctypes.data
  m.V, m.F = len(pos), len(faces)
  if 'tex' in mesh_tensors:
    tex = _np(mesh_tensors['tex'], np.float32).reshape(-1, mesh_tensors['tex'].shape[-2], 3)
    uv = _np(mesh_tensors['uv'], np.float32)
    uvi = _np(mesh_tensors['uv_idx'], np.int32)
    keep += [tex, uv, uvi]
    m.tex, m.uv, m.uv_idx = tex.ctypes.data, uv.ctypes.data, uvi.ctypes.data
    m.texH, m.texW = tex.shape[0], tex.shape[1]
    m.vcolor = None
  else:
    vc = _np(mesh_tensors['vertex_color'], np.float32)
    keep.append(vc)
    m.vcolor = vc.ctypes.data
    m.tex = m.uv = m.uv_idx = None
  color = np.zeros((B, Ho, Wo, 3), np.float32)
  depth = np.zeros((B, Ho, Wo), np.float32)
  normal = np.zeros((B, Ho, Wo, 3), np.float32)
  xyz = np.zeros((B, Ho, Wo, 3), np.float32)
  rast = np.zeros((B, Ho, Wo, 4), np.float32)
  if light_dir is not None and np.array_equal(np.asarray(light_dir, dtype=float).reshape(-1), [0, 0, 1]):
    mode, lvec = 0, np.array([0, 0, -1], np.float32)
  elif light_dir is not None:
    mode, lvec = 1, -np.asarray(light_dir, dtype=np.float32).reshape(3)
  else:
    mode, lvec = 2, np.asarray(light_pos, dtype=np.float32).reshape(3)
  lvec = np.ascontiguousarray(lvec)
  lcol = None if light_color is None else np.ascontiguousarray(np.asarray(light_color, dtype=np.float32).reshape(3))
  rc = _lib().oracle_render_lit(ctypes.byref(m), ctypes.c_int(B), M.ctypes.data_as(ctypes.c_void_p),
                            pose.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(Ho), ctypes.c_int(Wo),
                            ctypes.c_int(1 if use_light else 0), ctypes.c_float(w_ambient), ctypes.c_float(w_diffuse),
                            ctypes.c_int(mode), lvec.ctypes.data_as(ctypes.c_void_p),
                            None if lcol is None else lcol.ctypes.data_as(ctypes.c_void_p),
                            color.ctypes.data_as(ctypes.c_void_p), depth.ctypes.data_as(ctypes.c_void_p),
                            normal.ctypes.data_as(ctypes.c_void_p), xyz.ctypes.data_as(ctypes.c_void_p),
                            rast.ctypes.data_as(ctypes.c_void_p))
  if rc != 0:
    raise MemoryError('oracle_render failed')
  extra['xyz_map'] = torch.from_numpy(xyz)
  extra['rast'] = torch.from_numpy(rast)
  normal_map = torch.from_numpy(normal) if (get_normal or use_light) else None
  return torch.from_numpy(color), torch.from_numpy(depth), normal_map

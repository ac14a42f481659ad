"""ctypes binding of libfoundationpose_amd.so (C-ABI: include/foundationpose_amd.h).

There is no CPU fallback: if the HIP library has not been built, importing an op raises with the
build command.  torch is used only for device memory and the current HIP stream.
"""
import ctypes
import os
import weakref
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int64, c_uint, c_void_p

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libfoundationpose_amd.so')
_lib = None

FP_REFINE_SHARED_TRANSLATION = 1      # include/foundationpose_amd.h
FP_NET_REFINE, FP_NET_SCORE = 0, 1


class FpTensor(Structure):
  _fields_ = [('name', c_char_p), ('data', c_void_p), ('ndim', c_int), ('shape', c_int64 * 4)]


class FpRefineCfg(Structure):
  _fields_ = [('crop_ratio', c_double), ('normalize_xyz', c_int), ('trans_rep_tanh', c_int),
              ('trans_normalizer', c_float * 3), ('rot_normalizer', c_float)]


class FpRenderOpts(Structure):
  """fp_render_opts (include/foundationpose_amd.h): the non-default arguments of nvdiffrast_render."""
  _fields_ = [('struct_size', ctypes.c_size_t), ('use_light', c_int), ('w_ambient', c_float), ('w_diffuse', c_float), ('light_mode', c_int), ('light_vec', c_float * 3),
              ('has_light_color', c_int), ('light_color', c_float * 3), ('has_projection', c_int), ('projection', c_double * 16), ('d_rast', c_void_p)]


class FpTrackArgs(Structure):
  """fp_track_args (include/foundationpose_amd.h): one tracking frame, every launch of it."""
  _fields_ = [('struct_size', ctypes.c_size_t), ('refine_net', c_void_p), ('score_net', c_void_p), ('mesh', c_void_p), ('d_rgb', c_void_p),
              ('rgb_is_u8', c_int), ('d_depth', c_void_p), ('H', c_int), ('W', c_int), ('K', c_void_p), ('mesh_diameter', c_double),
              ('refine_cfg', c_void_p), ('score_crop_ratio', c_double), ('score_normalize_xyz', c_int), ('iteration', c_int), ('n_hyp', c_int),
              ('d_perturb', c_void_p), ('model_center', c_float * 3), ('d_pose', c_void_p), ('d_pose_of_mesh', c_void_p), ('d_poses', c_void_p),
              ('d_scores', c_void_p), ('d_best', c_void_p), ('d_depth_f', c_void_p), ('d_xyz', c_void_p), ('d_rgb_f', c_void_p)]


class FpObjectBatch(Structure):
  _fields_ = [('mesh', c_void_p), ('d_rgb', c_void_p), ('d_geom', c_void_p), ('H', c_int), ('W', c_int), ('K', c_void_p),
              ('mesh_diameter', c_double), ('n', c_int)]


class FoundationPoseAmdError(RuntimeError):
  pass


_PROTOS = {
  'fp_last_error': (c_char_p, []),
  'fp_version': (c_int, []),
  'fp_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
  'fp_ctx_destroy': (c_int, [c_void_p]),
  'fp_ctx_reserve': (c_int, [c_void_p, c_int]),
  'fp_ctx_arena_generation': (c_int, [c_void_p]),
  'fp_mesh_create': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
  'fp_mesh_destroy': (c_int, [c_void_p]),
  'fp_crop_window_tf': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_double, c_double, c_int, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_render': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
  'fp_render_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, POINTER(FpRenderOpts), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
  'fp_render_net': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_float, c_void_p, c_void_p]),
  'fp_crop_observed': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p, c_void_p]),
  'fp_warp_nearest': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
  'fp_erode_depth': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p]),
  'fp_bilateral_filter_depth': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p]),
  'fp_depth2xyzmap': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p]),
  'fp_depth_prefilter': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_float,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
  'fp_depth2xyzmap_f64': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_mask_depth_stats': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
  'fp_net_create': (c_int, [c_void_p, c_int, POINTER(FpTensor), c_int, c_int, POINTER(c_void_p)]),
  'fp_net_destroy': (c_int, [c_void_p]),
  'fp_net_rot_dim': (c_int, [c_void_p]),
  'fp_refine_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_net_tokens': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
  'fp_score_features': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
  'fp_score_tail': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_score_tail_scores': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
  'fp_score_predict_rows_multi': (c_int, [c_void_p, c_void_p, POINTER(FpObjectBatch), c_int, c_double, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_track_frame': (c_int, [c_void_p, POINTER(FpTrackArgs), c_void_p]),
  'fp_pose_update': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_float, c_void_p, c_void_p]),
  'fp_pose_update_deepim': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p]),
  'fp_refine_predict': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_double, POINTER(FpRefineCfg), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_refine_predict_multi': (c_int, [c_void_p, c_void_p, POINTER(FpObjectBatch), c_int, POINTER(FpRefineCfg), c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_refine_predict_multi_flags': (c_int, [c_void_p, c_void_p, POINTER(FpObjectBatch), c_int, POINTER(FpRefineCfg), c_void_p, c_int, c_void_p, c_void_p, c_uint, c_void_p]),
  'fp_score_predict_features_multi': (c_int, [c_void_p, c_void_p, POINTER(FpObjectBatch), c_int, c_double, c_int, c_void_p, c_void_p, c_void_p]),
  'fp_score_predict_features': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_double, c_double, c_int, c_void_p, c_int, c_void_p, c_void_p]),
  'fp_conv3x3_band_f16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
  'fp_conv2d_f16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
  'fp_token_linear_f16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
  'fp_head_mlp_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 8 + [c_void_p, c_void_p]),
  'fp_conv3x3_wino_f16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
  'fp_attention_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
  'fp_cluster_poses': (c_int, [c_float, c_float, c_void_p, c_int, c_void_p, c_int, c_void_p]),
  'fp_prof_enable': (c_int, [c_void_p, c_int]),
  'fp_prof_read': (c_int, [c_void_p, c_char_p, POINTER(c_double), POINTER(c_int64), POINTER(c_double)]),
  'fp_prof_read_busy': (c_int, [c_void_p, c_char_p, POINTER(c_double)]),
  'fp_prof_reset': (c_int, [c_void_p]),
}


def exported_symbols():
  """Names declared in include/foundationpose_amd.h (kept in step by tests/test_abi.py)."""
  return sorted(_PROTOS)


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise FoundationPoseAmdError(
        f'{LIB_PATH} is missing: build the HIP library first (python -c "import __graft_entry__ as g; g.build()" '
        f'or make -C foundationpose_amd/csrc).  foundationpose_amd has no CPU fallback.')
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
      fn = getattr(L, name)
      fn.restype = res
      fn.argtypes = args
    _lib = L
  return _lib


def check(rc):
  if rc != 0:
    msg = lib().fp_last_error()
    raise FoundationPoseAmdError(f'libfoundationpose_amd error {rc}: {msg.decode() if msg else "?"}')


def ptr(t):
  """Device (or host) address of a contiguous torch tensor / numpy array; None -> NULL."""
  if t is None:
    return None
  if torch.is_tensor(t):
    assert t.is_contiguous(), 'tensor must be contiguous'
    return c_void_p(t.data_ptr())
  assert t.flags['C_CONTIGUOUS']
  return c_void_p(t.ctypes.data)


def stream_ptr(device=None):
  return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def k_ptr(K):
  """3x3 intrinsics as a float64 host array (kept alive by the caller)."""
  if torch.is_tensor(K):
    K = K.detach().cpu().numpy()
  Kd = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3))
  return Kd, c_void_p(Kd.ctypes.data)


class Context:
  """Per-device fp_ctx (workspace arena + profiling).  The opaque handle the reference calls
  `glctx` (dr.RasterizeCudaContext, src/estimater.py:102,168; main.py:42)."""
  _by_device = {}

  def __init__(self, device_index):
    self.device_index = device_index
    h = c_void_p()
    check(lib().fp_ctx_create(device_index, byref(h)))
    self.handle = h

  @classmethod
  def get(cls, device=None):
    if device is None:
      idx = torch.cuda.current_device()
    else:
      d = torch.device(device)
      idx = d.index if d.index is not None else torch.cuda.current_device()
    if idx not in cls._by_device:
      if not torch.cuda.is_available():
        raise FoundationPoseAmdError('no HIP device visible: foundationpose_amd needs an MI355X (no CPU fallback)')
      cls._by_device[idx] = Context(idx)
    return cls._by_device[idx]

  def reserve(self, max_hyp):
    check(lib().fp_ctx_reserve(self.handle, int(max_hyp)))

  def arena_generation(self):
    return int(lib().fp_ctx_arena_generation(self.handle))

  def prof_enable(self, on=True):
    """False / 0: off; 1: events around the dominant kernel class only; True / 2: around every class."""
    check(lib().fp_prof_enable(self.handle, 2 if on is True else int(on)))

  def prof_reset(self):
    check(lib().fp_prof_reset(self.handle))

  def prof_read(self, cls_name):
    ms, n, fl = c_double(), c_int64(), c_double()
    check(lib().fp_prof_read(self.handle, cls_name.encode(), byref(ms), byref(n), byref(fl)))
    busy = c_double()
    check(lib().fp_prof_read_busy(self.handle, cls_name.encode(), byref(busy)))
    return dict(total_ms=ms.value, launches=n.value, flops=fl.value, busy_ms=busy.value)


class DeviceMesh:
  """fp_mesh built from a reference-layout mesh_tensors dict (src/Utils.py:104-130)."""

  def __init__(self, ctx, mesh_tensors):
    f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
    i32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.int32))
    pos, faces, vn = f32(mesh_tensors['pos']), i32(mesh_tensors['faces']), f32(mesh_tensors['vnormals'])
    h = c_void_p()
    if 'tex' in mesh_tensors:
      tex = f32(mesh_tensors['tex']).reshape(mesh_tensors['tex'].shape[-3], mesh_tensors['tex'].shape[-2], 3)
      uv, uvi = f32(mesh_tensors['uv']), i32(mesh_tensors['uv_idx'])
      check(lib().fp_mesh_create(ctx.handle, ptr(pos), len(pos), ptr(faces), len(faces), ptr(vn), None, ptr(uv), len(uv), ptr(uvi),
                                 ptr(tex), tex.shape[0], tex.shape[1], byref(h)))
    else:
      vc = f32(mesh_tensors['vertex_color'])
      check(lib().fp_mesh_create(ctx.handle, ptr(pos), len(pos), ptr(faces), len(faces), ptr(vn), ptr(vc), None, 0, None, None, 0, 0,
                                 byref(h)))
    self.handle = h
    self.ctx = ctx

  def __del__(self):
    try:
      if self.handle:
        lib().fp_mesh_destroy(self.handle)
        self.handle = None
    except Exception:
      pass


_mesh_cache = {}      # id(mesh_tensors['pos']) -> [(device, pos version, signature, DeviceMesh), ...]; dropped when that tensor dies


def _mesh_signature(mesh_tensors):
  """Identity + in-place version of every tensor of the dict.  The entries hold the tensors themselves (strong references,
  except `pos`, whose death drops the entry), so an `id` cannot be recycled while its entry is alive."""
  return tuple((k, v, v._version) for k, v in sorted(mesh_tensors.items()) if torch.is_tensor(v) and k != 'pos')


def device_mesh(ctx, mesh_tensors):
  """The fp_mesh of a reference-layout mesh_tensors dict, uploaded once per (device, set of tensor objects).  The cache is
  keyed on the `pos` tensor OBJECT (its entry is dropped when the tensor dies - a dict built inside one call is not retained) and
  checked against the identity and in-place version counter of every other tensor, so a re-coloured, re-textured or edited
  mesh of the same size is uploaded again instead of being mistaken for the previous one at the same address."""
  pos = mesh_tensors['pos']
  sig = _mesh_signature(mesh_tensors)
  entries = _mesh_cache.get(id(pos))
  if entries is None:
    entries = _mesh_cache[id(pos)] = []
    weakref.finalize(pos, _mesh_cache.pop, id(pos), None)
  for e in entries:
    if e[0] == ctx.device_index and e[1] == pos._version and len(e[2]) == len(sig) and \
       all(a[0] == b[0] and a[1] is b[1] and a[2] == b[2] for a, b in zip(e[2], sig)):
      return e[3]
  m = DeviceMesh(ctx, mesh_tensors)
  entries[:] = [e for e in entries if e[0] != ctx.device_index][-3:]      # one live variant per device and pos object (+ a few devices)
  entries.append((ctx.device_index, pos._version, sig, m))
  return m


def object_batches(ctx, objects, geom_key):
  """list of dicts (rgb, <geom_key>, K, mesh_tensors, mesh_diameter, ob_in_cams) -> (FpObjectBatch array, concatenated
  poses (N,4,4) on the device, keep-alive list).  geom_key = 'xyz_map' (refiner) or 'depth' (scorer)."""
  dev = torch.device('cuda', ctx.device_index)
  arr = (FpObjectBatch * len(objects))()
  keep, poses = [], []
  for i, ob in enumerate(objects):
    rgb = torch.as_tensor(ob['rgb'], device=dev, dtype=torch.float).contiguous()
    geom = torch.as_tensor(ob[geom_key], device=dev, dtype=torch.float).contiguous()
    Kd, _ = k_ptr(ob['K'])
    dm = device_mesh(ctx, ob['mesh_tensors'])
    p = torch.as_tensor(ob['ob_in_cams'], device=dev, dtype=torch.float).reshape(-1, 4, 4)
    keep += [rgb, geom, Kd, dm]
    poses.append(p)
    arr[i].mesh, arr[i].d_rgb, arr[i].d_geom = dm.handle, rgb.data_ptr(), geom.data_ptr()
    arr[i].H, arr[i].W = rgb.shape[0], rgb.shape[1]
    arr[i].K = Kd.ctypes.data
    arr[i].mesh_diameter = float(ob['mesh_diameter'])
    arr[i].n = len(p)
  return arr, torch.cat(poses, 0).contiguous().clone(), keep


class DeviceNet:
  """fp_net built from a reference-layout state_dict (torch tensors).  The float32 host copies of the parameters are kept, so
  the network can be rebuilt on another device (`to`, the counterpart of nn.Module.to in src/estimater.py:97-100)."""

  def __init__(self, ctx, kind, state_dict, use_bn=True):
    self._host = [(k.encode(), np.ascontiguousarray(v.detach().cpu().float().numpy())) for k, v in state_dict.items()
                  if torch.is_tensor(v) and v.dtype.is_floating_point]
    self.kind = kind
    self.use_bn = bool(use_bn)
    self.handle = None
    self._create(ctx)

  def _create(self, ctx):
    arr = (FpTensor * len(self._host))()
    for n, (kb, a) in enumerate(self._host):
      arr[n].name = kb
      arr[n].data = a.ctypes.data
      arr[n].ndim = a.ndim
      for i in range(min(a.ndim, 4)):
        arr[n].shape[i] = a.shape[i]
    h = c_void_p()
    check(lib().fp_net_create(ctx.handle, self.kind, arr, len(self._host), 1 if self.use_bn else 0, byref(h)))
    self.handle = h
    self.ctx = ctx
    self.rot_dim = lib().fp_net_rot_dim(h) if self.kind == FP_NET_REFINE else 0

  def to(self, device):
    """Move the packed weights to `device` (a torch device / string): a no-op on the same device, otherwise the fp_net is
    rebuilt there from the host copies and the old one freed.  Returns self, like nn.Module.to."""
    ctx = Context.get(device)
    if ctx.device_index != self.ctx.device_index:
      old = self.handle
      self._create(ctx)
      lib().fp_net_destroy(old)
    return self

  def __del__(self):
    try:
      if self.handle:
        lib().fp_net_destroy(self.handle)
        self.handle = None
    except Exception:
      pass

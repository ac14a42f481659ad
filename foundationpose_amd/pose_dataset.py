"""BatchPoseData - the batch record that travels between the crop builders and the networks
(learning/datasets/pose_dataset.py:66-135).

The record keeps the reference's field names so callers written against it keep working.  On this path the
network input is ONE fused tensor (`net_input`: (2N,160,160,8) fp16, channels-last, rows [0,N) = side A,
[N,2N) = side B; channels rgb, xyz, 0, 0) written by the HIP kernels; the per-side planar tensors
(`rgbAs`, `xyz_mapAs`, ...) are float32 views made from it on demand by make_crop_data_batch.
"""
import torch

_FIELDS = ('rgbAs', 'rgbBs', 'depthAs', 'depthBs', 'normalAs', 'normalBs', 'maskAs', 'maskBs', 'poseA', 'poseB',
           'xyz_mapAs', 'xyz_mapBs', 'tf_to_crops', 'Ks', 'crop_masks', 'model_pts', 'mesh_diameters', 'labels')


class BatchPoseData:
  """Attribute bag; every non-None attribute is a tensor whose dim 0 is the hypothesis."""

  def __init__(self, **fields):
    unknown = set(fields) - set(_FIELDS) - {'net_input'}
    if unknown:
      raise TypeError(f'BatchPoseData has no field(s) {sorted(unknown)}')
    for k in _FIELDS:
      setattr(self, k, fields.get(k))
    self.net_input = fields.get('net_input')

  def _present(self):
    return [(k, v) for k, v in vars(self).items() if v is not None]

  def __len__(self):
    for k, v in self._present():
      return len(v) // 2 if k_is_paired(k) else len(v)
    return 0

  def pin_memory(self):
    """pose_dataset.py:111-118: in place, attributes that cannot be pinned are left alone."""
    for k, v in self._present():
      if torch.is_tensor(v) and not v.is_cuda:
        try:
          setattr(self, k, v.pin_memory())
        except RuntimeError:
          pass
    return self

  def cuda(self):
    """pose_dataset.py:120-127: in place."""
    for k, v in self._present():
      if torch.is_tensor(v):
        setattr(self, k, v.cuda())
    return self

  def select_by_indices(self, ids):
    """pose_dataset.py:129-134: a new record holding rows `ids` of every present field."""
    out = BatchPoseData()
    ids = torch.as_tensor(ids)
    for k, v in self._present():
      i = ids.to(v.device)
      if k_is_paired(k):
        n = len(v) // 2
        setattr(out, k, torch.cat([v[:n][i], v[n:][i]], 0))
      else:
        setattr(out, k, v[i])
    return out


def k_is_paired(name):
  """`net_input` stacks side A over side B, so it has 2 rows per hypothesis."""
  return name == 'net_input'


def planar_views(net_input):
  """(2N,h,w,8) fp16 net tensor -> rgbAs, xyz_mapAs, rgbBs, xyz_mapBs as (N,3,h,w) float32."""
  n = len(net_input) // 2
  p = net_input[..., :6].permute(0, 3, 1, 2).float()
  return p[:n, :3], p[:n, 3:], p[n:, :3], p[n:, 3:]

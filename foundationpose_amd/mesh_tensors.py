"""make_mesh_tensors: mesh object -> dict of tensors the renderer consumes (reference src/Utils.py:104-130)."""
import logging
import numpy as np
import torch


def _is_textured(visual):
  return hasattr(visual, 'uv') and (hasattr(visual, 'image') or hasattr(visual, 'material'))


def make_mesh_tensors(mesh, device='cuda', max_tex_size=None):
  """Same keys / dtypes / value ranges as the reference: pos (V,3) f32, faces (F,3) i32, vnormals (V,3) f32
  and either vertex_color (V,3) f32 in [0,1] or tex (1,Ht,Wt,3) f32 in [0,1] + uv (V,2) (v flipped) + uv_idx."""
  mesh_tensors = {}
  visual = mesh.visual
  if _is_textured(visual):
    if hasattr(visual, 'image'):
      img = np.asarray(visual.image)
    else:
      img = np.array(visual.material.image.convert('RGB'))
    img = img[..., :3]
    if max_tex_size is not None:
      max_size = max(img.shape[0], img.shape[1])
      if max_size > max_tex_size:
        # nearest-neighbour decimation (the reference calls cv2.resize, absent here); cold path
        scale = max_tex_size / max_size
        ys = (np.arange(int(round(img.shape[0] * scale))) / scale).astype(int).clip(0, img.shape[0] - 1)
        xs = (np.arange(int(round(img.shape[1] * scale))) / scale).astype(int).clip(0, img.shape[1] - 1)
        img = img[ys][:, xs]
    mesh_tensors['tex'] = torch.as_tensor(np.ascontiguousarray(img), device=device, dtype=torch.float)[None] / 255.0
    mesh_tensors['uv_idx'] = torch.as_tensor(np.asarray(mesh.faces), device=device, dtype=torch.int)
    uv = torch.as_tensor(np.asarray(visual.uv), device=device, dtype=torch.float).clone()
    uv[:, 1] = 1 - uv[:, 1]
    mesh_tensors['uv'] = uv
  else:
    if getattr(visual, 'vertex_colors', None) is None:
      logging.info("WARN: mesh doesn't have vertex_colors, assigning a pure color")
      visual.vertex_colors = np.tile(np.array([128, 128, 128]).reshape(1, 3), (len(mesh.vertices), 1))
    mesh_tensors['vertex_color'] = torch.as_tensor(np.asarray(visual.vertex_colors)[..., :3], device=device, dtype=torch.float) / 255.0
  mesh_tensors.update({
    'pos': torch.tensor(np.asarray(mesh.vertices), device=device, dtype=torch.float),
    'faces': torch.tensor(np.asarray(mesh.faces), device=device, dtype=torch.int),
    'vnormals': torch.tensor(np.asarray(mesh.vertex_normals), device=device, dtype=torch.float),
  })
  return mesh_tensors

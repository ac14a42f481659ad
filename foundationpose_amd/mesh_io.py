"""Mesh / intrinsics ingestion for the estimator: Wavefront OBJ -> SimpleMesh with
angle-weighted vertex normals and vertex colours (`v x y z r g b`) or a uv-mapped texture array."""
import numpy as np

from .synthetic import SimpleMesh, TextureVisual


def load_obj(path, texture_image=None):
  """Triangles and polygons (fan-triangulated); `v/vt/vn` index forms; negative (relative) indices.
  Vertices are split per (v, vt) pair when texture coordinates are present, so faces index uv directly
  (make_mesh_tensors uses `mesh.faces` as `uv_idx`, src/Utils.py:115)."""
  v, vc, vt, corners = [], [], [], []
  with open(path) as f:
    for line in f:
      p = line.split()
      if not p or p[0].startswith('#'):
        continue
      if p[0] == 'v':
        v.append([float(x) for x in p[1:4]])
        vc.append([float(x) for x in p[4:7]] if len(p) >= 7 else None)
      elif p[0] == 'vt':
        vt.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0])
      elif p[0] == 'f':
        idx = []
        for tok in p[1:]:
          parts = tok.split('/')
          vi = int(parts[0])
          ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
          idx.append((vi - 1 if vi > 0 else len(v) + vi, (ti - 1 if ti > 0 else len(vt) + ti) if ti else -1))
        for k in range(1, len(idx) - 1):
          corners.append((idx[0], idx[k], idx[k + 1]))
  if not v or not corners:
    raise ValueError(f'{path}: no geometry')
  v = np.asarray(v, dtype=np.float64)
  use_uv = bool(vt) and all(c[1] >= 0 for tri in corners for c in tri)
  if use_uv:
    remap, verts, uvs, faces = {}, [], [], []
    for tri in corners:
      face = []
      for key in tri:
        if key not in remap:
          remap[key] = len(verts)
          verts.append(v[key[0]])
          uvs.append(vt[key[1]])
        face.append(remap[key])
      faces.append(face)
    mesh = SimpleMesh(np.asarray(verts), np.asarray(faces))
    if texture_image is not None:
      mesh.visual = TextureVisual(uv=np.asarray(uvs, dtype=np.float64), image=np.asarray(texture_image))
    return mesh
  faces = np.asarray([[c[0] for c in tri] for tri in corners], dtype=np.int64)
  colors = None
  if all(c is not None for c in vc):
    rgb = np.clip(np.asarray(vc) * 255.0, 0, 255).astype(np.uint8)
    colors = np.concatenate([rgb, np.full((len(rgb), 1), 255, np.uint8)], 1)
  return SimpleMesh(v, faces, vertex_colors=colors)


def save_obj(mesh, path):
  colors = np.asarray(mesh.visual.vertex_colors)[:, :3] / 255.0 if hasattr(mesh.visual, 'vertex_colors') else None
  with open(path, 'w') as f:
    for i, p in enumerate(mesh.vertices):
      c = '' if colors is None else ' %.6f %.6f %.6f' % tuple(colors[i])
      f.write('v %.9g %.9g %.9g%s\n' % (p[0], p[1], p[2], c))
    for t in mesh.faces:
      f.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))


def load_intrinsics(path):
  """3x3 matrix as 9 whitespace/comma separated numbers (the cam_K.txt convention)."""
  vals = np.fromstring(open(path).read().replace(',', ' '), sep=' ')
  if vals.size != 9:
    raise ValueError(f'{path}: expected 9 numbers, got {vals.size}')
  return vals.reshape(3, 3).astype(np.float64)

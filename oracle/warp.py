"""Oracle: kornia 0.7.2 `geometry.transform.warp_perspective` restated on torch-CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.  kornia is pinned at requirements.txt:81 but is not
in the reference checkout nor in this image: PARITY UNPINNED; the behaviour below is kornia
0.7.2's published implementation (normalize_homography -> inverse -> meshgrid -> grid_sample),
SURVEY.md Appendix A3.  Reference call sites: learning/training/predict_pose_refine.py:63-76,
predict_score.py:89-99, learning/datasets/h5_dataset.py:89-104,147-161.
"""
import torch
import torch.nn.functional as F


def normal_transform_pixel(height, width, eps=1e-14, dtype=torch.float32):
  tr = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
  width_denom = eps if width == 1 else width - 1.0
  height_denom = eps if height == 1 else height - 1.0
  tr[0, 0] = tr[0, 0] * 2.0 / width_denom
  tr[1, 1] = tr[1, 1] * 2.0 / height_denom
  return tr[None]


def normalize_homography(dst_pix_trans_src_pix, dsize_src, dsize_dst):
  src_h, src_w = dsize_src
  dst_h, dst_w = dsize_dst
  dt = dst_pix_trans_src_pix.dtype
  src_norm_trans_src_pix = normal_transform_pixel(src_h, src_w, dtype=dt)
  src_pix_trans_src_norm = torch.linalg.inv(src_norm_trans_src_pix)
  dst_norm_trans_dst_pix = normal_transform_pixel(dst_h, dst_w, dtype=dt)
  return dst_norm_trans_dst_pix @ (dst_pix_trans_src_pix @ src_pix_trans_src_norm)


def _convert_points_from_homogeneous(points, eps=1e-8):
  z_vec = points[..., -1:]
  mask = torch.abs(z_vec) > eps
  scale = torch.where(mask, 1.0 / (z_vec + eps), torch.ones_like(z_vec))
  return scale * points[..., :-1]


def warp_grid(M, src_hw, dsize):
  """The sampling grid kornia hands to F.grid_sample: (B,h,w,2) normalised source coords."""
  H, W = src_hw
  h_out, w_out = dsize
  B = M.shape[0]
  dst_norm_trans_src_norm = normalize_homography(M, (H, W), (h_out, w_out))
  src_norm_trans_dst_norm = torch.linalg.inv(dst_norm_trans_src_norm)
  xs = torch.linspace(-1, 1, w_out, dtype=M.dtype)
  ys = torch.linspace(-1, 1, h_out, dtype=M.dtype)
  gy, gx = torch.meshgrid(ys, xs, indexing='ij')
  grid = torch.stack([gx, gy], dim=-1)[None].expand(B, h_out, w_out, 2)
  pts_h = torch.cat([grid, torch.ones_like(grid[..., :1])], dim=-1)          # (B,h,w,3)
  T = src_norm_trans_dst_norm[:, None, None]                                  # (B,1,1,3,3)
  out_h = (T @ pts_h[..., None])[..., 0]
  return _convert_points_from_homogeneous(out_h)


def warp_perspective(src, M, dsize, mode='bilinear', padding_mode='zeros', align_corners=False):
  """src (B,C,H,W) (may be an expanded view), M (B,3,3) -> (B,C,h,w)."""
  B, _, H, W = src.shape
  grid = warp_grid(M, (H, W), dsize)
  return F.grid_sample(src, grid, align_corners=align_corners, mode=mode, padding_mode=padding_mode)


TIE_EPS = 1e-4


def round_half_even_snapped(x):
  """nearbyint with coordinates closer than TIE_EPS px to a half-integer treated as EXACT ties.

  Why: with the axis-aligned crop transforms of this path some nearest-neighbour lookups are exact
  ties in real arithmetic - e.g. the scorer's crop->full-res warp (h5_dataset.py:158) maps the
  full-res pixel on the window edge to crop coordinate (tf q)*160/159 - 0.5 = -0.5 exactly.
  kornia's float32 normalise/invert chain carries ~1e-5 px of noise, so in the reference itself
  such a lookup lands on either side of the tie at random per hypothesis (implementation noise, not
  algorithm).  The oracle and the HIP kernel both resolve every coordinate within TIE_EPS of a tie
  as exact arithmetic + round-half-to-even does, which makes the result deterministic."""
  f = torch.floor(x)
  t = x - f
  tie = (t - 0.5).abs() < TIE_EPS
  even = torch.where(torch.remainder(f, 2) == 0, f, f + 1)
  return torch.where(tie, even, torch.floor(x + 0.5)).long()


def warp_perspective_nearest(src, M, dsize):
  """warp_perspective(mode='nearest', padding_mode='zeros', align_corners=False): the same kornia
  chain evaluated in float64, ties resolved by round_half_even_snapped.  src (B,C,H,W) float32."""
  B, C, H, W = src.shape
  grid = warp_grid(M.double(), (H, W), dsize)
  x = ((grid[..., 0] + 1) * W - 1) / 2
  y = ((grid[..., 1] + 1) * H - 1) / 2
  ix, iy = round_half_even_snapped(x), round_half_even_snapped(y)
  inb = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
  lin = (iy.clamp(0, H - 1) * W + ix.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
  out = torch.gather(src.reshape(B, C, H * W), 2, lin).reshape(B, C, dsize[0], dsize[1])
  return out * inb[:, None].to(out.dtype)


def source_affine_coeffs(M, src_hw, dsize):
  """Closed form of the map above for an axis-aligned M = [[sx,0,tx],[0,sy,ty],[0,0,1]]:
  output pixel (i,j) samples un-normalised source position
      x = ax*i + bx,   y = ay*j + by        (grid_sample align_corners=False pixel units)
  with, in exact arithmetic,  q = M^-1 p  and  x = q*W/(W-1) - 0.5.
  Returned in float64 (B,4) = ax,bx,ay,by.  Used to document / cross-check the fused HIP gather."""
  H, W = src_hw
  M = M.double()
  sx, sy, tx, ty = M[:, 0, 0], M[:, 1, 1], M[:, 0, 2], M[:, 1, 2]
  ax = (1.0 / sx) * W / (W - 1.0)
  bx = (-tx / sx) * W / (W - 1.0) - 0.5
  ay = (1.0 / sy) * H / (H - 1.0)
  by = (-ty / sy) * H / (H - 1.0) - 0.5
  return torch.stack([ax, bx, ay, by], dim=-1)

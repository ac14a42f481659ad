"""Debug canvases of `get_vis=True` (predict_pose_refine.py:241-293, predict_score.py:27-52,219-224; src/Utils.py:293-300,456-478).

Outside the hot path and host-side only.  The reference draws them with cv2 (JET colour map, `resize`, `putText`) and torchvision's
`make_grid`; neither is in this image, so the layout is rebuilt with numpy / torch and two things differ, on purpose and visibly
documented here: the JET map is the analytic one (clip(1.5 - |4x - k|), OpenCV's LUT differs by at most a few grey levels) and the text
labels (`id:..`, `score:..`; `cv_draw_text`, src/Utils.py:630-653) are drawn with a built-in 5x7 bitmap font instead of cv2's Hershey
strokes: same text, colour and anchor, other glyph shapes.  Rows are in the same order, so row i of the scorer's canvas is the i-th best
hypothesis."""
import numpy as np
import torch


def _jet(v_u8):
  """uint8 (H,W) -> RGB uint8 (H,W,3): analytic JET (blue -> cyan -> yellow -> red)."""
  x = v_u8.astype(np.float32) / 255.0
  r = np.clip(1.5 - np.abs(4 * x - 3), 0, 1)
  g = np.clip(1.5 - np.abs(4 * x - 2), 0, 1)
  b = np.clip(1.5 - np.abs(4 * x - 1), 0, 1)
  return (np.stack([r, g, b], -1) * 255).astype(np.uint8)


def depth_to_vis(depth, zmin=None, zmax=None, mode='rgb', inverse=True):
  """Depth image -> uint8 picture, the rule of src/Utils.py:456-478: `inverse` shows zmin / depth (invalid pixels, depth < 1 mm,
  black); otherwise the depth is normalised to [zmin, zmax] and every pixel at or beyond either bound is drawn at full scale.
  zmin / zmax default to the image's own extremes.  mode 'gray' -> (H,W), 'rgb' -> JET-coloured (H,W,3)."""
  if mode not in ('gray', 'rgb'):
    raise RuntimeError(f"depth_to_vis: mode {mode!r} (the reference knows 'gray' and 'rgb')")
  d = np.asarray(depth, dtype=np.float32)
  lo = d.min() if zmin is None else zmin
  hi = d.max() if zmax is None else zmax
  with np.errstate(divide='ignore', invalid='ignore'):
    if inverse:
      level = np.where(d < 0.001, 0.0, lo / (d + 1e-8))
    else:
      inside = (d > lo) & (d < hi)
      level = np.where(inside, (d - lo) / (hi - lo), 1.0)
    # src/Utils.py:473-476: 'gray' clips to [0, 255] before the uint8 cast, 'rgb' casts (vis * 255) WITHOUT a clip - with a caller-supplied
    # zmin above the nearest depth the inverse levels exceed 1 and wrap in the cast, as they do in the reference
    if mode == 'gray':
      return (level * 255).clip(0, 255).astype(np.uint8)
    return _jet((level * 255).astype(np.uint8))


# 5x7 bitmap glyphs (rows top to bottom, 5 bits each, MSB = left column) for the labels of the debug canvases
_GLYPHS = {
  '0': (14, 17, 19, 21, 25, 17, 14), '1': (4, 12, 4, 4, 4, 4, 14), '2': (14, 17, 1, 2, 4, 8, 31), '3': (31, 2, 4, 2, 1, 17, 14),
  '4': (2, 6, 10, 18, 31, 2, 2), '5': (31, 16, 30, 1, 1, 17, 14), '6': (6, 8, 16, 30, 17, 17, 14), '7': (31, 1, 2, 4, 8, 8, 8),
  '8': (14, 17, 17, 14, 17, 17, 14), '9': (14, 17, 17, 15, 1, 2, 12),
  'a': (0, 0, 14, 1, 15, 17, 15), 'b': (16, 16, 22, 25, 17, 17, 30), 'c': (0, 0, 14, 16, 16, 17, 14), 'd': (1, 1, 13, 19, 17, 17, 15),
  'e': (0, 0, 14, 17, 31, 16, 14), 'f': (6, 9, 8, 28, 8, 8, 8), 'g': (0, 15, 17, 17, 15, 1, 14), 'h': (16, 16, 22, 25, 17, 17, 17),
  'i': (4, 0, 12, 4, 4, 4, 14), 'j': (2, 0, 6, 2, 2, 18, 12), 'k': (16, 16, 18, 20, 24, 20, 18), 'l': (12, 4, 4, 4, 4, 4, 14),
  'm': (0, 0, 26, 21, 21, 17, 17), 'n': (0, 0, 22, 25, 17, 17, 17), 'o': (0, 0, 14, 17, 17, 17, 14), 'p': (0, 30, 17, 17, 30, 16, 16),
  'q': (0, 13, 19, 17, 15, 1, 1), 'r': (0, 0, 22, 25, 16, 16, 16), 's': (0, 0, 14, 16, 14, 1, 30), 't': (8, 8, 28, 8, 8, 9, 6),
  'u': (0, 0, 17, 17, 17, 19, 13), 'v': (0, 0, 17, 17, 17, 10, 4), 'w': (0, 0, 17, 17, 21, 21, 10), 'x': (0, 0, 17, 10, 4, 10, 17),
  'y': (0, 17, 17, 17, 15, 1, 14), 'z': (0, 0, 31, 2, 4, 8, 31),
  ':': (0, 12, 12, 0, 12, 12, 0), ',': (0, 0, 0, 0, 12, 4, 8), '.': (0, 0, 0, 0, 0, 12, 12), '-': (0, 0, 0, 31, 0, 0, 0),
  '+': (0, 4, 4, 31, 4, 4, 0), '_': (0, 0, 0, 0, 0, 0, 31), '/': (1, 1, 2, 4, 8, 16, 16), ' ': (0, 0, 0, 0, 0, 0, 0),
}
_UNKNOWN_GLYPH = (31, 17, 17, 17, 17, 17, 31)


def _text_mask(line, k):
  """Boolean (7k, 6k * len(line)) mask of `line` in the 5x7 font scaled by the integer k (one blank column between glyphs)."""
  rows = np.zeros((7, 6 * len(line)), dtype=bool)
  for n, ch in enumerate(line):
    g = _GLYPHS.get(ch, _GLYPHS.get(ch.lower(), _UNKNOWN_GLYPH))
    for r in range(7):
      for c in range(5):
        rows[r, 6 * n + c] = (g[r] >> (4 - c)) & 1
  return np.kron(rows, np.ones((k, k), dtype=bool))


def cv_draw_text(img, text, uv_top_left, color=(255, 255, 255), fontScale=0.5, thickness=1, fontFace=None, outline_color=None, line_spacing=1.5):
  """src/Utils.py:630-653 without cv2: every line of `text` is drawn into `img` (H,W,3; modified in place and returned) with its top
  left corner at `uv_top_left`, moved inside the image as the reference moves it, the next line `line_spacing` text heights below.
  The glyphs are a 5x7 bitmap font scaled to about the height of cv2's FONT_HERSHEY_SIMPLEX at `fontScale` (0.5 -> 14 pixels);
  `thickness` and `fontFace` are accepted and ignored; `outline_color` draws the mask dilated by one pixel underneath."""
  H, W = img.shape[:2]
  uv = np.array(uv_top_left, dtype=float)
  assert uv.shape == (2,)
  k = max(1, int(round(fontScale * 4 - 1e-9)))
  for line in text.splitlines():
    mask = _text_mask(line, k)
    h, w = mask.shape
    # the reference's four `while` loops on the bottom-left corner: right until u >= 0, left until u + w < W (a line wider than the
    # image loses its beginning), up until the baseline is inside, down until the top is
    u0 = min(max(int(uv[0]), 0), W - 1 - w)
    v0 = max(min(int(uv[1]) + h, H - 1) - h, 0)
    cu, cv = max(0, -u0), max(0, -v0)                  # clipped columns / rows of the mask
    mask = mask[cv:max(cv, min(h, H - v0)), cu:max(cu, min(w, W - u0))]
    u0, v0 = max(u0, 0), max(v0, 0)
    region = img[v0:v0 + mask.shape[0], u0:u0 + mask.shape[1]]
    if outline_color is not None:
      fat = mask.copy()
      fat[1:] |= mask[:-1]; fat[:-1] |= mask[1:]; fat[:, 1:] |= mask[:, :-1]; fat[:, :-1] |= mask[:, 1:]
      region[fat] = np.asarray(outline_color, dtype=img.dtype)[:region.shape[-1]] if region.ndim == 3 else outline_color[0]
    region[mask] = np.asarray(color, dtype=img.dtype)[:region.shape[-1]] if region.ndim == 3 else color[0]
    uv[1] = v0 + h * line_spacing                   # (h: the full text height, also for a clipped line)
  return img


def make_grid_image(imgs, nrow, padding=5, pad_value=255):
  """src/Utils.py:293-300 = torchvision.utils.make_grid on (B,H,W,C) images: `nrow` images per row, `padding` pixels of
  `pad_value` around and between them; returns uint8 (H',W',C)."""
  imgs = [np.asarray(im).astype(np.uint8) for im in imgs]          # (the reference truncates to uint8 after the grid: same values)
  B = len(imgs)
  H, W, C = imgs[0].shape
  if B == 1:
    return imgs[0]                                     # make_grid returns a single image unpadded
  xmaps = min(nrow, B)
  ymaps = int(np.ceil(B / xmaps))
  h, w = H + padding, W + padding
  grid = np.full((h * ymaps + padding, w * xmaps + padding, C), pad_value, dtype=np.uint8)
  for k in range(B):
    y, x = divmod(k, xmaps)
    grid[y * h + padding:y * h + padding + H, x * w + padding:x * w + padding + W] = imgs[k]
  return grid


def write_png(path, img):
  """uint8 (H,W,3) -> an 8-bit RGB PNG (what imageio.imwrite does for the debug canvases, src/estimater.py:217,221); zlib only."""
  import struct
  import zlib
  img = np.ascontiguousarray(img, dtype=np.uint8)
  H, W, _ = img.shape
  raw = np.concatenate([np.zeros((H, 1), np.uint8), img.reshape(H, W * 3)], axis=1).tobytes()       # filter type 0 per scanline

  def chunk(tag, data):
    body = tag + data
    return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
  with open(path, 'wb') as f:
    f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw, 6)) +
            chunk(b'IEND', b''))


def _row_images(pose_data, i):
  rgbA = (pose_data.rgbAs[i] * 255).permute(1, 2, 0).data.cpu().numpy()
  rgbB = (pose_data.rgbBs[i] * 255).permute(1, 2, 0).data.cpu().numpy()
  H, W = rgbA.shape[:2]
  if pose_data.depthAs is not None:
    dA = pose_data.depthAs[i].data.cpu().numpy().reshape(H, W)
    dB = pose_data.depthBs[i].data.cpu().numpy().reshape(H, W)
  else:
    dA = pose_data.xyz_mapAs[i][2].data.cpu().numpy().reshape(H, W)
    dB = pose_data.xyz_mapBs[i][2].data.cpu().numpy().reshape(H, W)
  return rgbA, rgbB, dA, dB


def refine_canvas(pose_data_before, pose_data_after, padding=2):
  """predict_pose_refine.py:241-293: per hypothesis a row (render, observed crop, render depth, observed depth; the depths share one
  colour range), all rows stacked, the start poses on the left and the refined poses on the right."""
  halves = []
  for pd in (pose_data_before, pose_data_after):
    rows = []
    for i in range(len(pd.rgbAs)):
      rgbA, rgbB, dA, dB = _row_images(pd, i)
      zmin, zmax = min(dA.min(), dB.min()), max(dA.max(), dB.max())
      row = [rgbA, rgbB, depth_to_vis(dA, zmin=zmin, zmax=zmax, inverse=False), depth_to_vis(dB, zmin=zmin, zmax=zmax, inverse=False)]
      row = make_grid_image(row, nrow=len(row), padding=padding, pad_value=255)
      if pd is pose_data_before:             # predict_pose_refine.py:265: the rows of the start poses carry their index
        row = cv_draw_text(row, text=f'id:{i}', uv_top_left=(10, 10), color=(0, 255, 0), fontScale=0.5)
      rows.append(row)
    halves.append(make_grid_image(rows, nrow=1, padding=padding, pad_value=255))
  return make_grid_image(halves, nrow=2, padding=padding, pad_value=255)


def score_canvas(pose_data, ids, scores, pad_margin=5):
  """predict_score.py:27-52: hypotheses in the order `ids` (best first), each a row render | render depth | observed crop | observed
  depth, scaled to 100 pixels of height (bilinear), 5-pixel white separators."""
  assert len(scores) == len(ids)
  canvas = []
  for i in [int(x) for x in ids]:
    rgbA, rgbB, dA, dB = _row_images(pose_data, i)
    zmin, zmax = dA.min(), dA.max()
    pad = np.ones((rgbA.shape[0], pad_margin, 3)) * 255
    row = np.concatenate([rgbA, pad, depth_to_vis(dA, zmin=zmin, zmax=zmax, inverse=False), pad, rgbB, pad,
                          depth_to_vis(dB, zmin=zmin, zmax=zmax, inverse=False)], axis=1)
    s = 100 / row.shape[0]
    t = torch.from_numpy(np.ascontiguousarray(row, dtype=np.float32)).permute(2, 0, 1)[None]
    t = torch.nn.functional.interpolate(t, size=(int(round(row.shape[0] * s)), int(round(row.shape[1] * s))), mode='bilinear', align_corners=False)
    row = t[0].permute(1, 2, 0).numpy()
    row = cv_draw_text(row, text=f'id:{i}, score:{float(scores[i]):.3f}', uv_top_left=(10, 10), color=(0, 255, 0), fontScale=0.5)      # predict_score.py:47
    canvas.append(row)
    canvas.append(np.ones((pad_margin, row.shape[1], 3)) * 255)
  return np.concatenate(canvas, axis=0).astype(np.uint8)

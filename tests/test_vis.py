"""Debug canvases (foundationpose_amd/vis.py): the pieces whose definition does not depend on cv2 are checked exactly
(make_grid geometry = torchvision.utils.make_grid, depth_to_vis's clipping rule of src/Utils.py:456-478, the PNG container)."""
import struct
import zlib

import numpy as np
import pytest
import torch

from foundationpose_amd import vis as V


def test_make_grid_geometry_is_torchvisions():
  imgs = [np.full((4, 6, 3), 10 * (k + 1), np.float64) for k in range(5)]
  g = V.make_grid_image(imgs, nrow=2, padding=2, pad_value=255)
  assert g.dtype == np.uint8 and g.shape == (3 * (4 + 2) + 2, 2 * (6 + 2) + 2, 3)       # ceil(5/2) rows, 2 columns, padding around and between
  assert (g[:2] == 255).all() and (g[:, :2] == 255).all()
  for k in range(5):
    y, x = divmod(k, 2)
    assert (g[y * 6 + 2:y * 6 + 6, x * 8 + 2:x * 8 + 8] == 10 * (k + 1)).all()
  assert (g[14:18, 10:16] == 255).all()                                                  # the empty sixth cell
  assert V.make_grid_image(imgs[:1], nrow=1, padding=2).shape == (4, 6, 3)                 # a single image comes back unpadded


def test_depth_to_vis_rules():
  d = np.array([[0.0, 0.5, 0.75, 1.0, 2.0]], np.float32)
  g = V.depth_to_vis(d, zmin=0.5, zmax=1.0, mode='gray', inverse=False)
  assert g.tolist() == [[255, 255, 127, 255, 255]]                  # clipped values and both ends count as invalid -> 1.0
  g = V.depth_to_vis(d, zmin=0.5, mode='gray', inverse=True)
  assert g[0, 0] == 0 and g[0, 1] in (254, 255) and g[0, 3] == 127   # zmin / depth; depth < 1 mm -> 0
  c = V.depth_to_vis(d, zmin=0.5, zmax=1.0, inverse=False)
  assert c.shape == (1, 5, 3) and c.dtype == np.uint8
  assert c[0, 2].tolist() == [127, 255, 127] or abs(int(c[0, 2, 1]) - 255) <= 2          # mid-range of JET is green
  assert c[0, 0, 0] > 100 and c[0, 0, 2] == 0                       # 1.0 -> red end
  # src/Utils.py:473-476: 'gray' clips before the uint8 cast, 'rgb' does not - a level above 1 (inverse mode, depth nearer than a given zmin)
  # saturates in 'gray' and WRAPS in 'rgb', exactly as (vis * 255).astype(np.uint8) does in the reference
  near = np.array([[0.4, 0.5]], np.float32)                          # zmin / depth = 1.25 -> 318.75 -> uint8 62 ; 1.0 -> 255
  # gray: 1.25 clips to 1.0 -> 255; 0.5 / (0.5 + 1e-8) is exactly 1.0 in float32 (1e-8 is below half an ulp of 0.5) -> 255
  assert V.depth_to_vis(near, zmin=0.5, mode='gray', inverse=True).tolist() == [[255, 255]]
  wrapped = (np.float32(0.5) / (near + np.float32(1e-8)) * 255).astype(np.uint8)
  assert wrapped[0, 0] < 100                                          # the cast wrapped
  assert np.array_equal(V.depth_to_vis(near, zmin=0.5, mode='rgb', inverse=True), V._jet(wrapped))


def test_png_container(tmp_path):
  img = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
  p = tmp_path / 'x.png'
  V.write_png(str(p), img)
  raw = p.read_bytes()
  assert raw[:8] == b'\x89PNG\r\n\x1a\n'
  pos, chunks = 8, {}
  while pos < len(raw):
    n, tag = struct.unpack('>I4s', raw[pos:pos + 8])
    body = raw[pos + 8:pos + 8 + n]
    assert struct.unpack('>I', raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
    chunks[tag] = body
    pos += 12 + n
  assert struct.unpack('>IIBBBBB', chunks[b'IHDR']) == (7, 5, 8, 2, 0, 0, 0)
  rows = np.frombuffer(zlib.decompress(chunks[b'IDAT']), np.uint8).reshape(5, 1 + 21)
  assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(5, 7, 3), img)


@pytest.mark.gpu
def test_get_vis_canvases(tmp_path):
  """get_vis=True of both predictors and debug=2 of the estimator (predict_pose_refine.py:241-293, predict_score.py:219-224,
  src/estimater.py:216-221,263-266): canvases of the reference's layout; their crops are the ones the networks saw."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.estimater import FoundationPose
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
  from foundationpose_amd.predict_score import ScorePredictor
  from oracle import geometry as G
  from . import util
  sc = util.scene(0)
  n = 3
  poses = util.hypotheses(sc, n, jitter_seed=2)
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  xyz_map = G.depth2xyzmap(depth, sc['K'])
  mt = util.to_dev(sc['mt'])
  refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0, head_gain=0.1), cfg=REFINE_DEFAULT)
  scorer = ScorePredictor(state_dict=S.make_score_state_dict(1), cfg=SCORE_DEFAULT)
  kw = dict(mesh_tensors=mt, mesh_diameter=sc['diameter'])
  refined, vis = refiner.predict(sc['rgb'], depth, sc['K'], poses, xyz_map, iteration=1, get_vis=True, **kw)
  row_h, row_w = 160 + 2 * 2, 4 * (160 + 2) + 2
  half_h, half_w = n * (row_h + 2) + 2, row_w + 2 * 2
  assert vis.dtype == np.uint8 and vis.shape == (half_h + 4, 2 * (half_w + 2) + 2, 3)
  pd = make_crop_data_batch((160, 160), poses, None, sc['rgb'], depth, sc['K'], 1.2, xyz_map, cfg=refiner.cfg, **kw)
  first = vis[2 + 2 + 2:2 + 2 + 2 + 160, 2 + 2 + 2:2 + 2 + 2 + 160]                    # grid / column / row paddings, then rgbA of hypothesis 0
  want = (pd.rgbAs[0] * 255).permute(1, 2, 0).cpu().numpy().astype(np.uint8)
  label = np.zeros((160, 160), bool)
  label[8:8 + 14, 8:8 + 48] = True                               # 'id:0' at (10,10) of the row = (8,8) of its first crop: 4 glyphs of 12 x 14 pixels
  np.testing.assert_array_equal(first[~label], want[~label])
  green = (first[label] == np.array([0, 255, 0], np.uint8)).all(-1)
  assert 40 < int(green.sum()) < 14 * 48 // 2                     # (the glyphs' pixels, pure green; predict_pose_refine.py:265)
  refined_again, none = refiner.predict(sc['rgb'], depth, sc['K'], poses, xyz_map, iteration=1, **kw)
  assert none is None and torch.equal(refined, refined_again)
  scores, svis = scorer.predict(sc['rgb'], depth, sc['K'], refined, get_vis=True, **kw)
  w_full = 4 * 160 + 3 * 5
  assert svis.dtype == np.uint8 and svis.shape == (n * (100 + 5), int(round(w_full * 100 / 160)), 3)
  for r in range(n):                                              # predict_score.py:47: 'id:.., score:..' in green at (10,10) of every row
    box = svis[r * 105 + 10:r * 105 + 24, 10:10 + 12 * 10]
    assert int((box == np.array([0, 255, 0], np.uint8)).all(-1).sum()) > 60
  est = FoundationPose(model_pts=sc['mesh'].vertices, model_normals=sc['mesh'].vertex_normals, mesh=sc['mesh'], scorer=scorer, refiner=refiner,
                       debug=2, debug_dir=str(tmp_path))
  est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=1)
  for name in ('vis_refiner.png', 'vis_score.png'):
    assert (tmp_path / name).read_bytes()[:8] == b'\x89PNG\r\n\x1a\n'
  extra = {}
  est.track_one(rgb=sc['rgb'], depth=sc['depth'], K=sc['K'], iteration=1, extra=extra)
  assert extra['vis'].dtype == np.uint8 and extra['vis'].shape == (row_h + 4, 2 * (row_w + 2) + 2, 3)      # one hypothesis: one row per half


def test_cv_draw_text_bitmap_labels():
  """cv_draw_text (src/Utils.py:630-653) without cv2: text of the given colour at the given corner, kept inside the image, one line per
  text line `line_spacing` heights apart; only the glyph shapes differ from the reference's Hershey strokes."""
  from foundationpose_amd import Utils as U
  from foundationpose_amd.vis import _text_mask, cv_draw_text
  assert U.cv_draw_text is cv_draw_text
  img = np.full((60, 260, 3), 7, np.uint8)
  out = cv_draw_text(img, text='id:3, score:99.125', uv_top_left=(10, 10), color=(0, 255, 0), fontScale=0.5)
  assert out is img
  on = (img == np.array([0, 255, 0], np.uint8)).all(-1)
  mask = _text_mask('id:3, score:99.125', 2)
  assert mask.shape == (14, 12 * 18) and int(on.sum()) == int(mask.sum())
  ys, xs = np.nonzero(on)
  assert ys.min() >= 10 and ys.max() < 24 and xs.min() >= 10
  assert (img[~on] == 7).all()
  # a corner outside the image is moved inside; two lines are 1.5 text heights apart
  img2 = np.zeros((80, 120, 3), np.float32)
  cv_draw_text(img2, text='ab\n12', uv_top_left=(-20, -5), color=(255, 255, 255), fontScale=0.5)
  ys, xs = np.nonzero(img2[..., 0] > 0)
  assert xs.min() >= 0 and ys.min() >= 0 and ys.max() < 80
  rows_used = np.unique(ys)
  assert rows_used.min() < 14 and rows_used.max() >= 21 and rows_used.max() < 21 + 14
  # a line wider than the image is moved left until its end is inside: it loses its beginning, as in the reference
  img4 = np.zeros((30, 100, 3), np.uint8)
  cv_draw_text(img4, text='id:3, score:99.125', uv_top_left=(10, 5), color=(9, 9, 9), fontScale=0.5)
  assert int((img4[..., 0] == 9).sum()) == int(mask[:, 216 - 99:].sum())
  # unknown characters draw a box instead of raising; an outline lies under the text
  img3 = np.zeros((40, 80, 3), np.uint8)
  cv_draw_text(img3, text='#', uv_top_left=(5, 5), color=(255, 0, 0), fontScale=0.25, outline_color=(0, 0, 255))
  assert (img3 == np.array([255, 0, 0], np.uint8)).all(-1).sum() == 20 and (img3 == np.array([0, 0, 255], np.uint8)).all(-1).sum() > 0

"""GPU parity: each HIP kernel, called through the C-ABI, against the CPU oracle on the same seeded
inputs (sizes the oracle finishes in seconds).  Tolerances are stated per test."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def sc():
  return util.scene(0)


@pytest.fixture(scope='module')
def fp():
  from foundationpose_amd import Utils, _lib
  return dict(U=Utils, L=_lib, ctx=_lib.Context.get('cuda:0'))


def test_crop_window_tf(sc, fp):
  from oracle import geometry as G
  poses = util.hypotheses(sc, 64, jitter_seed=3)
  for ratio in (1.2, 1.1):
    tf_o = G.compute_crop_window_tf_batch(torch.from_numpy(poses), sc['K'], ratio, (160, 160), sc['diameter'])
    tf_g = fp['U'].compute_crop_window_tf_batch(poses=poses, K=sc['K'], crop_ratio=ratio, out_size=(160, 160), method='box_3d',
                                                mesh_diameter=sc['diameter'])
    # same float32 op order on both sides -> bit-exact
    np.testing.assert_array_equal(tf_g.cpu().numpy(), tf_o.numpy())
  with pytest.raises(RuntimeError):
    fp['U'].compute_crop_window_tf_batch(poses=poses, K=sc['K'], method='min_box', mesh_diameter=0.1, out_size=(160, 160))


def _render_pair(sc, fp, poses, bbox, out, mt_cpu=None, use_light=True):
  from oracle.render import nvdiffrast_render as orender
  mt_cpu = mt_cpu or sc['mt']
  eo, eg = {}, {'rast': None}
  co, do, no = orender(K=sc['K'], H=480, W=640, ob_in_cams=poses, mesh_tensors=mt_cpu, bbox2d=bbox, output_size=out,
                       use_light=use_light, get_normal=True, extra=eo)
  cg, dg, ng = fp['U'].nvdiffrast_render(K=sc['K'], H=480, W=640, ob_in_cams=torch.from_numpy(poses).cuda(),
                                         mesh_tensors=util.to_dev(mt_cpu), bbox2d=None if bbox is None else bbox.cuda(),
                                         output_size=out, use_light=use_light, get_normal=True, extra=eg)
  _assert_same_coverage_and_faces(eo['rast'], eg['rast'].cpu())
  return (co, do, no, eo['xyz_map']), (cg.cpu(), dg.cpu(), ng.cpu(), eg['xyz_map'].cpu())


def _assert_same_coverage_and_faces(rast_o, rast_g, what=''):
  """dr.rasterize's output (u, v, z/w, triangle id + 1) of the oracle and of the HIP rasteriser: WHICH pixels are covered and WHICH face
  wins each of them is index work - 64-bit integer edge functions on vertices snapped to 1/16 px, nearest z/w, ties to the lower face
  id - on both sides from the same float32 vertex transform: compared exactly.  Only the interpolated floats carry an allowance."""
  id_o, id_g = rast_o[..., 3].numpy().astype(np.int64), rast_g[..., 3].numpy().astype(np.int64)
  n_cov = int(((id_o > 0) != (id_g > 0)).sum())
  n_face = int((id_o != id_g).sum())
  print(f'rasteriser {what}: {int((id_o > 0).sum())} covered pixels, coverage differs on {n_cov}, winning face on {n_face}')
  assert n_cov == 0 and n_face == 0, f'{what}: coverage differs on {n_cov} pixels, the winning face on {n_face}'
  d = (rast_o[..., :3] - rast_g[..., :3]).abs()
  assert float(d.max()) <= 2e-6, f'{what}: barycentrics / z/w differ by {float(d.max()):.2e}'


@pytest.mark.parametrize('textured', [False, True])
def test_render_crops_match_oracle(sc, fp, textured):
  """Rasteriser: coverage and the winning face are decided in integer arithmetic -> IDENTICAL pixel sets and face ids (asserted
  exactly, _assert_same_coverage_and_faces); interpolants are the same fmaf chains -> agree to float32 rounding (2e-6 abs on
  values <= 1; the allowance is for these floats only)."""
  from oracle import geometry as G
  s = util.scene(0, textured=textured) if textured else sc
  poses = util.hypotheses(s, 12, jitter_seed=5)
  tf = G.compute_crop_window_tf_batch(torch.from_numpy(poses), s['K'], 1.2, (160, 160), s['diameter'])
  bbox = G.crop_bbox2d_ori(tf, (160, 160))
  ref, got = _render_pair(s, fp, poses, bbox, (160, 160), mt_cpu=s['mt'])
  cov_o, cov_g = ref[1] > 0, got[1] > 0
  assert torch.equal(cov_o, cov_g)                   # (coverage and the winning face: compared exactly inside _render_pair)
  assert float(cov_o.float().mean()) > 0.15          # the object fills a good part of the crop
  for name, a, b in zip(('color', 'depth', 'normal', 'xyz'), ref, got):
    frac, mx, med = util.mismatch_report(a.numpy(), b.numpy(), 2e-6)
    assert frac <= 2e-4, f'{name}: {frac:.2e} of values differ by > 2e-6 (max {mx:.2e})'


@pytest.mark.parametrize('variant', ['light_dir', 'light_pos', 'light_color', 'projection_mat'])
def test_render_light_and_projection_arguments(sc, fp, variant):
  """The non-default arguments of nvdiffrast_render (src/Utils.py:159-161: projection_mat; :200-211: light_dir, light_pos with
  light_dir=None, light_color) against the oracle's restatement of those lines, same tolerance as the default path."""
  from oracle import geometry as G
  from oracle.render import nvdiffrast_render as orender
  from oracle.geometry import projection_matrix_from_intrinsics
  s = util.scene(0, textured=True) if variant == 'light_color' else sc
  poses = util.hypotheses(s, 6, jitter_seed=11)
  tf = G.compute_crop_window_tf_batch(torch.from_numpy(poses), s['K'], 1.2, (160, 160), s['diameter'])
  bbox = G.crop_bbox2d_ori(tf, (160, 160))
  kw = dict(use_light=True)
  if variant == 'light_dir':
    kw.update(light_dir=np.array([0.3, -0.5, 0.8]))
  elif variant == 'light_pos':
    kw.update(light_dir=None, light_pos=np.array([0.4, -0.3, 0.1]))
  elif variant == 'light_color':
    kw.update(light_dir=np.array([-0.2, 0.1, 1.0]), light_color=np.array([1.0, 0.6, 0.2]), w_ambient=0.6, w_diffuse=0.7)
  else:                                 # another near / far pair and a sheared K than the ones the kernel derives itself
    K2 = s['K'].copy()
    K2[0, 1] = 3.0
    kw.update(projection_mat=projection_matrix_from_intrinsics(K2, height=480, width=640, znear=0.05, zfar=20.0))
  eo, eg = {}, {'rast': None}
  co, do, no = orender(K=s['K'], H=480, W=640, ob_in_cams=poses, mesh_tensors=s['mt'], bbox2d=bbox, output_size=(160, 160), get_normal=True,
                       extra=eo, **kw)
  cg, dg, ng = fp['U'].nvdiffrast_render(K=s['K'], H=480, W=640, ob_in_cams=torch.from_numpy(poses).cuda(), mesh_tensors=util.to_dev(s['mt']),
                                         bbox2d=bbox.cuda(), output_size=(160, 160), get_normal=True, extra=eg, **kw)
  assert float((do > 0).float().mean()) > 0.15
  _assert_same_coverage_and_faces(eo['rast'], eg['rast'].cpu(), variant)
  assert torch.equal(do > 0, dg.cpu() > 0)
  for name, a, b in (('color', co, cg), ('depth', do, dg), ('normal', no, ng), ('xyz', eo['xyz_map'], eg['xyz_map'])):
    frac, mx, _ = util.mismatch_report(a.numpy(), b.cpu().numpy(), 2e-6)
    assert frac <= 2e-4, f'{variant} {name}: {frac:.2e} of values differ by > 2e-6 (max {mx:.2e})'
  # the argument does something: the image differs from the default-light render
  c0, _, _ = orender(K=s['K'], H=480, W=640, ob_in_cams=poses, mesh_tensors=s['mt'], bbox2d=bbox, output_size=(160, 160), use_light=True)
  assert float((c0 - co).abs().max()) > 1e-2 or variant == 'projection_mat'


def test_render_full_frame_and_edge_cases(sc, fp):
  poses = util.hypotheses(sc, 2)
  poses[1, :3, 3] = [5.0, 5.0, 1.0]      # entirely outside the frustum -> empty image
  ref, got = _render_pair(sc, fp, poses, None, (480, 640))
  assert float((ref[1] > 0).float().sum()) > 1000
  for a, b in zip(ref, got):
    frac, mx, _ = util.mismatch_report(a.numpy(), b.numpy(), 2e-6)
    assert frac <= 2e-4
  assert float(got[1][1].abs().max()) == 0.0
  # object behind the camera: w <= 0 culls everything
  poses[0, 2, 3] = -0.5
  _, got = _render_pair(sc, fp, poses[:1], None, (64, 64))
  assert float(got[1].abs().max()) == 0.0
  # object THROUGH the camera plane (its far half in front of the camera, the rest behind): triangles with a vertex at w <= 0 are
  # rasterised in homogeneous coordinates (nvdiffrast clips them), same arithmetic on both sides, full frame and one crop window
  for tz, out, bbox in ((0.02, (120, 160), None), (0.05, (160, 160), torch.tensor([[200.0, 150.0, 440.0, 390.0]]))):
    poses[0, :3, 3] = [0.01, -0.02, tz]
    ref, got = _render_pair(sc, fp, poses[:1], bbox, out)
    assert float((ref[1] > 0).float().mean()) > 0.2          # the near part of the object fills much of the view
    for a, b in zip(ref, got):
      frac, mx, _ = util.mismatch_report(a.numpy(), b.numpy(), 5e-6)
      assert frac <= 5e-4, (tz, frac, mx)
  # zero hypotheses
  c, d, n = fp['U'].nvdiffrast_render(K=sc['K'], H=480, W=640, ob_in_cams=torch.zeros((0, 4, 4)).cuda(), mesh_tensors=util.to_dev(sc['mt']),
                                      output_size=(160, 160))
  assert c.shape == (0, 160, 160, 3)
  with pytest.raises(NotImplementedError):
    fp['U'].nvdiffrast_render(K=sc['K'], H=480, W=640, ob_in_cams=torch.from_numpy(poses).cuda(), mesh_tensors=util.to_dev(sc['mt']),
                              context='metal', glctx=None)


@pytest.mark.parametrize('textured', [False, True])
def test_render_solo_equals_three_launches(sc, fp, textured):
  """One or two hypotheses render in ONE launch (raster.hip render_kernel<.., true>: vertex pass, per-strip classification into LDS and
  triangle pass together - a tracking frame's rasteriser); larger batches in three.  Same functions on the same inputs and an
  order-independent z-buffer: every output of a hypothesis rendered alone or in a pair equals its slice of a batch of 8, bit for bit -
  the API's maps, dr.rasterize's own output and the fused fp16 network tensor; also for a pose through the camera plane."""
  from oracle import geometry as G
  from foundationpose_amd._lib import check, k_ptr, lib, ptr, stream_ptr
  if textured:
    sc = util.scene(0, textured=True)
  poses = util.hypotheses(sc, 8, jitter_seed=3)
  poses[5, :3, 3] = [0.01, -0.02, 0.05]              # through the camera plane: homogeneous rasterisation, list B only
  tf = G.compute_crop_window_tf_batch(torch.from_numpy(poses), sc['K'], 1.2, (160, 160), sc['diameter'])
  bbox = G.crop_bbox2d_ori(tf, (160, 160)).cuda().contiguous()
  bbox[5] = torch.tensor([200.0, 150.0, 440.0, 390.0])
  mt = util.to_dev(sc['mt'])
  dposes = torch.from_numpy(poses).cuda()
  ctx, dm = fp['ctx'], fp['L'].device_mesh(fp['ctx'], mt)
  Kd, Kp = k_ptr(sc['K'])

  def api(sel):
    e = {'rast': None}
    c, d, n = fp['U'].nvdiffrast_render(K=sc['K'], H=480, W=640, ob_in_cams=dposes[sel], mesh_tensors=mt, bbox2d=bbox[sel], output_size=(160, 160),
                                        get_normal=True, use_light=True, extra=e)
    return c, d, n, e['xyz_map'], e['rast']

  def net(sel):
    p, bb = dposes[sel].contiguous(), bbox[sel].contiguous()
    out = torch.zeros((len(p), 160, 160, 8), dtype=torch.float16, device='cuda')
    check(lib().fp_render_net(ctx.handle, dm.handle, ptr(p), len(p), Kp, 480, 640, ptr(bb), 160, 160, sc['diameter'], 1, 0.001, ptr(out), stream_ptr()))
    return (out,)
  for fn in (api, net):
    whole = fn(slice(0, 8))
    assert float((whole[0] != 0).float().mean()) > 0.05
    for sel in (slice(0, 1), slice(5, 6), slice(6, 8), slice(4, 6)):
      part = fn(sel)
      for k, (w, q) in enumerate(zip(whole, part)):
        assert torch.equal(w[sel], q), f'{fn.__name__} output {k} of hypotheses {sel} differs between the one-launch and the three-launch form'


def _net_tensor_to_planar(t, n):
  """fp16 NHWC8 net tensor -> (n,6,160,160) float32"""
  return t.reshape(n, 160, 160, 8)[..., :6].permute(0, 3, 1, 2).float().cpu()


@pytest.mark.parametrize('which', ['refine', 'score'])
def test_fused_crop_tensors_match_oracle(sc, fp, which):
  """A (render) and B (observed crop) network inputs, fused kernels vs the oracle's
  make_crop_data_batch.  fp16 storage: |err| <= 2^-11 relative (values <= 2) -> atol 1.5e-3; nearest-
  sampled channels may pick a neighbouring source pixel when the coordinate is within 1e-5 of .5
  (allowance 5e-4 of the pixels)."""
  from oracle import predict as OP
  from foundationpose_amd._lib import check, k_ptr, lib, ptr, stream_ptr
  L, ctx = fp['L'], fp['ctx']
  n = 8
  poses = util.hypotheses(sc, n, jitter_seed=7)
  from oracle import geometry as G
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  rgb_t = torch.as_tensor(sc['rgb'], dtype=torch.float32)
  if which == 'refine':
    cfg = dict(OP.DEFAULT_REFINE_CFG)
    xyz_map = torch.from_numpy(G.depth2xyzmap(depth, sc['K']))
    pd = OP.make_crop_data_batch_refine(cfg, poses, sc['mt'], rgb_t, torch.from_numpy(depth), sc['K'], xyz_map, sc['diameter'])
    geom, mode, thres = xyz_map.cuda().contiguous(), 0, 0.001
  else:
    cfg = dict(OP.DEFAULT_SCORE_CFG)
    pd = OP.make_crop_data_batch_score(cfg, poses, sc['mt'], rgb_t, torch.from_numpy(depth), sc['K'], sc['diameter'])
    geom, mode, thres = torch.from_numpy(depth).cuda().contiguous(), 1, 0.1
  A_ref = torch.cat([pd['rgbAs'], pd['xyz_mapAs']], 1)
  B_ref = torch.cat([pd['rgbBs'], pd['xyz_mapBs']], 1)
  dposes = torch.from_numpy(poses).cuda()
  tf = torch.empty((n, 3, 3), device='cuda')
  bbox = torch.empty((n, 4), device='cuda')
  Kd, Kp = k_ptr(sc['K'])
  s = stream_ptr()
  check(lib().fp_crop_window_tf(ctx.handle, ptr(dposes), n, Kp, cfg['crop_ratio'], sc['diameter'], 160, 160, ptr(tf), ptr(bbox), s))
  np.testing.assert_array_equal(tf.cpu().numpy(), pd['tf_to_crops'].numpy())
  net = torch.zeros((2 * n, 160, 160, 8), dtype=torch.float16, device='cuda')
  dm = L.device_mesh(ctx, util.to_dev(sc['mt']))
  check(lib().fp_render_net(ctx.handle, dm.handle, ptr(dposes), n, Kp, 480, 640, ptr(bbox), 160, 160, sc['diameter'], 1, thres, ptr(net), s))
  rgb_d = rgb_t.cuda().contiguous()
  check(lib().fp_crop_observed(ctx.handle, ptr(rgb_d), ptr(geom), 480, 640, Kp, ptr(tf), ptr(dposes), n, 160, 160, mode, sc['diameter'], 1, 1,
                               ptr(net[n:]), s))
  Bf32 = torch.empty((n, 6, 160, 160), device='cuda')
  check(lib().fp_crop_observed(ctx.handle, ptr(rgb_d), ptr(geom), 480, 640, Kp, ptr(tf), ptr(dposes), n, 160, 160, mode, sc['diameter'], 1, 0,
                               ptr(Bf32), s))
  torch.cuda.synchronize()
  A_g, B_g = _net_tensor_to_planar(net[:n], n), _net_tensor_to_planar(net[n:], n)
  assert float(net[..., 6:].abs().max()) == 0.0
  fa, ma, _ = util.mismatch_report(A_ref.numpy(), A_g.numpy(), 1.5e-3)
  assert fa <= 2e-4, f'A: {fa:.2e} mismatching (max {ma:.3f})'
  # B, float32 output: rgb (bilinear) tight, xyz (nearest) with the boundary allowance
  f_rgb, m_rgb, _ = util.mismatch_report(B_ref[:, :3].numpy(), Bf32[:, :3].cpu().numpy(), 2e-5)
  assert f_rgb <= 1e-4, f'rgbB: {f_rgb:.2e} (max {m_rgb:.2e})'
  # nearest lookups: oracle and kernel share the tie rule of oracle/warp.py:round_half_even_snapped
  lo = 0
  f_xyz, m_xyz, _ = util.mismatch_report(B_ref[:, 3:, lo:, lo:].numpy(), Bf32[:, 3:, lo:, lo:].cpu().numpy(), 1e-5)
  assert f_xyz <= 5e-4, f'xyzB: {f_xyz:.2e} (max {m_xyz:.2e})'
  fb, mb, _ = util.mismatch_report(B_ref[:, :, lo:, lo:].numpy(), B_g[:, :, lo:, lo:].numpy(), 1.5e-3)
  assert fb <= 5e-4
  assert float((B_ref[:, 3:] != 0).float().mean()) > 0.05     # the observed object is inside the crops


def test_use_normal_branch(sc, fp):
  """cfg['use_normal']=True (predict_pose_refine.py:49,58,74-76; VERDICT r4 missing item 2).  The refiner's batch then carries
  normalAs (rendered camera-frame normals, warped by tf_to_crops once more as the reference does) and normalBs (the frame's
  normal map cropped, nearest); RefineNet still reads rgb + xyz (:186-187), so predict() returns the SAME poses as without the
  flag; a missing normal_map raises like the reference's torch.as_tensor(None); without the flag a normal_map is ignored (:162-163);
  the scorer's flag changes nothing (predict_score.py:103-104)."""
  from oracle import geometry as G, predict as OP
  from oracle import warp as OW
  from foundationpose_amd import synthetic as S
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
  from foundationpose_amd.predict_score import ScorePredictor
  n = 6
  poses = util.hypotheses(sc, n, jitter_seed=11)
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  xyz_map = G.depth2xyzmap(depth, sc['K'])
  rng = np.random.default_rng(5)
  normal_map = rng.standard_normal((480, 640, 3)).astype(np.float32)
  normal_map /= np.linalg.norm(normal_map, axis=-1, keepdims=True)
  normal_map[depth < 0.001] = 0
  mt = util.to_dev(sc['mt'])
  cfg = dict(OP.DEFAULT_REFINE_CFG, use_normal=True)
  rgb_t = torch.as_tensor(sc['rgb'], dtype=torch.float32)
  ref = OP.make_crop_data_batch_refine(cfg, poses, sc['mt'], rgb_t, torch.from_numpy(depth), sc['K'], torch.from_numpy(xyz_map), sc['diameter'],
                                       normal_map=normal_map)
  pd = make_crop_data_batch((160, 160), poses, None, sc['rgb'], depth, sc['K'], cfg['crop_ratio'], xyz_map, normal_map=normal_map,
                            mesh_diameter=sc['diameter'], cfg=cfg, mesh_tensors=mt)
  assert pd.normalAs.shape == pd.normalBs.shape == (n, 3, 160, 160)
  # normalBs: copies of source pixels under the shared tie rule -> equal; normalAs: rendered floats (2e-6) picked by the same rule,
  # with the rasteriser tests' allowance for pixels on a triangle edge
  assert float((pd.normalBs.cpu() != ref['normalBs']).float().mean()) <= 5e-4
  assert float((ref['normalBs'] != 0).float().mean()) > 0.3
  frac, mx, _ = util.mismatch_report(ref['normalAs'].numpy(), pd.normalAs.cpu().numpy(), 2e-5)
  assert frac <= 5e-4, f'normalAs: {frac:.2e} (max {mx:.2e})'
  # (the reference's second warp samples the 160x160 crop at FRAME coordinates of the window, so most of normalAs is out of bounds = 0:
  # reproduced, not repaired; the rendered normals themselves are checked below and in the rasteriser tests)
  cov = (ref['normalAs'] != 0).any(1)
  assert float(cov.float().mean()) < 0.9
  _, _, nr = fp['U'].nvdiffrast_render(K=sc['K'], H=480, W=640, ob_in_cams=torch.from_numpy(poses).cuda(), mesh_tensors=mt, get_normal=True,
                                       output_size=(160, 160), bbox2d=G.crop_bbox2d_ori(pd.tf_to_crops.cpu(), (160, 160)))
  hit = (nr != 0).any(-1)
  assert 0.02 < float(hit.float().mean()) < 0.9 and float((torch.linalg.norm(nr, dim=-1)[hit] - 1).abs().max()) < 1e-5      # F.normalize'd (src/Utils.py:196)
  want_a = OW.warp_perspective_nearest(nr.cpu().permute(0, 3, 1, 2).contiguous(), pd.tf_to_crops.cpu(), (160, 160))
  frac, mx, _ = util.mismatch_report(want_a.numpy(), pd.normalAs.cpu().numpy(), 0.0)      # the second warp alone, on the HIP render
  assert frac <= 5e-4, f'normalAs (second warp): {frac:.2e} (max {mx:.2e})'
  for k in ('rgbAs', 'xyz_mapAs', 'rgbBs', 'xyz_mapBs'):          # the other fields as without the flag
    frac, mx, _ = util.mismatch_report(ref[k].numpy(), getattr(pd, k).cpu().numpy(), 1.5e-3)
    assert frac <= 5e-4, f'{k}: {frac:.2e} (max {mx:.3f})'
  # the generic warp itself, on a source batch (not broadcast) with a non-square source: against the oracle's kornia restatement
  src = torch.from_numpy(rng.standard_normal((n, 37, 53, 2)).astype(np.float32))
  tf = torch.eye(3).repeat(n, 1, 1)                                # x_dst = sx x_src + tx: part of every output falls outside the source
  tf[:, 0, 0] = torch.linspace(0.7, 1.3, n); tf[:, 1, 1] = torch.linspace(0.9, 0.6, n)
  tf[:, 0, 2] = torch.linspace(-6.0, 4.0, n); tf[:, 1, 2] = torch.linspace(3.0, -5.0, n)
  want = OW.warp_perspective_nearest(src.permute(0, 3, 1, 2).contiguous(), tf, (24, 40))
  got = torch.empty((n, 2, 24, 40), device='cuda')
  tfd, srcd = tf.cuda().contiguous(), src.cuda().contiguous()
  check(lib().fp_warp_nearest(fp['ctx'].handle, ptr(srcd), n, 37, 53, 2, ptr(tfd), n, 24, 40, ptr(got), stream_ptr()))
  assert torch.equal(got.cpu(), want) and 0.3 < float((want != 0).float().mean()) < 0.98
  with pytest.raises(RuntimeError, match='source batch'):
    check(lib().fp_warp_nearest(fp['ctx'].handle, ptr(srcd), 2, 37, 53, 2, ptr(tfd), n, 24, 40, ptr(got), stream_ptr()))
  # predict(): same poses with and without the flag; the reference's failure without a normal map; ignored without the flag
  rsd = S.make_refine_state_dict(0)
  plain = PoseRefinePredictor(state_dict=rsd, cfg=REFINE_DEFAULT)
  withn = PoseRefinePredictor(state_dict=rsd, cfg=dict(REFINE_DEFAULT, use_normal=True))
  kw = dict(rgb=sc['rgb'], depth=depth, K=sc['K'], ob_in_cams=poses, xyz_map=xyz_map, mesh_tensors=mt, mesh_diameter=sc['diameter'], iteration=2)
  p0, _ = plain.predict(**kw)
  p1, _ = withn.predict(normal_map=normal_map, **kw)
  p2, _ = plain.predict(normal_map=normal_map, **kw)
  assert torch.equal(p0, p1) and torch.equal(p0, p2) and float((p0.cpu() - torch.from_numpy(poses)).abs().max()) > 1e-4
  with pytest.raises(RuntimeError, match='NoneType'):
    withn.predict(**kw)
  _, vis = withn.predict(normal_map=normal_map, get_vis=True, **kw)
  assert vis is not None
  ssd = S.make_score_state_dict(1)
  skw = dict(rgb=sc['rgb'], depth=depth, K=sc['K'], ob_in_cams=p0, mesh_tensors=mt, mesh_diameter=sc['diameter'])
  s0, _ = ScorePredictor(state_dict=ssd, cfg=SCORE_DEFAULT).predict(**skw)
  s1, _ = ScorePredictor(state_dict=ssd, cfg=dict(SCORE_DEFAULT, use_normal=True)).predict(normal_map=normal_map, **skw)
  assert torch.equal(torch.as_tensor(s0), torch.as_tensor(s1))


@pytest.mark.parametrize('which', ['refine', 'score'])
def test_make_crop_data_batch_api(sc, fp, which):
  """The reference's intermediate API (predict_pose_refine.py:24-89, predict_score.py:56-114): make_crop_data_batch ->
  BatchPoseData.  Planar fields are views of the fp16 net tensor -> atol 1.5e-3 with the nearest-sampling allowance of
  the test above; tf_to_crops, depthBs bit-exact (same float32 op order / a copy of source pixels); depthAs to float32
  rounding.  Then: the step-by-step loop through this API is the fused C call, bit for bit."""
  from oracle import geometry as G, predict as OP
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  n = 8
  poses = util.hypotheses(sc, n, jitter_seed=7)
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  rgb_t = torch.as_tensor(sc['rgb'], dtype=torch.float32)
  mt = util.to_dev(sc['mt'])
  if which == 'refine':
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
    cfg = dict(OP.DEFAULT_REFINE_CFG)
    xyz_map = G.depth2xyzmap(depth, sc['K'])
    ref = OP.make_crop_data_batch_refine(cfg, poses, sc['mt'], rgb_t, torch.from_numpy(depth), sc['K'], torch.from_numpy(xyz_map), sc['diameter'])
    pd = make_crop_data_batch((160, 160), poses, None, sc['rgb'], depth, sc['K'], cfg['crop_ratio'], xyz_map, mesh_diameter=sc['diameter'],
                              cfg=cfg, mesh_tensors=mt)
  else:
    from foundationpose_amd.predict_score import ScorePredictor, make_crop_data_batch
    cfg = dict(OP.DEFAULT_SCORE_CFG)
    ref = OP.make_crop_data_batch_score(cfg, poses, sc['mt'], rgb_t, torch.from_numpy(depth), sc['K'], sc['diameter'])
    pd = make_crop_data_batch((160, 160), poses, None, sc['rgb'], depth, sc['K'], cfg['crop_ratio'], mesh_diameter=sc['diameter'],
                              cfg=cfg, mesh_tensors=mt)
  assert len(pd) == n and pd.net_input.shape == (2 * n, 160, 160, 8) and pd.normalAs is None and pd.poseB is None
  np.testing.assert_array_equal(pd.tf_to_crops.cpu().numpy(), ref['tf_to_crops'].numpy())
  np.testing.assert_array_equal(pd.poseA.cpu().numpy(), poses)
  np.testing.assert_array_equal(pd.Ks.cpu().numpy(), np.broadcast_to(sc['K'].astype(np.float32), (n, 3, 3)))
  assert float((pd.mesh_diameters - sc['diameter']).abs().max()) < 1e-7
  for k, allow in (('rgbAs', 2e-4), ('xyz_mapAs', 2e-4), ('rgbBs', 1e-4), ('xyz_mapBs', 5e-4)):
    frac, mx, _ = util.mismatch_report(ref[k].numpy(), getattr(pd, k).cpu().numpy(), 1.5e-3)
    assert frac <= allow, f'{k}: {frac:.2e} (max {mx:.3f})'
  if which == 'score':
    frac, mx, _ = util.mismatch_report(ref['depthAs'].numpy(), pd.depthAs.cpu().numpy(), 2e-6)
    assert frac <= 2e-4, f'depthAs: {frac:.2e} (max {mx:.2e})'
    assert float((pd.depthBs.cpu() != ref['depthBs']).float().mean()) <= 5e-4
    scorer = ScorePredictor(state_dict=S.make_score_state_dict(1), cfg=SCORE_DEFAULT)
    f_step = scorer.forward_features(pd)
    f_fused = scorer.extract_features(sc['rgb'], depth, sc['K'], poses, mesh_tensors=mt, mesh_diameter=sc['diameter'])
    assert torch.equal(f_step, f_fused)
    sub = pd.select_by_indices(torch.tensor([5, 2, 0, 7, 3]))
    assert torch.equal(scorer.forward_features(sub), f_step[[5, 2, 0, 7, 3]])
    # batches of 1 .. 4 hypotheses run the 3x3 layers in their split-K form (another fp32 summation order): equal to rounding
    sub2 = pd.select_by_indices(torch.tensor([5, 2]))
    torch.testing.assert_close(scorer.forward_features(sub2), f_step[[5, 2]], rtol=0, atol=2e-4 * float(f_step.abs().max()))
  else:
    refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT)
    cur = torch.from_numpy(poses).cuda()
    for _ in range(2):
      pd = make_crop_data_batch((160, 160), cur, None, sc['rgb'], depth, sc['K'], refiner.cfg['crop_ratio'], xyz_map,
                                mesh_diameter=sc['diameter'], cfg=refiner.cfg, mesh_tensors=mt)
      out = refiner.forward(pd)
      cur = refiner.update_poses(pd.poseA, out['trans'], out['rot'], sc['diameter'])
    fused, _ = refiner.predict(sc['rgb'], depth, sc['K'], poses, xyz_map, mesh_tensors=mt, mesh_diameter=sc['diameter'], iteration=2)
    assert torch.equal(cur, fused)
    assert torch.equal(out['trans'], refiner.last_trans_update) and torch.equal(out['rot'], refiner.last_rot_update)


def test_depth_filters(sc, fp):
  from oracle import geometry as G
  U = fp['U']
  d = sc['depth'].copy()
  d[100:104, 200:260] = 150.0          # beyond zfar
  e_o = G.erode_depth(d, radius=2)
  e_g = U.erode_depth(d, radius=2, device='cuda')
  assert isinstance(e_g, np.ndarray)
  np.testing.assert_array_equal(e_g, e_o)
  b_o = G.bilateral_filter_depth(e_o, radius=2)
  b_g = U.bilateral_filter_depth(e_o, radius=2, device='cuda')
  np.testing.assert_allclose(b_g, b_o, atol=2e-6, rtol=0)      # expf vs np.exp: <= 2 ulp on O(1) metres
  # tensor in -> tensor out (src/Utils.py:353-355,393-394)
  assert torch.is_tensor(U.erode_depth(torch.from_numpy(d).cuda(), radius=2))
  xb_o = G.depth2xyzmap_batch(torch.from_numpy(b_o)[None], torch.as_tensor(sc['K'], dtype=torch.float32)[None], zfar=np.inf)
  xb_g = U.depth2xyzmap_batch(torch.from_numpy(b_o)[None].cuda(), torch.as_tensor(sc['K'], dtype=torch.float32)[None], zfar=np.inf)
  np.testing.assert_allclose(xb_g.cpu().numpy(), xb_o.numpy(), atol=1e-6)
  # all-invalid and tiny images
  z = np.zeros((5, 7), np.float32)
  assert U.erode_depth(z, radius=2).max() == 0 and U.bilateral_filter_depth(z, radius=2).max() == 0


@pytest.mark.parametrize('hw', [(480, 640), (61, 83), (5, 7), (8, 32), (9, 33)])
def test_depth_prefilter_equals_the_three_kernels_chained(sc, fp, hw):
  """Utils.depth_prefilter (one launch, LDS tiles: the prelude of a tracking frame) against erode_depth -> bilateral_filter_depth ->
  depth2xyzmap_batch on the device: bit-identical depth and xyz map, on images with holes, values beyond zfar, borders that cut
  the 32x8 tiles, and images smaller than one tile."""
  U = fp['U']
  H, W = hw
  rs = np.random.RandomState(H * 1000 + W)
  d = sc['depth'][:H, :W].copy() if (H, W) != (480, 640) else sc['depth'].copy()
  d = np.ascontiguousarray(d) + (rs.uniform(-1, 1, d.shape) * np.where(np.arange(W) < W // 2, 0.002, 0.01)[None]).astype(np.float32)   # (noise: erosion and the 1 cm gate both bite)
  d[rs.uniform(size=d.shape) < 0.05] = 0.0
  d[rs.uniform(size=d.shape) < 0.02] = 150.0
  d[:, W // 2] = np.where(rs.uniform(size=H) < 0.5, 0.0005, d[:, W // 2])
  K32 = np.asarray(sc['K'], dtype=np.float32)
  dt = torch.from_numpy(d).cuda()
  chain_d = U.bilateral_filter_depth(U.erode_depth(dt, radius=2, device='cuda'), radius=2, device='cuda')
  chain_x = U.depth2xyzmap_batch(chain_d[None], K32[None], zfar=np.inf)[0]
  fused_d, fused_x = U.depth_prefilter(dt, sc['K'], radius=2)
  assert fused_d.shape == (H, W) and fused_x.shape == (H, W, 3)
  assert torch.equal(fused_d, chain_d), int((fused_d != chain_d).sum())
  assert torch.equal(fused_x, chain_x)
  rgb = torch.from_numpy(rs.randint(0, 256, (H, W, 3)).astype(np.uint8)).cuda()
  d3, x3, rgb_f = U.depth_prefilter(dt, sc['K'], radius=2, rgb_u8=rgb)             # the frame's colours in the same launch
  assert torch.equal(d3, chain_d) and torch.equal(x3, chain_x) and torch.equal(rgb_f, rgb.to(torch.float))
  if H * W > 1000:
    assert float((fused_d > 0).float().mean()) > 0.2 and float((fused_d == 0).float().mean()) > 0.02      # (both outcomes of the erosion occur)
  with pytest.raises(RuntimeError, match='radius'):
    U.depth_prefilter(dt, sc['K'], radius=3)


def test_pose_update(fp):
  from oracle import predict as OP
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  rs = np.random.RandomState(3)
  n = 37
  A = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
  from foundationpose_amd.synthetic import random_rotation
  for i in range(n):
    A[i, :3, :3] = random_rotation(rs)
  A[:, :3, 3] = rs.randn(n, 3) * 0.3
  trans = (rs.randn(n, 3) * 0.5).astype(np.float32)
  trans[0] = 0
  for rot_dim, rep in ((3, 'axis_angle'), (6, '6d')):
    rot = (rs.randn(n, rot_dim) * 0.7).astype(np.float32)
    if rot_dim == 3:
      rot[0] = 0          # exercises the eps clamp of so3_exp_map
    for norm_xyz in (True, False):
      cfg = dict(OP.DEFAULT_REFINE_CFG, rot_rep=rep, normalize_xyz=norm_xyz)
      ref, _, _ = OP.pose_update(cfg, torch.from_numpy(A), torch.from_numpy(trans), torch.from_numpy(rot), 0.191)
      out = torch.empty((n, 4, 4), device='cuda')
      tn = np.asarray(cfg['trans_normalizer'], dtype=np.float32)
      A_d, t_d, r_d = torch.from_numpy(A).cuda(), torch.from_numpy(trans).cuda(), torch.from_numpy(rot).cuda()
      check(lib().fp_pose_update(fp['ctx'].handle, ptr(A_d), ptr(t_d), ptr(r_d), n, rot_dim, 0 if norm_xyz else 1, ptr(tn), cfg['rot_normalizer'],
                                 np.float32(0.191 / 2) if norm_xyz else 1.0, ptr(out), stream_ptr()))
      np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-6)
  # trans_rep='deepim' (predict_pose_refine.py:201-215): shift of the projected centre inside the crop + depth ratio.  The oracle
  # inverts tf_to_crops / K with torch.inverse, the kernel by the adjugate: equal to a few float32 roundings of 0.2 .. 1.5 m
  from oracle import geometry as G
  K = np.array([[1066.778, 0.7, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])
  A[:, :3, 3] = np.c_[rs.uniform(-0.15, 0.15, n), rs.uniform(-0.1, 0.1, n), rs.uniform(0.4, 1.5, n)]
  tf = G.compute_crop_window_tf_batch(torch.from_numpy(A), K, 1.2, (160, 160), 0.191)
  trans = np.c_[rs.randn(n, 2) * 0.05, 1 + rs.randn(n) * 0.03].astype(np.float32)
  rot = (rs.randn(n, 3) * 0.7).astype(np.float32)
  for norm_xyz in (False, True):
    cfg = dict(OP.DEFAULT_REFINE_CFG, trans_rep='deepim', normalize_xyz=norm_xyz)
    ref, td, _ = OP.pose_update(cfg, torch.from_numpy(A), torch.from_numpy(trans), torch.from_numpy(rot), 0.191, tf_to_crops=tf, Ks=K)
    assert float(td.abs().max()) > 1e-3
    out = torch.empty((n, 4, 4), device='cuda')
    Kd = np.ascontiguousarray(K, dtype=np.float64)
    A_d, t_d, r_d, tf_d = torch.from_numpy(A).cuda(), torch.from_numpy(trans).cuda(), torch.from_numpy(rot).cuda(), tf.reshape(n, 9).contiguous().cuda()
    check(lib().fp_pose_update_deepim(fp['ctx'].handle, ptr(A_d), ptr(t_d), ptr(r_d), n, 3, ptr(tf_d), Kd.ctypes.data, 160.0,
                                      cfg['rot_normalizer'], np.float32(0.191 / 2) if norm_xyz else 1.0, ptr(out), stream_ptr()))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=5e-6)


def _pack_conv_weight(w, cin_pad):
  cout, cin, k, _ = w.shape
  kraw = k * k * cin_pad
  kpad = (kraw + 31) // 32 * 32
  p = torch.zeros((cout, kpad), dtype=torch.float16)
  wp = torch.zeros((cout, k, k, cin_pad))
  wp[..., :cin] = w.permute(0, 2, 3, 1)
  p[:, :kraw] = wp.reshape(cout, -1).half()
  return p


@pytest.mark.parametrize('shape', [
  # (N, H, W, Cin, Cout, k, stride, residual, relu)
  (3, 40, 40, 128, 128, 3, 1, True, True),
  (2, 40, 40, 256, 256, 3, 1, False, True),
  (2, 80, 80, 64, 128, 3, 2, False, True),
  (3, 20, 20, 512, 512, 3, 1, True, True),
  (2, 40, 40, 256, 512, 3, 2, False, True),
  (328, 40, 40, 128, 128, 3, 1, True, True),        # 1025 tiles: four full rounds + one tile (quarter tiles for the remainder)
  (9, 40, 40, 256, 512, 3, 2, False, True),         # band-in-LDS stride-2 kernel (conv_s2.hip): 8-row tiles straddle images, last tile partial
  (9, 40, 40, 64, 128, 3, 2, False, False),         # the same with 128-cout blocks, no ReLU
  (2, 160, 160, 6, 64, 7, 2, False, True),
  (1, 20, 20, 512, 512, 3, 1, True, True),          # one hypothesis: conv_small.hip (13 x 16 workgroups; FP_SMALL=0: split-K, 8 shares of 2 chunks + finishing pass)
  (1, 40, 40, 256, 256, 3, 1, False, True),         # split-K, 4 shares of 2 chunks
  (2, 40, 40, 128, 128, 3, 1, True, False),         # 26 quarter tiles of a 128-channel layer: too few chunks to split, one launch of 128-pixel tiles, no ReLU
  (2, 40, 40, 256, 256, 3, 1, True, False),         # split-K, 4 shares of 2 chunks, residual, no ReLU
  (1, 40, 40, 128, 128, 3, 1, True, True),          # conv_small.hip (a few images: 32 x 32 tiles, K split over the waves): 50 x 4 workgroups
  (3, 20, 20, 512, 512, 3, 1, True, False),         # ... tiles that straddle images, last tile partial (1200 pixels), no ReLU
  (1, 40, 40, 256, 128, 3, 1, False, True),         # ... Cout != Cin
  (1, 80, 80, 64, 128, 3, 2, False, True),          # ... the 64 -> 128 stride-2 layer behind the stem: four waves of 16 channels, 305-pixel band
  (2, 80, 80, 64, 128, 3, 2, True, False),          # ... tiles that straddle rows and the two images
  (1, 40, 40, 256, 512, 3, 2, False, True),         # stride 2, 72 K-steps at one hypothesis: split-K of the implicit GEMM (4 shares of 18 steps), last 64-pixel tile partial
  (4, 40, 40, 256, 512, 3, 2, False, False),        # the same at the largest batch that takes it (1600 pixels), no ReLU
  (1, 1, 1000, 512, 1024, 1, 1, False, False),     # a Linear layer (1x1, M=1000 tokens: ragged last tile)
  (1, 1, 130, 512, 64, 1, 1, False, False),
])
def test_conv_igemm_vs_fp32_reference(fp, shape):
  """MFMA implicit GEMM vs torch fp32 conv2d on the same fp16-rounded operands; fp32 accumulate ->
  the only error is accumulation order + the fp16 output rounding: rtol 2e-3 of the output scale."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  N, H, W, Cin, Cout, k, stride, use_res, relu = shape
  g = torch.Generator().manual_seed(sum(shape))
  x = torch.randn((N, Cin, H, W), generator=g).half()
  w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).half()
  b = torch.randn((Cout,), generator=g) * 0.1
  pad = (k - 1) // 2
  ref = torch.nn.functional.conv2d(x.float(), w.float(), b, stride=stride, padding=pad)
  cin_pad = 8 if Cin < 8 else Cin
  xin = torch.zeros((N, H, W, cin_pad), dtype=torch.float16)
  xin[..., :Cin] = x.permute(0, 2, 3, 1)
  res = None
  if use_res:
    res = torch.randn(ref.shape, generator=g).half()
    ref = ref + res.float()
    res_d = res.permute(0, 2, 3, 1).contiguous().cuda()
  if relu:
    ref = torch.relu(ref)
  wp = _pack_conv_weight(w.float(), cin_pad).cuda()
  x_d, b_d = xin.cuda(), b.cuda()
  Ho, Wo = ref.shape[-2:]
  for out_f32 in (0, 1):
    out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else torch.float16, device='cuda')
    check(lib().fp_conv2d_f16(fp['ctx'].handle, ptr(x_d), N, H, W, cin_pad, ptr(wp), ptr(b_d), Cout, k, k, stride, pad,
                              ptr(res_d) if use_res else None, 1 if relu else 0, ptr(out), out_f32, stream_ptr()))
    got = out.float().permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= (2e-3 if not out_f32 else 2e-4) * scale + 1e-5, f'out_f32={out_f32}: err {err:.3e} scale {scale:.2f}'


@pytest.mark.parametrize('N,C,use_res,relu', [(20, 128, False, True), (20, 128, True, True), (37, 128, True, False), (504, 128, True, True),
                                              (18, 256, False, True), (21, 256, True, True), (252, 256, True, True)])
def test_band_kernel_equals_halo_kernel(fp, N, C, use_res, relu):
  """The band-in-LDS form of the 128 -> 128 and 256 -> 256 layers on 40x40 maps (conv_s1b.hip, what the networks run for > 40
  hypotheses) accumulates every output element in the order of the general 3x3 stride-1 kernel (32-channel groups, kernel rows, taps, two
  16-channel steps) from the same bias and rounds once behind the same fp32 residual add: BIT-identical outputs, and within the fp32
  reference's tolerance."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  g = torch.Generator(device='cuda').manual_seed(1000 + N + C)
  x = torch.randn((N, 40, 40, C), device='cuda', generator=g).half()
  x[:, :, :, ::3] = x[:, :, :, ::3].relu()                       # some exact zeros, as behind a ReLU
  w = (torch.randn((C, C, 3, 3), device='cuda', generator=g) * (2.0 / (C * 9)) ** 0.5).half()
  b = torch.randn((C,), device='cuda', generator=g) * 0.1
  res = torch.randn((N, 40, 40, C), device='cuda', generator=g).half() if use_res else None
  wp = _pack_conv_weight(w.float().cpu(), C).cuda()
  outs = []
  for band in (0, 1):
    out = torch.full((N, 40, 40, C), float('nan'), dtype=torch.float16, device='cuda')
    if band:
      check(lib().fp_conv3x3_band_f16(fp['ctx'].handle, ptr(x), N, C, ptr(wp), ptr(b), ptr(res) if use_res else None, 1 if relu else 0, ptr(out), stream_ptr()))
    else:
      check(lib().fp_conv2d_f16(fp['ctx'].handle, ptr(x), N, 40, 40, C, ptr(wp), ptr(b), C, 3, 3, 1, 1, ptr(res) if use_res else None,
                                1 if relu else 0, ptr(out), 0, stream_ptr()))
    outs.append(out)
  torch.cuda.synchronize()
  assert not bool(torch.isnan(outs[1]).any())
  assert torch.equal(outs[0], outs[1])
  if N <= 37:
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1)
    if use_res:
      ref = ref + res.float().permute(0, 3, 1, 2)
    if relu:
      ref = torch.relu(ref)
    err = float((outs[1].float().permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= 2e-3 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize('shape', [(3, 40, 128, 128, False, True), (5, 40, 256, 256, True, True), (7, 20, 512, 512, True, True), (2, 20, 512, 512, False, False),
                                   (9, 40, 128, 256, True, True), (33, 20, 512, 512, True, True), (13, 40, 256, 256, False, True)])
def test_conv_winograd_rows_vs_fp32_reference(fp, shape):
  """conv_wino.hip (F(2,3) along rows, 2/3 of the matrix work) against torch's fp32 conv2d on the same fp16 activations and fp32 weights.
  The transformed weights u = G g and the transformed inputs v = B^T d are fp16, so the error is NOT only accumulation order: tolerance
  3e-3 of the output scale (the direct kernel: 2e-3 on fp16-rounded weights) - and the emulation of exactly these roundings in plain torch
  (tests/tools/winograd_precision.wino_conv3x3_rows) has to agree to summation order + the output rounding.  Batches of 2 .. 33 images:
  tiles that start anywhere in a row, cross image boundaries (20x20: up to three images per 512-pixel tile) and end past the tensor."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  from tests.tools.winograd_precision import wino_conv3x3_rows
  N, HW, Cin, Cout, use_res, relu = shape
  g = torch.Generator().manual_seed(sum(int(v) for v in shape))
  x = torch.randn((N, Cin, HW, HW), generator=g).half().relu()
  w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5
  b = torch.randn((Cout,), generator=g) * 0.1
  ref = torch.nn.functional.conv2d(x.float(), w, b, padding=1)
  emu = wino_conv3x3_rows(x.float(), w, b)
  res = None
  if use_res:
    res = torch.randn(ref.shape, generator=g).half()
    ref, emu = ref + res.float(), emu + res.float()
  if relu:
    ref, emu = torch.relu(ref), torch.relu(emu)
  x_d = x.permute(0, 2, 3, 1).contiguous().cuda()
  res_d = res.permute(0, 2, 3, 1).contiguous().cuda() if use_res else None
  b_d = b.cuda()
  out = torch.full((N, HW, HW, Cout), float('nan'), dtype=torch.float16, device='cuda')
  wc = w.contiguous()
  check(lib().fp_conv3x3_wino_f16(fp['ctx'].handle, ptr(x_d), N, HW, Cin, Cout, wc.data_ptr(), ptr(b_d), ptr(res_d) if use_res else None,
                                  1 if relu else 0, ptr(out), stream_ptr()))
  got = out.float().permute(0, 3, 1, 2).cpu()
  assert not bool(torch.isnan(got).any())
  scale = float(ref.abs().max())
  e_ref, e_emu = float((got - ref).abs().max()), float((got - emu.half().float()).abs().max())
  print(f'{shape}: |wino - fp32| {e_ref:.2e}, |wino - torch emulation of its roundings| {e_emu:.2e} (output scale {scale:.2f})')
  assert e_ref <= 3e-3 * scale + 1e-5
  assert e_emu <= 1.2e-3 * scale + 1e-5          # one fp16 ulp of the largest outputs: summation order only


@pytest.mark.parametrize('C,HW,sizes', [(512, 20, (8, 32, 50, 70, 63)), (256, 40, (4, 12, 63, 20, 32)), (128, 40, (8, 40, 24, 63, 64))])
def test_halo_kernel_output_does_not_depend_on_the_tile_size(fp, C, HW, sizes):
  """The 3x3 stride-1 kernel cuts what is less than a round of 512-pixel tiles into tiles of 1 .. 4 x 128 pixels, whichever finishes first
  (halo_plan, conv_halo.hip): batches of different sizes therefore run different tile shapes.  A pixel's arithmetic must not depend on
  the tile it is in: the images two batches share come out BIT-identical, and agree with the fp32 reference.  (Sizes: tiles of 1, 2, 3 and 4
  x 128 pixels; none of the 1 .. 4-hypothesis launches that run split-K - another accumulation order, its own size class.)"""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  g = torch.Generator(device='cuda').manual_seed(77 + C)
  nmax = max(sizes)
  x = torch.randn((nmax, HW, HW, C), device='cuda', generator=g).half().relu()
  w = (torch.randn((C, C, 3, 3), device='cuda', generator=g) * (2.0 / (C * 9)) ** 0.5).half()
  b = torch.randn((C,), device='cuda', generator=g) * 0.1
  res = torch.randn((nmax, HW, HW, C), device='cuda', generator=g).half()
  wp = _pack_conv_weight(w.float().cpu(), C).cuda()
  outs = {}
  for n in sizes:
    out = torch.full((n, HW, HW, C), float('nan'), dtype=torch.float16, device='cuda')
    check(lib().fp_conv2d_f16(fp['ctx'].handle, ptr(x), n, HW, HW, C, ptr(wp), ptr(b), C, 3, 3, 1, 1, ptr(res), 1, ptr(out), 0, stream_ptr()))
    outs[n] = out
  torch.cuda.synchronize()
  nmin = min(sizes)
  for n in sizes:
    assert not bool(torch.isnan(outs[n]).any())
    assert torch.equal(outs[n][:nmin], outs[nmin]), f'batch of {n} against batch of {nmin}'
    m = min(n, sorted(sizes)[1])
    assert torch.equal(outs[n][:m], outs[sorted(sizes)[1]][:m])
  ref = torch.relu(torch.nn.functional.conv2d(x[:nmin].float().permute(0, 3, 1, 2), w.float(), b, padding=1) + res[:nmin].float().permute(0, 3, 1, 2))
  err = float((outs[nmin].float().permute(0, 3, 1, 2) - ref).abs().max())
  assert err <= 2e-3 * float(ref.abs().max()) + 1e-5


def test_conv_rejects_bad_shapes(fp):
  from foundationpose_amd._lib import FoundationPoseAmdError, check, lib, ptr, stream_ptr
  x = torch.zeros((1, 4, 4, 48), dtype=torch.float16, device='cuda')
  w = torch.zeros((64, 448), dtype=torch.float16, device='cuda')
  b = torch.zeros((64,), device='cuda')
  o = torch.zeros((1, 4, 4, 64), dtype=torch.float16, device='cuda')
  with pytest.raises(FoundationPoseAmdError):
    check(lib().fp_conv2d_f16(fp['ctx'].handle, ptr(x), 1, 4, 4, 48, ptr(w), ptr(b), 64, 3, 3, 1, 1, None, 1, ptr(o), 0, stream_ptr()))


def test_same_size_meshes_are_not_mistaken_for_each_other(sc):
  """The uploaded mesh is cached per mesh_tensors OBJECTS (identity + in-place version), not per device address: two
  different meshes of equal size rendered back to back through `mesh=` (temporary mesh_tensors whose device addresses the
  allocator re-uses), a re-coloured copy, and an in-place edit all render their own geometry / colours."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.Utils import make_mesh_tensors, nvdiffrast_render
  pose = torch.as_tensor(sc['gt_pose'])[None].cuda()
  kw = dict(K=sc['K'], H=480, W=640, ob_in_cams=pose, use_light=True)
  mesh_a = S.make_mustard_mesh(seed=0)
  mesh_b = S.make_mustard_mesh(seed=0)
  mesh_b.vertices = mesh_b.vertices * np.array([0.6, 1.3, 0.8])        # same V / F, other shape
  mesh_c = S.make_mustard_mesh(seed=7)                                   # same shape, other colours
  ref = {}
  for name, m in (('a', mesh_a), ('b', mesh_b), ('c', mesh_c)):
    mt = make_mesh_tensors(m)
    col, dep, _ = nvdiffrast_render(mesh_tensors=mt, **kw)
    ref[name] = (col.clone(), dep.clone())
    del mt
  assert not torch.equal(ref['a'][1], ref['b'][1]) and not torch.equal(ref['a'][0], ref['c'][0]) and torch.equal(ref['a'][1], ref['c'][1])
  for _ in range(2):                                                     # temporaries built inside the call, alternating
    for name, m in (('a', mesh_a), ('b', mesh_b), ('c', mesh_c)):
      col, dep, _ = nvdiffrast_render(mesh=m, **kw)
      assert torch.equal(col, ref[name][0]) and torch.equal(dep, ref[name][1]), name
  mt = make_mesh_tensors(mesh_a)
  col0, _, _ = nvdiffrast_render(mesh_tensors=mt, **kw)
  mt['vertex_color'].mul_(0.5)                                           # in-place edit of a cached dict
  col1, _, _ = nvdiffrast_render(mesh_tensors=mt, **kw)
  assert torch.equal(col0, ref['a'][0]) and not torch.equal(col1, col0)
  mt2 = dict(mt, pos=mt['pos'] * 1.1)                                    # shares every tensor but `pos`
  _, dep2, _ = nvdiffrast_render(mesh_tensors=mt2, **kw)
  assert not torch.equal(dep2, ref['a'][1])


@pytest.mark.parametrize('n_hyp', [3, 1, 7])
def test_token_linear_epilogues_vs_fp32_reference(fp, n_hyp):
  """csrc/tok_gemm.hip (every nn.Linear of the transformer heads) against torch fp32 on the same fp16-rounded operands:
  rows (+ReLU), the transposed V image, residual + LayerNorm rows, and the LayerNorm sums over groups of 16 tokens.
  M = 400 n: 1200 and 2800 tokens end in a ragged 128-token tile, 400 leaves the last tile 1/8 full.  The GEMM accumulates
  in fp32, so the error is accumulation order + one fp16 rounding of the output; the LayerNorm statistics stay in fp32."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  M = 400 * n_hyp
  g = torch.Generator().manual_seed(100 + n_hyp)
  x = torch.randn((M, 512), generator=g).half()
  w = (torch.randn((512, 512), generator=g) * (1.0 / 512) ** 0.5).half().float()
  b = torch.randn((512,), generator=g) * 0.1
  res = (torch.randn((M, 512), generator=g) * 2 + 0.5).half()
  gam, bet = torch.rand((512,), generator=g) + 0.5, torch.randn((512,), generator=g) * 0.1
  lin = x.float() @ w.T + b
  x_d, res_d = x.cuda(), res.cuda()

  def run(epi, relu, out, use_res=False, ln=False):
    check(lib().fp_token_linear_f16(fp['ctx'].handle, ptr(x_d), M, ptr(w.numpy()), ptr(b.numpy()), epi, relu, ptr(res_d) if use_res else None,
                                    ptr(gam.numpy()) if ln else None, ptr(bet.numpy()) if ln else None, 400, ptr(out), stream_ptr()))
    return out
  scale = float(lin.abs().max())
  for relu in (0, 1):
    out = run(0, relu, torch.full((M, 512), float('nan'), dtype=torch.float16, device='cuda'))
    ref = torch.relu(lin) if relu else lin
    assert float((out.float().cpu() - ref).abs().max()) <= 2e-3 * scale
  # transposed V image: channel c of token t of hypothesis b at [b][c >> 7][c & 127][vt_col(t)], zero pad columns
  vt = run(1, 0, torch.full((n_hyp, 4, 128, 416), float('nan'), dtype=torch.float16, device='cuda')).float().cpu()
  t = np.arange(400)
  col = (t & ~15) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 2) | (t & 3)
  got_v = vt[..., torch.from_numpy(col)].permute(0, 3, 1, 2).reshape(M, 512)
  assert float((got_v - lin).abs().max()) <= 2e-3 * scale
  assert float(vt[..., 400:].abs().max()) == 0
  # residual + LayerNorm (fp32 statistics on the fp32 sum)
  y = lin + res.float()
  ln_ref = torch.nn.functional.layer_norm(y, (512,), gam, bet, 1e-5)
  out = run(2, 0, torch.full((M, 512), float('nan'), dtype=torch.float16, device='cuda'), use_res=True, ln=True)
  err = float((out.float().cpu() - ln_ref).abs().max())
  assert err <= 2.5e-3 * float(ln_ref.abs().max()), err
  # sums of the normalised rows (no gamma / beta) over groups of 16 tokens: fp32 end to end after the MFMA
  nrm = torch.nn.functional.layer_norm(y, (512,), None, None, 1e-5)
  gs_ref = nrm.reshape(M // 16, 16, 512).sum(1)
  gs = run(3, 0, torch.full((M // 16, 512), float('nan'), dtype=torch.float32, device='cuda'), use_res=True)
  assert float((gs.cpu() - gs_ref).abs().max()) <= 2e-4 * float(gs_ref.abs().max()) + 1e-4
  # a hypothesis' results do not depend on its place in the batch (tile alignment): bit-exact
  if n_hyp > 1:
    x1 = x[400:800].contiguous().cuda()
    r1 = res[400:800].contiguous().cuda()
    g1 = torch.empty((25, 512), dtype=torch.float32, device='cuda')
    check(lib().fp_token_linear_f16(fp['ctx'].handle, ptr(x1), 400, ptr(w.numpy()), ptr(b.numpy()), 3, 0, ptr(r1), None, None, 400, ptr(g1), stream_ptr()))
    assert torch.equal(g1, gs[25:50])


@pytest.mark.parametrize('n_hyp', [1, 3, 7, 40])
def test_token_qkv_equals_the_64_token_kernel(fp, n_hyp):
  """csrc/tok_qkv.hip - the in-projections of the transformer heads over a resident 128-token tile, all column blocks in one launch -
  against csrc/tok_gemm.hip's 64-token kernel: fp16 rows (+ReLU) and the transposed V image, BIT-identical (bias first, k ascending in
  steps of 16, one rounding), and within the fp32 reference's tolerance.  M = 400 n: ragged last tiles; 40 hypotheses = 125 tiles."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  M = 400 * n_hyp
  g = torch.Generator().manual_seed(500 + n_hyp)
  x = torch.randn((M, 512), generator=g).half()
  w = (torch.randn((512, 512), generator=g) * (1.0 / 512) ** 0.5).half().float()
  b = torch.randn((512,), generator=g) * 0.1
  x_d = x.cuda()

  def run(epi, relu, out):
    check(lib().fp_token_linear_f16(fp['ctx'].handle, ptr(x_d), M, ptr(w.numpy()), ptr(b.numpy()), epi, relu, None, None, None, 400, ptr(out), stream_ptr()))
    return out
  for relu in (0, 1):
    a = run(0, relu, torch.full((M, 512), float('nan'), dtype=torch.float16, device='cuda'))
    c = run(4, relu, torch.full((M, 512), float('nan'), dtype=torch.float16, device='cuda'))
    assert not bool(torch.isnan(c).any()) and torch.equal(a, c)
  va = run(1, 0, torch.full((n_hyp, 4, 128, 416), float('nan'), dtype=torch.float16, device='cuda'))
  vc = run(5, 0, torch.full((n_hyp, 4, 128, 416), float('nan'), dtype=torch.float16, device='cuda'))
  assert not bool(torch.isnan(vc).any()) and torch.equal(va, vc)
  lin = x.float() @ w.T + b
  assert float((c.float().cpu() - torch.relu(lin)).abs().max()) <= 2e-3 * float(lin.abs().max())


@pytest.mark.parametrize('n_hyp', [1, 2])
def test_token_qkv_few_image_form(fp, n_hyp):
  """tok_qkv_small_kernel - the in-projections of a tracking frame (one or two hypotheses): 32-token tiles x 128 columns per workgroup, K in
  four quarters added in order.  Another fp32 summation order than tok_gemm.hip's, so not its bit pattern: within one fp16 ulp of it on
  nearly every element, within the fp32 reference's tolerance everywhere; the transposed V image has the same layout (token order, zero
  pad behind the last tokens) - every element, NaN-prefilled buffers."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  M = 400 * n_hyp
  g = torch.Generator().manual_seed(700 + n_hyp)
  x = torch.randn((M, 512), generator=g).half()
  w = (torch.randn((512, 512), generator=g) * (1.0 / 512) ** 0.5).half().float()
  b = torch.randn((512,), generator=g) * 0.1
  x_d = x.cuda()

  def run(epi, relu, out):
    check(lib().fp_token_linear_f16(fp['ctx'].handle, ptr(x_d), M, ptr(w.numpy()), ptr(b.numpy()), epi, relu, None, None, None, 400, ptr(out), stream_ptr()))
    return out
  lin = x.float() @ w.T + b
  scale = float(lin.abs().max())
  for relu in (0, 1):
    a = run(0, relu, torch.full((M, 512), float('nan'), dtype=torch.float16, device='cuda'))
    c = run(6, relu, torch.full((M, 512), float('nan'), dtype=torch.float16, device='cuda'))
    assert not bool(torch.isnan(c).any())
    ref = torch.relu(lin) if relu else lin
    assert float((c.float().cpu() - ref).abs().max()) <= 2e-3 * scale
    assert float((a != c).float().mean()) < 0.05 and float((a.float() - c.float()).abs().max()) <= 4e-3 * scale      # (the same values to an fp16 ulp)
  va = run(1, 0, torch.full((n_hyp, 4, 128, 416), float('nan'), dtype=torch.float16, device='cuda'))
  vc = run(7, 0, torch.full((n_hyp, 4, 128, 416), float('nan'), dtype=torch.float16, device='cuda'))
  assert not bool(torch.isnan(vc).any())
  assert torch.equal(vc[..., 400:], torch.zeros_like(vc[..., 400:])) and torch.equal(va[..., 400:], vc[..., 400:])
  assert float((va != vc).float().mean()) < 0.05 and float((va.float() - vc.float()).abs().max()) <= 4e-3 * scale


@pytest.mark.parametrize('B', [3, 1, 2])
@pytest.mark.parametrize('T', [400, 384, 230, 64, 37, 1])
def test_attention_vs_reference(fp, T, B):
  """Fused MHA core vs softmax(QK^T/sqrt(128))V in fp32 on the same fp16 operands.  P is rounded to
  fp16 before the PV MFMA: |err| <= 2^-11 * sum|p v| -> atol 2e-3 on O(1) values.  T = 400 is the networks' token
  count (tail key block 16/64 full); 384 and 64 end on a block boundary (no tail), 230 needs two query blocks with a
  partly empty second one, 37 and 1 are single partial blocks.  B = 1 and 2 run the split-K form of a tracking frame (attention_small_kernel:
  a key block per wave, partial softmaxes merged through LDS), B = 3 the flash-style kernel of the batches."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  g = torch.Generator().manual_seed(5)
  qk = (torch.randn((B * T, 1024), generator=g) * 1.5).half()
  v = torch.randn((B * T, 512), generator=g).half()
  vt = torch.zeros((B, 4, 128, 416), dtype=torch.float16)
  t = np.arange(T)
  col = (t & ~15) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 2) | (t & 3)       # the image's token order (foundationpose_amd.h)
  vt[..., torch.from_numpy(col)] = v.reshape(B, T, 4, 128).permute(0, 2, 3, 1)
  out = torch.empty((B * T, 512), dtype=torch.float16, device='cuda')
  qk_d, vt_d = qk.cuda(), vt.cuda()
  check(lib().fp_attention_f16(fp['ctx'].handle, ptr(qk_d), ptr(vt_d), B, T, ptr(out), stream_ptr()))
  q = qk[:, :512].float().reshape(B, T, 4, 128).transpose(1, 2)
  k = qk[:, 512:].float().reshape(B, T, 4, 128).transpose(1, 2)
  vv = v.float().reshape(B, T, 4, 128).transpose(1, 2)
  ref = (torch.softmax(q @ k.transpose(-1, -2) / 128 ** 0.5, -1) @ vv).transpose(1, 2).reshape(B * T, 512)
  err = float((out.float().cpu() - ref).abs().max())
  assert err <= 2e-3, err


@pytest.mark.parametrize('B,T', [(130, 400), (70, 400), (300, 37), (140, 230), (270, 64), (90, 130), (33, 400)])
def test_attention_persistent_items_equal_single_items(fp, B, T):
  """More (hypothesis, head, query block) items than CUs: a workgroup walks several items and its DMA ring, Q^T staging and store
  accounting run across the item boundaries.  The arithmetic of an item does not depend on that, so the batch must equal launches
  of at most 256 items (one item per workgroup: no boundary inside a workgroup) BIT FOR BIT, and the fp32 reference within the
  kernel's tolerance.  (300, 37): one key block per item; (140, 230): second query block partly empty (waves without queries issue no
  stores); (270, 64): no key tail; (90, 130): three key blocks; (33, 400): 264 items - only 8 workgroups get a second one."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  g = torch.Generator(device='cuda').manual_seed(100 + B + T)
  qk = (torch.randn((B * T, 1024), device='cuda', generator=g) * 1.5).half()
  vt = torch.zeros((B, 4, 128, 416), device='cuda', dtype=torch.float16)
  t = np.arange(T)
  col = (t & ~15) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 2) | (t & 3)
  v = torch.randn((B, 4, 128, T), device='cuda', generator=g).half()
  vt[..., torch.from_numpy(col).cuda()] = v
  out = torch.full((B * T, 512), float('nan'), dtype=torch.float16, device='cuda')
  check(lib().fp_attention_f16(fp['ctx'].handle, ptr(qk), ptr(vt), B, T, ptr(out), stream_ptr()))
  nqb = -(-T // 224)
  step = max(3, 256 // (4 * nqb))
  while 0 < B % step <= 2:                # (launches of one or two hypotheses take the split-K form of a tracking frame: not this kernel)
    step -= 1
  ref16 = torch.full_like(out, float('nan'))
  for b0 in range(0, B, step):
    nb = min(step, B - b0)
    check(lib().fp_attention_f16(fp['ctx'].handle, ptr(qk[b0 * T:]), ptr(vt[b0:]), nb, T, ptr(ref16[b0 * T:]), stream_ptr()))
  torch.cuda.synchronize()
  assert not bool(torch.isnan(out).any())
  assert torch.equal(out, ref16)
  q = qk[:, :512].float().reshape(B, T, 4, 128).transpose(1, 2)
  k = qk[:, 512:].float().reshape(B, T, 4, 128).transpose(1, 2)
  vv = v.float().permute(0, 1, 3, 2)
  ref = (torch.softmax(q @ k.transpose(-1, -2) / 128 ** 0.5, -1) @ vv).transpose(1, 2).reshape(B * T, 512)
  assert float((out.float() - ref).abs().max()) <= 2e-3


def test_attention_run_to_run_identical(fp):
  """Regression: an inline-asm v_max3 that was the first reader of the S accumulators (no hazard wait states inside asm)
  read half-written MFMA results - within tolerance (softmax is shift-invariant) but different from run to run."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  g = torch.Generator(device='cuda').manual_seed(9)
  B, T = 24, 400
  qk = (torch.randn((B * T, 1024), device='cuda', generator=g) * 1.5).half()
  vt = torch.zeros((B, 4, 128, 416), device='cuda', dtype=torch.float16)
  vt[..., :T] = torch.randn((B, 4, 128, T), device='cuda', generator=g).half()
  outs = []
  for _ in range(4):
    out = torch.full((B * T, 512), float('nan'), dtype=torch.float16, device='cuda')
    check(lib().fp_attention_f16(fp['ctx'].handle, ptr(qk), ptr(vt), B, T, ptr(out), stream_ptr()))
    torch.cuda.synchronize()
    outs.append(out)
  assert not bool(torch.isnan(outs[0]).any())
  for o in outs[1:]:
    assert torch.equal(o, outs[0])


def test_register_prelude_on_device(sc, fp, golden):
  """back-projection and guess_translation's reductions without a host copy of the depth image.
  Pinned by the REFERENCE's own outputs (tests/golden: depth2xyzmap, FoundationPose.guess_translation run unmodified)."""
  from foundationpose_amd import Utils as U
  from foundationpose_amd.estimater import FoundationPose
  from oracle import geometry as G
  K = sc['K']
  # (1) golden inputs of the reference (12 x 16 image with 20 % dropout, rectangular mask)
  depth, mask = golden['d2x_depth'], golden['gt_mask']
  d_dev = torch.from_numpy(depth).cuda()
  xyz = U.depth2xyzmap(d_dev, K)
  assert xyz.is_cuda and np.array_equal(xyz.cpu().numpy(), golden['d2x_xyz'])            # bit-exact (float64 math, one rounding)
  xyz_np = U.depth2xyzmap(depth, K)                                                      # numpy in -> numpy out, same kernel
  assert isinstance(xyz_np, np.ndarray) and xyz_np.dtype == np.float32 and np.array_equal(xyz_np, golden['d2x_xyz'])
  uvs = np.array([[3, 2], [10, 7]])
  only = U.depth2xyzmap(depth, K, uvs=uvs)
  assert np.array_equal(only[2, 3], golden['d2x_xyz'][2, 3]) and np.array_equal(only[7, 10], golden['d2x_xyz'][7, 10])
  assert np.count_nonzero(only.any(-1)) <= 2
  dummy = type('E', (), {})()
  t = FoundationPose.guess_translation(dummy, depth=d_dev, mask=mask, K=K)
  np.testing.assert_allclose(t, golden['gt_center'], rtol=0, atol=1e-12)
  t0 = FoundationPose.guess_translation(dummy, depth=d_dev, mask=np.zeros_like(mask), K=K)
  assert np.array_equal(t0, golden['gt_center_empty'])
  # (2) full frame: every statistic against numpy, odd and even counts of usable pixels (median = mean of two middles)
  dfull = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  for drop in (0, 1):
    m = sc['mask'].copy()
    if drop:
      r, c = np.argwhere(m & (dfull >= 0.001))[0]
      m[r, c] = False
    st = U.mask_depth_stats(torch.from_numpy(dfull).cuda(), m)
    rows, cols = np.nonzero(m)
    usable = m & (dfull >= 0.001)
    assert (st['cmin'], st['cmax'], st['rmin'], st['rmax']) == (cols.min(), cols.max(), rows.min(), rows.max())
    assert st['n_mask'] == m.sum() and st['n_usable'] == usable.sum()
    assert st['median'] == np.median(dfull[usable])                                        # exact
  np.testing.assert_array_equal(U.depth2xyzmap(torch.from_numpy(dfull).cuda(), K).cpu().numpy(), G.depth2xyzmap(dfull, K))
  # (3) no usable pixel: mask entirely on dropped depth
  z = np.zeros_like(dfull)
  st = U.mask_depth_stats(torch.from_numpy(z).cuda(), sc['mask'])
  assert st['n_usable'] == 0 and st['n_mask'] == sc['mask'].sum() and st['median'] == 0


@pytest.mark.parametrize('n_hyp', [3, 1, 7, 40])
def test_head_mlp_vs_fp32_reference(fp, n_hyp):
  """csrc/head_mlp.hip - out-projection + LayerNorm1 + linear1 + ReLU + linear2 + LayerNorm2 sums of one RefineNet head in one launch
  (refine_network.py:56-70,88-91) - against torch fp32 on the same fp16-rounded operands, with x1 and ff rounded to fp16 where the
  kernel rounds them (they are GEMM operands: fp16 in the reference's autocast too).  M = 400 n: 1200 and 2800 tokens end in a ragged
  128-token tile, 16 000 are 125 workgroups.  Also: a hypothesis' result does not depend on its place in the batch (bit-exact), and the fused launch agrees with
  the three unfused tok_gemm launches it replaces to fp32 summation order."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  M = 400 * n_hyp
  g = torch.Generator().manual_seed(300 + n_hyp)
  att = torch.randn((M, 512), generator=g).half()
  tok = (torch.randn((M, 512), generator=g) * 2 + 0.5).half()
  mk = lambda: (torch.randn((512, 512), generator=g) * (1.0 / 512) ** 0.5).half().float()
  w_out, w1, w2 = mk(), mk(), mk()
  b_out, b1, b2 = (torch.randn((512,), generator=g) * 0.1 for _ in range(3))
  gam, bet = torch.rand((512,), generator=g) + 0.5, torch.randn((512,), generator=g) * 0.1
  x1 = torch.nn.functional.layer_norm(tok.float() + att.float() @ w_out.T + b_out, (512,), gam, bet, 1e-5).half().float()
  ff = torch.relu(x1 @ w1.T + b1).half().float()
  y = torch.nn.functional.layer_norm(x1 + ff @ w2.T + b2, (512,), None, None, 1e-5)
  ref = y.reshape(M // 16, 16, 512).sum(1)
  att_d, tok_d = att.cuda(), tok.cuda()
  args = [ptr(t.numpy()) for t in (w_out, b_out, gam, bet, w1, b1, w2, b2)]

  def run(a_d, t_d, m):
    out = torch.full((m // 16, 512), float('nan'), dtype=torch.float32, device='cuda')
    check(lib().fp_head_mlp_f16(fp['ctx'].handle, ptr(a_d), ptr(t_d), m, *args, ptr(out), stream_ptr()))
    return out
  got = run(att_d, tok_d, M)
  # x1 / ff are rounded to fp16 on both sides, but a value that sits on a rounding boundary may round differently (the fp32 sums
  # differ in their last bits): one fp16 ulp of one operand moves a 512-term dot product by ~1e-3 of an O(1) normalised value
  err = float((got.cpu() - ref).abs().max())
  print(f'head_mlp M={M}: max |gsum - ref| = {err:.2e} on values of magnitude {float(ref.abs().max()):.1f}')
  assert err <= 2.5e-3 * float(ref.abs().max()) + 1e-3, err
  # a hypothesis' result does not depend on its place in the batch: bit-exact within a size class (1 .. 4 hypotheses run the 64-token form,
  # larger batches the 128-token form: csrc/head_mlp.hip)
  if n_hyp == 3:
    one = run(att[400:800].contiguous().cuda(), tok[400:800].contiguous().cuda(), 400)
    assert torch.equal(one, got[25:50])
  elif n_hyp > 5:
    part = run(att[400:2400].contiguous().cuda(), tok[400:2400].contiguous().cuda(), 2000)        # 5 hypotheses: tiles of 128 tokens cut elsewhere
    assert torch.equal(part, got[25:150])
  # the three launches it replaces (tok_gemm.hip): same values up to the fp32 summation order of the LayerNorm statistics
  def lin(x_d, w, b, epi, relu, res_d, ln, out):
    check(lib().fp_token_linear_f16(fp['ctx'].handle, ptr(x_d), M, ptr(w.numpy()), ptr(b.numpy()), epi, relu, ptr(res_d) if res_d is not None else None,
                                    ptr(gam.numpy()) if ln else None, ptr(bet.numpy()) if ln else None, 400, ptr(out), stream_ptr()))
    return out
  x1_d = lin(att_d, w_out, b_out, 2, 0, tok_d, True, torch.empty((M, 512), dtype=torch.float16, device='cuda'))
  ff_d = lin(x1_d, w1, b1, 0, 1, None, False, torch.empty((M, 512), dtype=torch.float16, device='cuda'))
  gs_d = lin(ff_d, w2, b2, 3, 0, x1_d, False, torch.empty((M // 16, 512), dtype=torch.float32, device='cuda'))
  d = float((gs_d - got).abs().max())
  print(f'head_mlp M={M}: max |fused - three launches| = {d:.2e}')
  assert d <= 2.5e-3 * float(ref.abs().max()) + 1e-3


def test_profiling_busy_time_is_the_union_of_launch_spans(fp):
  """fp_prof_read / fp_prof_read_busy (bench.py's roofline figure): per kernel class the sum of the launch spans and the time with at
  least one launch executing.  A RefineNet pass on 8 hypotheses: the convolutions follow one another on one stream (busy == sum of
  spans); the two heads run on two streams, so the `linear` class must report busy <= sum of spans, and both > 0."""
  from foundationpose_amd import _lib, synthetic as S
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  ctx = fp['ctx']
  net = _lib.DeviceNet(ctx, _lib.FP_NET_REFINE, S.make_refine_state_dict(0), True)
  x = (torch.rand((16, 160, 160, 8), device='cuda') - 0.5).half()
  trans, rot = torch.empty((8, 3), device='cuda'), torch.empty((8, 3), device='cuda')
  run = lambda: check(lib().fp_refine_forward(ctx.handle, net.handle, ptr(x), 8, ptr(trans), ptr(rot), stream_ptr()))
  run()
  torch.cuda.synchronize()
  ctx.prof_reset()
  ctx.prof_enable(2)
  for _ in range(3):
    run()
  torch.cuda.synchronize()
  ctx.prof_enable(False)
  conv, lin = ctx.prof_read('conv3x3_halo'), ctx.prof_read('linear')
  ctx.prof_reset()
  assert conv['launches'] == 3 * 12 and lin['launches'] == 3 * 3          # (q | k | v of both heads: one launch; one fused MLP per head)
  assert conv['busy_ms'] > 0 and abs(conv['busy_ms'] - conv['total_ms']) <= 0.02 * conv['total_ms'] + 0.01
  assert 0 < lin['busy_ms'] <= lin['total_ms'] * 1.001 + 0.001


def test_render_mesh_too_large_for_lds_and_wide_output(fp):
  """Rasteriser paths the 8k-vertex bench mesh does not take: (a) a mesh of more than 8192 vertices, whose A records do not fit
  beside the strip in LDS (the triangle pass gathers them from global memory instead); (b) an output wider than a crop (320 x 240: the
  strip count follows the LDS budget).  Both against the oracle, same tolerance as test_render_crops_match_oracle."""
  from oracle import geometry as G
  s = util.scene(0, n_theta=128, n_z=110)
  assert s['mt']['pos'].shape[0] > 8192
  poses = util.hypotheses(s, 5, jitter_seed=9)
  tf = G.compute_crop_window_tf_batch(torch.from_numpy(poses), s['K'], 1.2, (160, 160), s['diameter'])
  bbox = G.crop_bbox2d_ori(tf, (160, 160))
  for out, bb in (((160, 160), bbox), ((240, 320), None)):
    ref, got = _render_pair(s, fp, poses if bb is not None else poses[:2], bb if bb is not None else None, out, mt_cpu=s['mt'])
    cov_o, cov_g = ref[1] > 0, got[1] > 0
    assert float((cov_o != cov_g).float().mean()) <= 1e-4 and float(cov_o.float().mean()) > 0.005
    for name, a, b in zip(('color', 'depth', 'normal', 'xyz'), ref, got):
      frac, mx, med = util.mismatch_report(a.numpy(), b.numpy(), 2e-6)
      assert frac <= 2e-4, f'{out} {name}: {frac:.2e} of values differ by > 2e-6 (max {mx:.2e})'


@pytest.mark.parametrize('n_crops,limit', [(12, 4000000), (65, 34000000)])
def test_render_in_sub_batches(tmp_path, n_crops, limit):
  """A render whose worst-case scratch (face lists sized for every face in every strip) exceeds the limit goes out in sub-batches
  (raster.hip: launch_render; 1 GiB by default - 220 full-frame poses) - the standalone render of the nvdiffrast_render API and the render
  inside a fused refinement pass alike.  With the limit lowered in a child process (the knob is read once per process): 12 crops in
  sub-batches of one or two; 65 crops as 33 + 32, where a plan made for the last 32 alone would use other strips than the one the
  scratch was sized for (every sub-batch runs on the plan of a full one).  Images and refined poses bit-identical to the single launch."""
  import os, subprocess, sys
  script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'render_dump.py')
  outs = []
  for name, env in (('whole', {}), ('chunked', {'FP_RENDER_SCRATCH_MAX': str(limit)})):
    path = str(tmp_path / (name + '.npz'))
    r = subprocess.run([sys.executable, script, path], env=dict(os.environ, N_CROPS=str(n_crops), **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    outs.append(np.load(path))
  for k in ('c', 'd', 'n', 'x', 'r'):
    assert np.array_equal(outs[0][k], outs[1][k]), k
  assert float((outs[0]['d'] > 0).mean()) > 0.15

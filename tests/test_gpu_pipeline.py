"""GPU parity of the networks and of the whole register()/track_one() path against the CPU oracle
and against the reference-generated golden vectors, through the Python mirror of the reference API
(which calls the C-ABI)."""
import numpy as np
import pytest
import torch

from tests import util
from tests.util import net_inputs

pytestmark = pytest.mark.gpu


def to_net_tensor(A, B):
  """(n,6,160,160) fp32 A,B -> fp16 NHWC8 [2n][160][160][8] on the device (A first, then B)."""
  x = torch.cat([A, B], 0).permute(0, 2, 3, 1)
  out = torch.zeros((x.shape[0], 160, 160, 8), dtype=torch.float16)
  out[..., :6] = x.half()
  return out.cuda().contiguous()


@pytest.fixture(scope='module')
def nets_gpu():
  from foundationpose_amd import _lib, synthetic as S
  ctx = _lib.Context.get('cuda:0')
  rsd, ssd = S.make_refine_state_dict(0), S.make_score_state_dict(1)
  return dict(ctx=ctx, rsd=rsd, ssd=ssd, rnet=_lib.DeviceNet(ctx, _lib.FP_NET_REFINE, rsd, True),
              snet=_lib.DeviceNet(ctx, _lib.FP_NET_SCORE, ssd, True), L=_lib)


def _refine_gpu(ng, net, A, B):
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  n = len(A)
  x = to_net_tensor(A, B)
  trans = torch.empty((n, 3), device='cuda')
  rot = torch.empty((n, net.rot_dim), device='cuda')
  check(lib().fp_refine_forward(ng['ctx'].handle, net.handle, ptr(x), n, ptr(trans), ptr(rot), stream_ptr()))
  return trans.cpu(), rot.cpu()


def _tokens_gpu(ng, net, A, B):
  """Trunk output of the HIP network as the reference's encodeAB[4] activation (N,512,20,20): tokens minus pos_embed."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  n = len(A)
  x = to_net_tensor(A, B)
  tok = torch.empty((n, 400, 512), dtype=torch.float16, device='cuda')
  check(lib().fp_net_tokens(ng['ctx'].handle, net.handle, ptr(x), n, ptr(tok), stream_ptr()))
  feat = tok.float().cpu() - S.positional_embedding()[:, :400]
  return feat.permute(0, 2, 1).reshape(n, 512, 20, 20)


def assert_tracks_input(got, want, rel, what):
  """`got` must follow the INPUT-DEPENDENT part of the reference output `want` (samples along axis 0): after removing each
  column's mean over the samples, the error has to stay below `rel` x that column's spread over the samples - a kernel that
  ignored, permuted or mixed up its inputs has an error of the order of the spread itself.  The common part (the mean over
  samples) has to agree to half a spread."""
  got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
  got, want = got.reshape(len(got), -1), want.reshape(len(want), -1)
  spread = want.std(0)
  dm = np.abs((got - got.mean(0)) - (want - want.mean(0))).max(0)
  common = np.abs(got.mean(0) - want.mean(0))
  live = spread > 0.05 * spread.mean()          # (sub-sampled activation taps hold a few dead / constant channels)
  print(f'{what}: spread over samples min {spread[live].min():.2e} mean {spread.mean():.2e}; de-meaned error max {dm.max():.2e} '
        f'(worst ratio {np.max(dm[live] / spread[live]):.3f}); common-part error max {common.max():.2e}')
  assert (dm[live] <= rel * spread[live]).all(), f'{what}: does not follow its input'
  assert dm.max() <= rel * spread.mean() * 4 and (common <= 0.5 * np.maximum(spread, spread.mean())).all(), what


def test_refine_net_vs_reference_golden(nets_gpu, golden):
  """The HIP RefineNet against the outputs of the REFERENCE's own nn.Module (fp32, CPU) on 8 golden input pairs that
  differ the way crops do (tests/util.py:net_inputs): trunk activations (encodeAB[4] tap) and both head outputs must follow
  the input-dependent part of the reference's: the 200 sub-sampled trunk activations each to 20 % of their own spread over the
  8 samples (single activations of small spread are the noisiest thing compared here: worst measured ratio 0.14), the head
  outputs to 10 % (measured 0.3 %) - fp16 operands / fp32 accumulation through 17 GEMM layers; the reference itself runs this
  net under fp16 autocast."""
  A, B = net_inputs(11, 8)
  feat = _tokens_gpu(nets_gpu, nets_gpu['rnet'], A, B)
  assert_tracks_input(feat[:, ::64, ::4, ::4].numpy(), golden['refine_encAB4_sub'], 0.2, 'encodeAB[4] tap')
  trans, rot = _refine_gpu(nets_gpu, nets_gpu['rnet'], A, B)
  assert_tracks_input(trans.numpy(), golden['refine_trans'], 0.1, 'trans head')
  assert_tracks_input(rot.numpy(), golden['refine_rot'], 0.1, 'rot head')
  np.testing.assert_allclose(trans.numpy(), golden['refine_trans'], atol=2e-3)
  np.testing.assert_allclose(rot.numpy(), golden['refine_rot'], atol=2e-3)
  # permuting the inputs permutes the outputs (and the permuted outputs differ from the unpermuted ones)
  perm = [3, 0, 7, 1, 6, 2, 5, 4]
  tp, rp = _refine_gpu(nets_gpu, nets_gpu['rnet'], A[perm], B[perm])
  assert torch.equal(tp, trans[perm]) and torch.equal(rp, rot[perm]) and not torch.equal(tp, trans)


def test_refine_net_no_bn_6d_vs_golden(nets_gpu, golden):
  from foundationpose_amd import _lib, synthetic as S
  sd = S.make_refine_state_dict(seed=2, use_bn=False, rot_out_dim=6)
  net = _lib.DeviceNet(nets_gpu['ctx'], _lib.FP_NET_REFINE, sd, use_bn=False)
  assert net.rot_dim == 6
  A, B = net_inputs(12, 4)
  trans, rot = _refine_gpu(nets_gpu, net, A, B)
  assert_tracks_input(trans.numpy(), golden['refine_nobn_trans'], 0.15, 'no-BN trans head')
  assert_tracks_input(rot.numpy(), golden['refine_nobn_rot'], 0.15, 'no-BN 6d rot head')
  np.testing.assert_allclose(trans.numpy(), golden['refine_nobn_trans'], atol=2e-3)
  np.testing.assert_allclose(rot.numpy(), golden['refine_nobn_rot'], atol=2e-3)


def test_score_net_vs_reference_golden(nets_gpu, golden):
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  A, B = net_inputs(13, 8)
  feat = _tokens_gpu(nets_gpu, nets_gpu['snet'], A, B)
  assert_tracks_input(feat[:, ::64, ::4, ::4].numpy(), golden['score_encAB4_sub'], 0.2, 'encoderAB[4] tap')
  x = to_net_tensor(A, B)
  feats = torch.empty((8, 512), device='cuda')
  check(lib().fp_score_features(nets_gpu['ctx'].handle, nets_gpu['snet'].handle, ptr(x), 8, ptr(feats), stream_ptr()))
  assert_tracks_input(feats.cpu().numpy(), golden['score_feats'], 0.1, 'ScoreNet features')
  for L, key in ((8, 'score_logit_L8'), (4, 'score_logit_L4')):
    logits = torch.empty((8 // L, L), device='cuda')
    am = torch.empty((8 // L,), dtype=torch.int32, device='cuda')
    check(lib().fp_score_tail(nets_gpu['ctx'].handle, nets_gpu['snet'].handle, ptr(feats), 8 // L, L, ptr(logits), ptr(am), stream_ptr()))
    want = golden[key]
    got = logits.cpu().numpy()
    spread = float(want.std())
    err = float(np.abs((got - got.mean(-1, keepdims=True)) - (want - want.mean(-1, keepdims=True))).max())
    print(f'{key}: logit spread {spread:.2e}, differential error {err:.2e}')
    assert err < 0.1 * spread
    np.testing.assert_array_equal(am.cpu().numpy(), want.argmax(-1))
    # the tail itself (fp32 SIMT) on the REFERENCE features must reproduce the reference logits tightly
    fref = torch.from_numpy(golden['score_feats']).cuda()
    check(lib().fp_score_tail(nets_gpu['ctx'].handle, nets_gpu['snet'].handle, ptr(fref), 8 // L, L, ptr(logits), ptr(am), stream_ptr()))
    np.testing.assert_allclose(logits.cpu().numpy(), want, atol=2e-5)
    np.testing.assert_array_equal(am.cpu().numpy(), want.argmax(-1))


def test_no_narrower_than_the_reference_autocast(nets_gpu, golden):
  """The precision claim of DESIGN.md section 2, pinned: the reference runs both networks under fp16 autocast
  (predict_pose_refine.py:190, predict_score.py:193).  tests/golden/gen_golden.py ran the reference's own modules under
  torch.autocast(fp16) next to the fp32 run; the HIP path - fp16 operands, fp32 accumulation, ONE extra fp16 rounding (the token
  tensor) - must not be further from the reference's fp32 outputs than 2 x what the reference's fp16 path is, on the trunk tap, the
  two heads and the ScoreNet features (rms over the fixture's elements).  The reference's fp16 error is printed beside ours."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  rms = lambda d: float(np.sqrt((np.asarray(d, dtype=np.float64) ** 2).mean()))
  A, B = net_inputs(11, 8)
  tap = _tokens_gpu(nets_gpu, nets_gpu['rnet'], A, B)[:, ::64, ::4, ::4].numpy()
  trans, rot = _refine_gpu(nets_gpu, nets_gpu['rnet'], A, B)
  A3, B3 = net_inputs(13, 8)
  stap = _tokens_gpu(nets_gpu, nets_gpu['snet'], A3, B3)[:, ::64, ::4, ::4].numpy()
  feats = torch.empty((8, 512), device='cuda')
  check(lib().fp_score_features(nets_gpu['ctx'].handle, nets_gpu['snet'].handle, ptr(to_net_tensor(A3, B3)), 8, ptr(feats), stream_ptr()))
  rows = (('encodeAB[4] tap', tap, 'refine_encAB4_sub'), ('trans head', trans.numpy(), 'refine_trans'), ('rot head', rot.numpy(), 'refine_rot'),
          ('encoderAB[4] tap', stap, 'score_encAB4_sub'), ('ScoreNet features', feats.cpu().numpy(), 'score_feats'))
  for what, got, key in rows:
    ours, theirs = rms(got - golden[key]), rms(golden[key + '_ac16'] - golden[key])
    print(f'{what}: rms |hip - ref fp32| = {ours:.2e}, rms |ref fp16 autocast - ref fp32| = {theirs:.2e} (ratio {ours / theirs:.2f})')
    assert theirs > 0 and ours <= 2.0 * theirs, what


def test_state_dict_errors(nets_gpu):
  from foundationpose_amd import _lib
  sd = dict(nets_gpu['rsd'])
  sd.pop('encodeAB.2.net.0.weight')
  with pytest.raises(_lib.FoundationPoseAmdError, match='encodeAB.2.net.0.weight'):
    _lib.DeviceNet(nets_gpu['ctx'], _lib.FP_NET_REFINE, sd, True)
  with pytest.raises(_lib.FoundationPoseAmdError):
    _lib.DeviceNet(nets_gpu['ctx'], _lib.FP_NET_SCORE, nets_gpu['rsd'], True)   # wrong key family


@pytest.fixture(scope='module')
def estimators():
  """FoundationPose (HIP) and the oracle on the same scene, rotation grid and weights."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.estimater import FoundationPose
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.predict_score import ScorePredictor
  from oracle.predict import OracleFoundationPose
  from tests import cases
  sc = util.scene(0)
  # these tests chain refinement iterations (and frames): the low-gain refiner, see tests/cases.py
  rsd, ssd = S.make_refine_state_dict(0, head_gain=cases.GAIN_CHAIN), S.make_score_state_dict(1)
  mesh = S.make_mustard_mesh(seed=0)
  refiner = PoseRefinePredictor(state_dict=rsd, cfg=REFINE_DEFAULT)
  scorer = ScorePredictor(state_dict=ssd, cfg=SCORE_DEFAULT)
  np.random.seed(0)
  est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, refiner=refiner, scorer=scorer)
  # the rotation grid is an input fixture shared by both sides (icosphere order is unpinned)
  np.testing.assert_allclose(est.rot_grid.cpu().numpy(), sc['grid'], atol=1e-6)
  assert est.rot_grid.shape == (252, 4, 4)
  assert abs(est.diameter - sc['diameter']) < 2e-3
  est.diameter = sc['diameter']
  orc = OracleFoundationPose(sc['mt'], sc['diameter'], est.model_center, sc['grid'], rsd, ssd,
                             refine_cfg=dict(REFINE_DEFAULT), score_cfg=dict(SCORE_DEFAULT))
  return dict(sc=sc, est=est, orc=orc)


def test_register_matches_oracle_config0(estimators):
  """BASELINE config[0] (32 hypotheses, est_refine_iter=1) and a 2-iteration run.
  Per hypothesis: refined 4x4 pose within 1e-3 of the oracle (north_star tolerance); score logits
  within the fp16 noise floor (<5 % of the logit spread across hypotheses); identical argmax and
  best pose, with the oracle's top-1/top-2 margin at least 20x the measured logit noise."""
  from oracle import geometry as G
  from oracle import predict as OP
  sc, est, orc = estimators['sc'], estimators['est'], estimators['orc']
  full_g, full_o = est.rot_grid, orc.rot_grid
  try:
    est.rot_grid, orc.rot_grid = full_g[:32].contiguous(), full_o[:32]
    depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
    xyz_map = G.depth2xyzmap(depth, sc['K'])
    poses0 = util.hypotheses(sc, 32)
    for iteration in (1, 2):
      pg, _ = est.refiner.predict(mesh=est.mesh, mesh_tensors=est.mesh_tensors, rgb=sc['rgb'], depth=depth, K=sc['K'], ob_in_cams=poses0,
                                  xyz_map=xyz_map, mesh_diameter=est.diameter, iteration=iteration)
      po = OP.refine_predict(orc.refine_cfg, orc.refine_sd, sc['rgb'], depth, sc['K'], poses0, xyz_map, sc['mt'], sc['diameter'],
                             iteration=iteration, chunk=16)
      perr = float((pg.cpu() - po).abs().max())
      print(f'iteration={iteration}: max |pose_gpu - pose_oracle| over 32 hypotheses = {perr:.2e}')
      assert perr < 1e-3
      # scoring on the ORACLE's refined poses (isolates the scorer from refinement noise)
      # raw logits on both sides: the reference's `+ 100` (predict_score.py:209) quantises float32 scores to 7.6e-6
      fg = est.scorer.extract_features(sc['rgb'], depth, sc['K'], po.numpy(), mesh_tensors=est.mesh_tensors, mesh_diameter=est.diameter)
      sg = est.scorer.score_tail(fg, L=len(fg))[0].reshape(-1).cpu().numpy()
      tr = []
      OP.score_predict(orc.score_cfg, orc.score_sd, sc['rgb'], depth, sc['K'], po.numpy(), sc['mt'], sc['diameter'], chunk=16, trace=tr)
      so = tr[0]['logits'].numpy()
      # a common shift of all logits cannot change the ranking: split the error into common + differential
      common = float((sg - so).mean())
      noise, spread = float(np.abs((sg - sg.mean()) - (so - so.mean())).max()), float(so.std())
      top = np.sort(so)[::-1]
      margin = float(top[0] - top[1])
      print(f'iteration={iteration}: logit common shift {common:.2e}, differential noise {noise:.2e}, spread {spread:.2e}, '
            f'top1-top2 margin {margin:.2e}, argmax gpu/oracle {int(sg.argmax())}/{int(so.argmax())}')
      assert abs(common) < 5e-3 and noise < 0.25 * spread
      # identical argmax, unconditionally: the ScoreNet tail seed is chosen (tests/golden/gen_fullsize.py) so that the oracle's
      # top-1 / top-2 margin is far above the fp16 logit noise in every case of tests/cases.py
      assert margin >= 20 * noise, f'margin {margin:.2e} vs logit noise {noise:.2e}'
      assert int(sg.argmax()) == int(so.argmax())
      # the integrated call
      pose_g = est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=iteration)
      pose_o = orc.register(sc['K'], sc['rgb'], sc['depth'], sc['mask'], iteration=iteration, chunk=16)
      assert pose_g.dtype == np.float32 and pose_g.shape == (4, 4)
      assert est.poses.shape == (32, 4, 4) and est.scores.shape == (32,)
      assert bool((est.scores[:-1] >= est.scores[1:]).all())
      assert int(est.best_id) == int(orc.best_id)
      np.testing.assert_allclose(pose_g, pose_o, atol=1e-3)
      assert util.nearest_pose_error(est.poses.cpu().numpy(), np.asarray(orc.poses)) < 1e-3      # all 32 (near-ties may rank differently)
  finally:
    est.rot_grid, orc.rot_grid = full_g, full_o


def test_register_degenerate_inputs(estimators):
  """The reference's guards (src/estimater.py:139-147,185-189,251-253)."""
  from foundationpose_amd.estimater import FoundationPose
  sc, est = estimators['sc'], estimators['est']
  empty = np.zeros_like(sc['mask'])
  pose = est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=empty, iteration=1)
  assert pose.dtype == np.float64 and np.array_equal(pose, np.eye(4))
  tiny = empty.copy(); tiny[240, 320] = True; tiny[240, 321] = True
  pose = est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=tiny, iteration=1)
  assert np.array_equal(pose[:3, :3], np.eye(3)) and pose[2, 3] > 0
  fresh = object.__new__(FoundationPose)
  fresh.pose_last = None
  with pytest.raises(RuntimeError):
    fresh.track_one(sc['rgb'], sc['depth'], sc['K'], iteration=2)


def test_track_one_matches_oracle(estimators):
  sc, est, orc = estimators['sc'], estimators['est'], estimators['orc']
  start = torch.as_tensor(util.hypotheses(sc, 1)[0])
  start[:3, :3] = torch.as_tensor(sc['gt_pose'][:3, :3])
  est.pose_last = start.cuda()
  orc.pose_last = start.clone()
  for _ in range(2):
    pg = est.track_one(sc['rgb'], sc['depth'], sc['K'], iteration=2)
    po = orc.track_one(sc['rgb'], sc['depth'], sc['K'], iteration=2)
    assert pg.shape == (4, 4)
    np.testing.assert_allclose(pg, po, atol=1e-3)


def test_register_through_dist_group_world1(estimators):
  """The sharded path (foundationpose_amd/dist.py) over RCCL with a single rank must reproduce the
  plain path bit for bit (same kernels, same order; the all-gather is the identity)."""
  import os
  import torch.distributed as dist
  sc, est = estimators['sc'], estimators['est']
  full = est.rot_grid
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29617')
  dist.init_process_group('nccl', rank=0, world_size=1)
  try:
    est.rot_grid = full[:16].contiguous()
    p0 = est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=2)
    s0, id0 = est.scores.clone(), int(est.best_id)
    est.dist_group = dist.group.WORLD
    p1 = est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=2)
    assert int(est.best_id) == id0
    np.testing.assert_array_equal(p1, p0)
    assert torch.equal(est.scores, s0)
  finally:
    est.dist_group = None
    est.rot_grid = full
    dist.destroy_process_group()


def test_readme_pipeline_protocol_end_to_end(estimators):
  """readme.md:122-179: Pipeline + FoundationPoseEstimator + PoseTransformer -> data.pose_6d; the second
  frame (no mask) goes through track_one."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.pipeline import FoundationPoseEstimator, Pipeline, PipelineData, PoseTransformer, Processor
  sc, est = estimators['sc'], estimators['est']

  class Frame(Processor):
    def __init__(self, with_mask): self.with_mask = with_mask
    def process(self, data):
      data.rgb, data.depth, data.K = sc['rgb'], sc['depth'], sc['K']
      data.mask = sc['mask'] if self.with_mask else None
      return data
  stage = FoundationPoseEstimator(mesh=S.make_mustard_mesh(seed=0), K=sc['K'], est_refine_iter=1, track_refine_iter=1,
                                  scorer=est.scorer, refiner=est.refiner)
  np.random.seed(0)
  pipe = Pipeline('demo', stop_on_error=True).add_processor(Frame(True)).add_processor(stage).add_processor(PoseTransformer())
  out = pipe.run(PipelineData())
  assert not out.errors and out.pose.shape == (4, 4) and len(out.pose_6d) == 6 and np.isfinite(out.pose_6d).all()
  pipe.processors[0] = Frame(False)
  out2 = pipe.run(PipelineData())
  assert not out2.errors and np.abs(out2.pose - out.pose).max() < 0.1


def test_other_frame_size_and_intrinsics(estimators):
  """Nothing in the path is tied to 480x640 / the YCB intrinsics: a 360x500 frame (odd strides for every 16-byte access),
  a camera with other focal lengths and an off-centre principal point, object near the image border.  4 hypotheses,
  2 refinement iterations + scoring against the oracle, same tolerances as config[0]."""
  from foundationpose_amd import synthetic as S
  from oracle import geometry as G
  from oracle import predict as OP
  from oracle.render import nvdiffrast_render as oracle_render
  sc0, est, orc = estimators['sc'], estimators['est'], estimators['orc']
  H, W = 360, 500
  K = np.array([[610.0, 0, 231.5], [0, 595.0, 171.25], [0, 0, 1]])

  def rf(K_, H_, W_, pose):
    c, d, _ = oracle_render(K=K_, H=H_, W=W_, ob_in_cams=pose, mesh_tensors=sc0['mt'], use_light=True)
    return c[0].numpy(), d[0].numpy()
  sc = S.make_scene(rf, sc0['mt'], seed=5, H=H, W=W, K=K, t=(-0.16, 0.09, 0.62))       # partly cut by the left border
  assert sc['mask'].any() and sc['mask'][:, 0].any()
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  xyz_map = G.depth2xyzmap(depth, K)
  poses0 = sc0['grid'][[0, 7, 100, 251]].copy()
  poses0[:, :3, 3] = G.guess_translation(depth, sc['mask'], K).astype(np.float32)
  pg, _ = est.refiner.predict(mesh=est.mesh, mesh_tensors=est.mesh_tensors, rgb=sc['rgb'], depth=depth, K=K, ob_in_cams=poses0,
                              xyz_map=xyz_map, mesh_diameter=est.diameter, iteration=2)
  po = OP.refine_predict(orc.refine_cfg, orc.refine_sd, sc['rgb'], depth, K, poses0, xyz_map, sc0['mt'], sc0['diameter'], iteration=2, chunk=4)
  assert float((pg.cpu() - po).abs().max()) < 1e-3
  sg, _ = est.scorer.predict(mesh=est.mesh, mesh_tensors=est.mesh_tensors, rgb=sc['rgb'], depth=depth, K=K, ob_in_cams=po.numpy(),
                             mesh_diameter=est.diameter)
  so = OP.score_predict(orc.score_cfg, orc.score_sd, sc['rgb'], depth, K, po.numpy(), sc0['mt'], sc0['diameter'], chunk=4)
  sg, so = sg.cpu().numpy(), so.numpy()
  assert np.abs((sg - sg.mean()) - (so - so.mean())).max() < 0.25 * max(float(so.std()), 1e-4) and abs(float((sg - so).mean())) < 5e-3
  # the integrated call on this frame (device prelude: filtering, stats, float64 back-projection)
  pose = est.register(K=K, rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=1)
  assert pose.shape == (4, 4) and np.isfinite(pose).all()


@pytest.mark.parametrize('variant', ['plain_xyz_tanh', '6d_no_bn', 'deepim'])
def test_predictors_other_config_branches(variant):
  """The config branches the released models do not take but the reference implements (predict_pose_refine.py:195-231,
  h5_dataset.py:92-99,151-156): normalize_xyz=False (xyz only centred, translation through tanh * trans_normalizer) and
  rot_rep='6d' without BatchNorm.  8 hypotheses, 2 iterations, through PoseRefinePredictor / ScorePredictor vs the oracle's
  predict loops: poses within 1e-3 (north_star), same best hypothesis, logits within 2e-3."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.predict_score import ScorePredictor
  from oracle import geometry as G, predict as OP
  sc = util.scene(0)
  poses = util.hypotheses(sc, 8, jitter_seed=11)
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  xyz_map = G.depth2xyzmap(depth, sc['K'])
  mt = util.to_dev(sc['mt'])
  if variant == 'plain_xyz_tanh':
    rcfg = dict(REFINE_DEFAULT, normalize_xyz=False)
    scfg = dict(SCORE_DEFAULT, normalize_xyz=False)
    rsd, ssd = S.make_refine_state_dict(0, head_gain=0.1), S.make_score_state_dict(1)      # two chained iterations: low gain
  elif variant == 'deepim':
    # trans_rep='deepim' (predict_pose_refine.py:201-215): the head's third output is the depth RATIO, so its bias moves to 1
    rcfg = dict(REFINE_DEFAULT, trans_rep='deepim', normalize_xyz=False)
    scfg = dict(SCORE_DEFAULT, normalize_xyz=False)
    rsd, ssd = S.make_refine_state_dict(0, head_gain=0.1), S.make_score_state_dict(1)
    rsd['trans_head.1.bias'] = rsd['trans_head.1.bias'] + torch.tensor([0.0, 0.0, 1.0])
  else:
    rcfg = dict(REFINE_DEFAULT, rot_rep='6d', use_BN=False)
    scfg = dict(SCORE_DEFAULT, use_BN=False)
    rsd, ssd = S.make_refine_state_dict(seed=2, use_bn=False, rot_out_dim=6), S.make_score_state_dict(seed=3, use_bn=False)
  refiner = PoseRefinePredictor(state_dict=rsd, cfg=rcfg)
  got, _ = refiner.predict(sc['rgb'], depth, sc['K'], poses, xyz_map, mesh_tensors=mt, mesh_diameter=sc['diameter'], iteration=2)
  ocfg = dict(OP.DEFAULT_REFINE_CFG, **{k: rcfg[k] for k in ('normalize_xyz', 'rot_rep', 'use_BN', 'trans_rep')})
  ref = OP.refine_predict(ocfg, rsd, sc['rgb'], depth, sc['K'], poses, xyz_map, sc['mt'], sc['diameter'], iteration=2, chunk=8)
  assert float((torch.as_tensor(poses) - ref).abs().max()) > 1e-3          # the networks moved the poses
  np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=1e-3)
  if variant == 'deepim':
    return                  # trans_rep only exists in the refiner; the scorer's normalize_xyz=False branch is the first variant's
  scorer = ScorePredictor(state_dict=ssd, cfg=scfg)
  s_got, _ = scorer.predict(sc['rgb'], depth, sc['K'], ref.numpy(), mesh_tensors=mt, mesh_diameter=sc['diameter'])
  oscfg = dict(OP.DEFAULT_SCORE_CFG, **{k: scfg[k] for k in ('normalize_xyz', 'use_BN')})
  s_ref = OP.score_predict(oscfg, ssd, sc['rgb'], depth, sc['K'], ref.numpy(), sc['mt'], sc['diameter'], chunk=8)
  np.testing.assert_allclose(s_got.cpu().numpy(), s_ref.numpy(), atol=2e-3)
  assert int(s_got.argmax()) == int(s_ref.argmax())


def test_register_textured_symmetric_object():
  """A textured mesh (uv + texture image path of make_mesh_tensors / the rasteriser, src/Utils.py:104-130,196-199) declared
  2-fold symmetric about z (symmetry_tfs thins the rotation grid through the native cluster_poses, src/estimater.py:106-124),
  through FoundationPose.register() against the oracle on the same grid: 24 hypotheses, 1 iteration."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.estimater import FoundationPose
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.predict_score import ScorePredictor
  from oracle import geometry as G
  from oracle.predict import OracleFoundationPose
  sc = util.scene(0, textured=True)
  assert 'tex' in sc['mt'] and 'uv_idx' in sc['mt']
  rsd, ssd = S.make_refine_state_dict(0), S.make_score_state_dict(1)
  mesh = S.make_mustard_mesh(seed=0, textured=True)
  sym = np.stack([np.eye(4), np.diag([-1.0, -1.0, 1.0, 1.0])])
  np.random.seed(0)
  est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, symmetry_tfs=sym,
                       refiner=PoseRefinePredictor(state_dict=rsd, cfg=REFINE_DEFAULT), scorer=ScorePredictor(state_dict=ssd, cfg=SCORE_DEFAULT))
  grid_o = G.make_rotation_grid(symmetry_tfs=sym)
  assert 40 <= len(grid_o) < 252 and est.rot_grid.shape == grid_o.shape
  np.testing.assert_allclose(est.rot_grid.cpu().numpy(), grid_o, atol=1e-6)
  est.diameter = sc['diameter']
  est.rot_grid = est.rot_grid[:24].contiguous()
  orc = OracleFoundationPose(sc['mt'], sc['diameter'], est.model_center, grid_o[:24], rsd, ssd, refine_cfg=dict(REFINE_DEFAULT),
                             score_cfg=dict(SCORE_DEFAULT))
  pose_g = est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=1)
  pose_o = orc.register(sc['K'], sc['rgb'], sc['depth'], sc['mask'], iteration=1, chunk=12)
  so, sg = np.asarray(orc.scores), est.scores.cpu().numpy()           # both sorted, best first
  np.testing.assert_allclose(sg, so, atol=3e-3)
  # (scores = logits + 100 in float32 on both sides: quantised to 7.6e-6; the raw logits, the margin and the measured noise
  # of this very case are asserted in tests/test_gpu_golden_fullsize.py, case tex24)
  assert int(est.best_id) == int(orc.best_id)
  np.testing.assert_allclose(pose_g, pose_o, atol=1e-3)
  assert util.nearest_pose_error(est.poses.cpu().numpy(), np.asarray(orc.poses)) < 1e-3


def test_to_device_moves_the_predictors(nets_gpu):
  """src/estimater.py:88-102: to_device moves the tensors, both networks and re-creates the raster context.  On a one-GPU box
  the target is the device everything already lives on (no copy is made); the rebuild that a move to another GPU performs -
  fp_net re-created from the host copies of the parameters, the old one freed - is driven directly and must not change a bit."""
  from foundationpose_amd import _lib
  ng = nets_gpu
  A, B = net_inputs(5, 3)
  before = _refine_gpu(ng, ng['rnet'], A, B)
  net = _lib.DeviceNet(ng['ctx'], _lib.FP_NET_REFINE, ng['rsd'], True)
  old = net.handle
  assert net.to('cuda:0') is net and net.handle is old              # same device: nothing happens
  net._create(ng['ctx'])                                            # what to() does for a different device
  _lib.lib().fp_net_destroy(old)
  after = _refine_gpu(ng, net, A, B)
  assert torch.equal(before[0], after[0]) and torch.equal(before[1], after[1])

  class Pred:                                                       # records what FoundationPose.to_device asks of a predictor
    def __init__(self):
      self.moved = None

    def to_device(self, s):
      self.moved = s
  from foundationpose_amd.estimater import FoundationPose
  est = FoundationPose.__new__(FoundationPose)
  est.mesh_tensors = {'pos': torch.zeros(3, 3)}
  est.some_tensor = torch.ones(2)
  est.refiner, est.scorer, est.glctx = Pred(), Pred(), None
  est.to_device('cuda:0')
  assert est.refiner.moved == 'cuda:0' and est.scorer.moved == 'cuda:0'
  assert est.some_tensor.is_cuda and est.mesh_tensors['pos'].is_cuda


def test_shared_translation_first_pass_is_bit_identical():
  """FP_REFINE_SHARED_TRANSLATION (register's hypotheses: one rotation grid around ONE centre, src/estimater.py:126-135): the observed side of
  the first iteration is cropped and run through encodeA once per object instead of once per hypothesis.  Same kernels on the same
  inputs: the refined poses equal those of the plain pass BIT FOR BIT - 2 hypotheses (the few-image kernels), 8 (two chains), 56 and 100
  (the trunk in two halves), and three objects of 5 + 0 + 7 hypotheses in one pass; a host array is detected, a device tensor is not."""
  from oracle import geometry as G
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  sc = util.scene(0)
  r = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT)
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  xyz = G.depth2xyzmap(depth, sc['K'])
  mt = util.to_dev(sc['mt'])
  kw = dict(rgb=sc['rgb'], depth=depth, K=sc['K'], xyz_map=xyz, mesh_tensors=mt, mesh_diameter=sc['diameter'], iteration=2)
  assert r._shares_translation(util.hypotheses(sc, 4), None) and not r._shares_translation(util.hypotheses(sc, 4, jitter_seed=1), None)
  assert not r._shares_translation(torch.from_numpy(util.hypotheses(sc, 4)).cuda(), None)
  for n in (2, 8, 56, 100):
    poses = np.ascontiguousarray(np.concatenate([sc['grid']] * 2)[:n].astype(np.float32))
    poses[:, :3, 3] = util.hypotheses(sc, 1)[0, :3, 3]
    plain, _ = r.predict(ob_in_cams=poses, shared_translation=False, **kw)
    once, _ = r.predict(ob_in_cams=poses, shared_translation=True, **kw)
    auto, _ = r.predict(ob_in_cams=poses, **kw)
    assert torch.equal(plain, once) and torch.equal(plain, auto), f'{n} hypotheses: {float((plain - once).abs().max()):.2e}'
    assert float((plain.cpu() - torch.from_numpy(poses)).abs().max()) > 1e-4
  objs = []
  for k, n in enumerate((5, 0, 7)):
    p = sc['grid'][10 * k:10 * k + n].astype(np.float32).copy()
    p[:, :3, 3] = util.hypotheses(sc, 1)[0, :3, 3] + np.float32(0.004 * k)
    objs.append(dict(rgb=sc['rgb'], xyz_map=xyz, K=sc['K'], mesh_tensors=mt, mesh_diameter=sc['diameter'], ob_in_cams=p))
  plain = r.predict_multi([dict(o, shared_translation=False) for o in objs], iteration=2)
  once = r.predict_multi(objs, iteration=2)
  assert torch.equal(plain, once) and len(once) == 12


def test_schedules_and_kernel_forms_are_bit_identical():
  """Schedules and kernel forms that claim bit-identical results (one process each: the knobs are read once): the two sides of encodeA as
  one chain instead of two (FP_ONE_CHAIN=1), the trunk as one batch instead of two halves on two streams (FP_TRUNK_STREAMS=1), both heads
  on one stream (FP_HEADS_SERIAL=1), the 128 -> 128 layers on the general 3x3 kernel instead of the band form (FP_C128_BAND=0), the
  in-projections on the 64-token kernel instead of tok_qkv.hip (FP_QKV64=1), 512-pixel tiles only in the 3x3 kernel (FP_HALO_TAIL=0), the tail
  of a refinement pass (token means of both heads, pose update, next crop windows) as four launches instead of one (FP_TAIL_SPLIT=1), the
  rasteriser of one or two hypotheses as three launches instead of one (FP_RENDER_SOLO=0), the observed side of a first iteration per
  hypothesis instead of once per object (FP_NO_SHARED_B=1).  The
  fused passes at 1, 2, 8, 40, 56 and 100 hypotheses (tests/tools/variant_digest.py) must print the digests of the default build."""
  import json, os, subprocess, sys
  script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'variant_digest.py')

  def digests(knobs):
    r = subprocess.run([sys.executable, script], env=dict(os.environ, **knobs), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f'{knobs}: {r.stdout[-1500:]}\n{r.stderr[-1500:]}'
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('DIGEST ')][-1][7:])
  ref = digests({})
  assert len(set(ref.values())) == len(ref)
  for knobs in ({'FP_ONE_CHAIN': '1'}, {'FP_TRUNK_STREAMS': '1', 'FP_HEADS_SERIAL': '1'}, {'FP_C128_BAND': '0', 'FP_QKV64': '1'}, {'FP_HALO_TAIL': '0', 'FP_TAIL_SPLIT': '1'},
                {'FP_RENDER_SOLO': '0', 'FP_NO_SHARED_B': '1'}):
    assert digests(knobs) == ref, knobs

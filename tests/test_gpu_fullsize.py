"""GPU, BASELINE.json full sizes (252 hypotheses, est_refine_iter=5; 4 objects x 252): size-independent
properties the domain offers, since the CPU oracle needs minutes at this size -
  * per-hypothesis independence: batch-split invariance and permutation equivariance, BIT-exact
    (every output element has a fixed accumulation order, independent of tiling and batch position);
  * determinism (visibility is resolved by integer keys, no float atomics anywhere);
  * the 8-way sharded schedule of bench.py / dist.py reproduces the unsharded result bit for bit;
  * per-object grouping of the score tail (configs[3]: 4 objects x 252);
  * refined poses stay rigid transforms; scores are sorted; argmax is a valid hypothesis index."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

N = 252


@pytest.fixture(scope='module')
def env():
  from foundationpose_amd import _lib, synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.predict_score import ScorePredictor
  sc = util.scene(0)
  refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT)
  scorer = ScorePredictor(state_dict=S.make_score_state_dict(1), cfg=SCORE_DEFAULT)
  refiner.ctx.reserve(N)
  from oracle import geometry as G
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  return dict(sc=sc, refiner=refiner, scorer=scorer, depth=depth, xyz=G.depth2xyzmap(depth, sc['K']), mt=util.to_dev(sc['mt']),
              poses=torch.from_numpy(util.hypotheses(sc, N)).cuda(), lib=_lib)


def _net_inputs(n, seed):
  g = torch.Generator(device='cuda').manual_seed(seed)
  x = torch.zeros((2 * n, 160, 160, 8), dtype=torch.float16, device='cuda')
  x[..., :3] = torch.rand((2 * n, 160, 160, 3), device='cuda', generator=g).half()
  xyz = torch.randn((2 * n, 160, 160, 3), device='cuda', generator=g) * 0.5
  xyz[torch.rand((2 * n, 160, 160, 1), device='cuda', generator=g).expand_as(xyz) < 0.4] = 0
  x[..., 3:6] = xyz.half()
  return x


def _refine(env, x, n):
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  r = env['refiner']
  trans = torch.empty((n, 3), device='cuda')
  rot = torch.empty((n, 3), device='cuda')
  check(lib().fp_refine_forward(r.ctx.handle, r.model.handle, ptr(x), n, ptr(trans), ptr(rot), stream_ptr()))
  return trans, rot


def test_refine_net_batch_split_permutation_determinism(env):
  x = _net_inputs(N, 5)
  t_all, r_all = _refine(env, x, N)
  t_again, r_again = _refine(env, x, N)
  assert torch.equal(t_all, t_again) and torch.equal(r_all, r_again)                  # determinism
  assert float(t_all.std(0).min()) > 0 and torch.isfinite(t_all).all() and torch.isfinite(r_all).all()
  # split 252 = 100 + 152 (different tile boundaries): A|B halves re-packed per part
  for a, b in ((0, 100), (100, N)):
    part = torch.cat([x[a:b], x[N + a:N + b]], 0).contiguous()
    t_p, r_p = _refine(env, part, b - a)
    assert torch.equal(t_p, t_all[a:b]) and torch.equal(r_p, r_all[a:b])
  perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).cuda()
  xp = torch.cat([x[:N][perm], x[N:][perm]], 0).contiguous()
  t_q, r_q = _refine(env, xp, N)
  assert torch.equal(t_q, t_all[perm]) and torch.equal(r_q, r_all[perm])


def test_register_core_252_iter5_and_8way_sharding(env):
  """configs[1] + configs[2]: refine x5 + features + tail on 252 hypotheses; then the same with the
  hypotheses cut into the 8 shards bench.py/dist.py use (32 x7 + 28), processed one after the other on
  this GPU, gathered, and scored - must equal the unsharded run bit for bit, including the argmax."""
  from foundationpose_amd.dist import pack_rows, shard_ranges, unpack_rows
  sc, refiner, scorer = env['sc'], env['refiner'], env['scorer']
  kw = dict(rgb=sc['rgb'], depth=env['depth'], K=sc['K'], mesh_tensors=env['mt'], mesh_diameter=sc['diameter'])
  refined, _ = refiner.predict(ob_in_cams=env['poses'], xyz_map=env['xyz'], iteration=5, **kw)
  feats = scorer.extract_features(ob_in_cams=refined, **kw)
  logits, am = scorer.score_tail(feats, L=N)
  R = refined[:, :3, :3]
  eye = torch.eye(3, device='cuda')[None]
  assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 5e-5      # five chained updates keep R orthonormal
  assert float((torch.linalg.det(R) - 1).abs().max()) < 5e-5
  assert torch.equal(refined[:, 3], torch.tensor([0., 0., 0., 1.], device='cuda').expand(N, 4))
  assert float((refined[:, :3, 3] - env['poses'][:, :3, 3]).abs().max()) < 0.2   # refinement steps stay bounded
  assert 0 <= int(am[0]) < N and int(am[0]) == int(logits.argmax())
  world, shard = 8, 32
  blocks = []
  for a, b in shard_ranges(N, world):
    p, _ = refiner.predict(ob_in_cams=env['poses'][a:b], xyz_map=env['xyz'], iteration=5, **kw)
    f = scorer.extract_features(ob_in_cams=p, **kw)
    blocks.append(pack_rows(f, p, shard))
  f2, p2 = unpack_rows(torch.cat(blocks, 0), N, world)
  assert torch.equal(p2, refined) and torch.equal(f2, feats)
  logits2, am2 = scorer.score_tail(f2, L=N)
  assert torch.equal(logits2, logits) and int(am2[0]) == int(am[0])


def test_score_tail_groups_4x252(env):
  """configs[3]: 4 concurrent objects x 252 hypotheses -> per-object logits / argmax equal the 4 single-object calls."""
  scorer = env['scorer']
  g = torch.Generator(device='cuda').manual_seed(9)
  feats = torch.randn((4 * N, 512), device='cuda', generator=g)
  logits, am = scorer.score_tail(feats, L=N)
  assert logits.shape == (4, N) and am.shape == (4,)
  for o in range(4):
    lo, ao = scorer.score_tail(feats[o * N:(o + 1) * N].contiguous(), L=N)
    assert torch.equal(lo[0], logits[o]) and int(ao[0]) == int(am[o]) == int(logits[o].argmax())
  # permuting the hypotheses of one object permutes its logits (set-equivariance of att_cross), within fp32 summation noise
  perm = torch.randperm(N, generator=torch.Generator().manual_seed(2)).cuda()
  lp, _ = scorer.score_tail(feats[:N][perm].contiguous(), L=N)
  np.testing.assert_allclose(lp[0].cpu().numpy(), logits[0][perm].cpu().numpy(), atol=2e-6)


def test_renderer_determinism_and_pose_locality(env):
  """252 crops rendered twice are identical; changing one pose changes only that crop."""
  from foundationpose_amd._lib import check, k_ptr, lib, ptr, stream_ptr
  sc, ctx = env['sc'], env['refiner'].ctx
  dm = env['lib'].device_mesh(ctx, env['mt'])
  poses = env['poses'].clone()
  Kd, Kp = k_ptr(sc['K'])
  tf = torch.empty((N, 3, 3), device='cuda'); bbox = torch.empty((N, 4), device='cuda')

  def render(p):
    out = torch.empty((N, 160, 160, 8), dtype=torch.float16, device='cuda')
    check(lib().fp_crop_window_tf(ctx.handle, ptr(p), N, Kp, 1.2, sc['diameter'], 160, 160, ptr(tf), ptr(bbox), stream_ptr()))
    check(lib().fp_render_net(ctx.handle, dm.handle, ptr(p), N, Kp, 480, 640, ptr(bbox), 160, 160, sc['diameter'], 1, 0.001, ptr(out), stream_ptr()))
    return out
  a, b = render(poses), render(poses)
  assert torch.equal(a, b)
  cov = (a[..., 5] != 0).float().mean(dim=(1, 2))
  assert float(cov.min()) > 0.05                         # every hypothesis shows the object (end-on views cover ~10 %)
  poses[17, :3, 3] += torch.tensor([0.004, -0.003, 0.01], device='cuda')
  c = render(poses)
  same = (a == c).flatten(1).all(1)
  assert not bool(same[17]) and bool(same[torch.arange(N, device='cuda') != 17].all())


def test_multi_object_pass_equals_per_object_passes(env):
  """configs[3]-style batching (fp_refine_predict_multi / fp_score_predict_features_multi): two objects (different
  frames, 100 + 152 hypotheses) through ONE network pass per iteration == each object on its own, bit for bit."""
  from oracle import geometry as G
  sc, refiner, scorer = env['sc'], env['refiner'], env['scorer']
  sc2 = util.scene(1)
  depth2 = G.bilateral_filter_depth(G.erode_depth(sc2['depth']))
  objs = [dict(rgb=sc['rgb'], depth=env['depth'], xyz_map=env['xyz'], K=sc['K'], mesh_tensors=env['mt'], mesh_diameter=sc['diameter'],
               ob_in_cams=env['poses'][:100]),
          dict(rgb=sc2['rgb'], depth=depth2, xyz_map=G.depth2xyzmap(depth2, sc2['K']), K=sc2['K'], mesh_tensors=env['mt'],
               mesh_diameter=sc['diameter'], ob_in_cams=torch.from_numpy(util.hypotheses(sc2, 152)).cuda())]
  # objects 0 and 1 share mesh and camera (their crop windows and renders merge into one launch); object 2 has another
  # camera matrix and starts a launch of its own
  K3 = sc['K'].copy()
  K3[0, 0] *= 1.03
  K3[1, 2] += 2.5
  objs.append(dict(objs[0], K=K3, xyz_map=G.depth2xyzmap(env['depth'], K3), ob_in_cams=env['poses'][100:140]))
  refined = refiner.predict_multi(objs, iteration=2)
  assert refined.shape == (292, 4, 4)
  parts = []
  for ob in objs:
    p, _ = refiner.predict(rgb=ob['rgb'], depth=ob['depth'], K=ob['K'], ob_in_cams=ob['ob_in_cams'], xyz_map=ob['xyz_map'],
                           mesh_tensors=ob['mesh_tensors'], mesh_diameter=ob['mesh_diameter'], iteration=2)
    parts.append(p)
  assert torch.equal(refined, torch.cat(parts, 0))
  feats = scorer.extract_features_multi([dict(ob, ob_in_cams=p) for ob, p in zip(objs, parts)])
  fparts = [scorer.extract_features(ob['rgb'], ob['depth'], ob['K'], p, mesh_tensors=ob['mesh_tensors'], mesh_diameter=ob['mesh_diameter'])
            for ob, p in zip(objs, parts)]
  assert torch.equal(feats, torch.cat(fparts, 0))
  # an object with zero hypotheses on this rank is legal (252 hypotheses over more ranks than shards)
  empty = dict(objs[0], ob_in_cams=env['poses'][:0])
  r2 = refiner.predict_multi([empty, objs[1]], iteration=1)
  assert r2.shape == (152, 4, 4)


def test_four_objects_in_one_network_pass_1008(env):
  """configs[3] at full size in ONE pass: RefineNet and the ScoreNet feature extractor on 4 x 252 = 1008 hypotheses
  (2016 input images, 51.6 M stem pixels: the largest tensors the 32-bit lane offsets of the kernels see) must equal four
  252-hypothesis passes bit for bit."""
  from foundationpose_amd._lib import check, lib, ptr, stream_ptr
  r, sc = env['refiner'], env['scorer']
  n4 = 4 * N
  r.ctx.reserve(n4)
  x = _net_inputs(n4, 17)
  t_all, r_all = _refine(env, x, n4)
  feats = torch.empty((n4, 512), device='cuda')
  check(lib().fp_score_features(sc.ctx.handle, sc.model.handle, ptr(x), n4, ptr(feats), stream_ptr()))
  assert torch.isfinite(t_all).all() and torch.isfinite(feats).all()
  for o in range(4):
    part = torch.cat([x[o * N:(o + 1) * N], x[n4 + o * N:n4 + (o + 1) * N]], 0).contiguous()
    t_p, r_p = _refine(env, part, N)
    assert torch.equal(t_p, t_all[o * N:(o + 1) * N]) and torch.equal(r_p, r_all[o * N:(o + 1) * N])
    f_p = torch.empty((N, 512), device='cuda')
    check(lib().fp_score_features(sc.ctx.handle, sc.model.handle, ptr(part), N, ptr(f_p), stream_ptr()))
    assert torch.equal(f_p, feats[o * N:(o + 1) * N])


def test_bench_step_sharded_over_4_ranks_equals_one_rank():
  """bench.py's multi-GPU step with its ranks played one after the other on this GPU (the all-gather replaced by
  concatenation in rank order): 4 objects x 252 hypotheses over 4 ranks with the rotated shard assignment must finalise
  every object exactly like a single-rank job does - same argmax, same refined poses, bit for bit."""
  import bench
  dev = torch.device('cuda', 0)
  world = 4
  est, objects = bench.build_job(dev, n_objects=world, rank=0)
  est.refiner.ctx.reserve(bench.N_HYP)
  ref = {}
  for o in range(world):                                   # truth: each object as a one-rank job
    ref.update({o: v for v in [bench.step_finalize(est, objects[o:o + 1], 1, 0, bench.step_local(est, objects[o:o + 1], 1, 0))[0]]})
  rows = [bench.step_local(est, objects, world, r) for r in range(world)]
  gathered = torch.cat(rows, 0)
  seen = {}
  for r in range(world):
    seen.update(bench.step_finalize(est, objects, world, r, gathered))
  assert sorted(seen) == list(range(world))
  for o in range(world):
    assert int(seen[o][0][0]) == int(ref[o][0][0]) and torch.equal(seen[o][1], ref[o][1])

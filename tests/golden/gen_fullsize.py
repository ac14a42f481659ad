"""Generate tests/golden/fullsize.npz: the CPU ORACLE's results at BASELINE.json's full sizes, so that the GPU tests can
assert the acceptance criterion (refined 4x4 poses within 1e-3, identical argmax) where it is hardest: 252 hypotheses x
5 recurrent iterations (configs[1]), the four objects of configs[3], the 32-hypothesis cases of configs[0] and 10 frames
of configs[4] (track_one and the 64-hypothesis tracking mode).

Runs in the build container on the CPU only (oracle/ + foundationpose_amd.synthetic; it does NOT need /root/reference:
the oracle's networks are pinned to the reference's modules by tests/golden/gen_golden.py).  Cases are defined in
tests/cases.py, shared with the tests.

  python tests/golden/gen_fullsize.py feats     # heavy (~12 min on 8 cores): oracle refine + ScoreNet features per case,
                                                # cached in /tmp/fp_fullsize_feats.npz
  python tests/golden/gen_fullsize.py track     # configs[4] frames
  python tests/golden/gen_fullsize.py tail      # search the ScoreNet tail seed (att_cross + linear; the trunk and the
                                                # per-hypothesis features do not depend on it) that maximises the smallest
                                                # top-1 / top-2 logit margin over all cases; prints the ranking
  python tests/golden/gen_fullsize.py write     # fixture with synthetic.py's current default tail seed
  python tests/golden/gen_fullsize.py d16       # adds <case>/poses_iter_d16 (the refine chain with the HIP kernels' rounding points, oracle/nets.py refine_forward_d16)
  python tests/golden/gen_fullsize.py ac16      # adds <case>/poses_iter_ac16 (the refine chain with the network under CPU autocast fp16) to the fixture

The chosen tail seed is the default of foundationpose_amd.synthetic.make_score_state_dict; fullsize.npz records, per case,
the oracle's margin and logit spread, and the tests assert margin >= 20 x the fp16 logit noise they measure.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
CACHE = '/tmp/fp_fullsize_feats.npz'


def oracle_case(name, out):
  """Refined poses per iteration (hypothesis order) + fp32 ScoreNet features on the final poses."""
  from foundationpose_amd import synthetic as S
  from oracle import nets, predict as OP
  from tests import cases
  c = cases.case(name)
  sc = c['sc']
  rsd = S.make_refine_state_dict(**c['refine_sd_kw'])
  ssd = S.make_score_state_dict(**c['score_sd_kw'])
  rcfg = dict(OP.DEFAULT_REFINE_CFG, **{k: c['refine_cfg'][k] for k in ('normalize_xyz', 'rot_rep', 'use_BN', 'crop_ratio') if k in c['refine_cfg']})
  scfg = dict(OP.DEFAULT_SCORE_CFG, **{k: c['score_cfg'][k] for k in ('normalize_xyz', 'use_BN', 'crop_ratio') if k in c['score_cfg']})
  t0 = time.time()
  trace = []
  poses = OP.refine_predict(rcfg, rsd, sc['rgb'], c['depth'], sc['K'], c['poses0'], c['xyz_map'], sc['mt'], sc['diameter'],
                            iteration=c['iteration'], chunk=16, trace=trace)
  per_iter = [t['poseA'].numpy() for t in trace[1:]] + [poses.numpy()]
  rgb_t = torch.as_tensor(sc['rgb'], dtype=torch.float32)
  depth_t = torch.as_tensor(c['depth'], dtype=torch.float32)
  pd = OP.make_crop_data_batch_score(scfg, poses.numpy(), sc['mt'], rgb_t, depth_t, sc['K'], sc['diameter'])
  A = torch.cat([pd['rgbAs'], pd['xyz_mapAs']], dim=1).float()
  B = torch.cat([pd['rgbBs'], pd['xyz_mapBs']], dim=1).float()
  feats = torch.cat([nets.score_extract_feat(ssd, A[i:i + 16], B[i:i + 16], scfg['use_BN']) for i in range(0, len(A), 16)], 0)
  out[f'{name}/poses0'] = c['poses0']
  out[f'{name}/poses_iter'] = np.stack(per_iter).astype(np.float32)
  out[f'{name}/feats'] = feats.numpy()
  print(f'{name}: {len(c["poses0"])} hyp x {c["iteration"]} iterations + features in {time.time() - t0:.0f} s', flush=True)


N_TRK64 = 6


def oracle_tracking(out, n_frames=10):
  """configs[4]: (a) track_one, 1 hypothesis x 2 iterations per frame, chained; (b) 64-hypothesis mode, N_TRK64 frames: a
  start pose and 63 seeded perturbations of it, refine x2 + score.  The test gives both sides the same start pose per frame
  (teacher forcing), so a frame is judged on its own."""
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.tracking import tracking_hypotheses
  from oracle import geometry as G, nets, predict as OP
  from tests import cases
  sc, frames = cases.tracking_frames(n_frames)
  rsd, ssd = S.make_refine_state_dict(cases.REFINE_SEED, head_gain=cases.GAIN_CHAIN), S.make_score_state_dict(cases.SCORE_SEED)
  rcfg, scfg = dict(OP.DEFAULT_REFINE_CFG), dict(OP.DEFAULT_SCORE_CFG)
  t0 = time.time()
  start = torch.as_tensor(frames[0]['gt_pose']).clone()
  start[:3, 3] += torch.tensor([0.003, -0.002, 0.004])          # tracker starts slightly off the true pose
  out['trk/start'] = start.numpy()
  one, multi_in, multi_poses, multi_feats = [], [], [], []
  p1 = start.clone().reshape(1, 4, 4)
  p64 = start.clone()
  for fr in frames:
    depth = G.bilateral_filter_depth(G.erode_depth(fr['depth']))
    K32 = torch.as_tensor(np.asarray(fr['K']), dtype=torch.float32)
    xyz = G.depth2xyzmap_batch(torch.as_tensor(depth)[None], K32[None], zfar=np.inf)[0]
    p1 = OP.refine_predict(rcfg, rsd, fr['rgb'], depth, fr['K'], p1.numpy(), xyz, sc['mt'], sc['diameter'], iteration=2, chunk=16)
    one.append(p1[0].numpy())
    if len(multi_in) >= N_TRK64:
      continue
    hyp = tracking_hypotheses(p64, 64)
    multi_in.append(hyp.numpy())
    refined = OP.refine_predict(rcfg, rsd, fr['rgb'], depth, fr['K'], hyp.numpy(), xyz, sc['mt'], sc['diameter'], iteration=2, chunk=16)
    pd = OP.make_crop_data_batch_score(scfg, refined.numpy(), sc['mt'], torch.as_tensor(fr['rgb'], dtype=torch.float32),
                                       torch.as_tensor(depth), fr['K'], sc['diameter'])
    A = torch.cat([pd['rgbAs'], pd['xyz_mapAs']], dim=1).float()
    B = torch.cat([pd['rgbBs'], pd['xyz_mapBs']], dim=1).float()
    feats = torch.cat([nets.score_extract_feat(ssd, A[i:i + 16], B[i:i + 16], True) for i in range(0, 64, 16)], 0)
    p64 = refined[0]           # next frame starts from the refined UNPERTURBED hypothesis: the sequence (hence every frame's
                               # features) does not depend on the scorer's tail, which is chosen afterwards (stage 'tail')
    multi_poses.append(refined.numpy())
    multi_feats.append(feats.numpy())
  out['trk/one'] = np.stack(one).astype(np.float32)
  out['trk/multi_in'] = np.stack(multi_in).astype(np.float32)
  out['trk/multi_poses'] = np.stack(multi_poses).astype(np.float32)
  out['trk/multi_feats'] = np.stack(multi_feats).astype(np.float32)
  print(f'tracking: {n_frames} frames in {time.time() - t0:.0f} s', flush=True)


def logits_of(ssd, feats):
  """att_cross + linear on the oracle's fp32 features, evaluated in float64: the logits differ between hypotheses by 1e-4 .. 1e-3
  on top of O(1) common parts, so an fp32 evaluation carries summation noise of several 1e-6 - the fixture holds the exact values."""
  from oracle import nets
  sd64 = {k: v.double() for k, v in ssd.items() if k.startswith('att_cross.') or k.startswith('linear.')}
  return nets.score_tail(sd64, torch.as_tensor(feats).double(), len(feats)).reshape(-1).numpy()


def margin_stats(lg):
  o = np.sort(lg)[::-1]
  return float(o[0] - o[1]), float(lg.std())


def stage_feats():
  from tests import cases
  torch.set_num_threads(os.cpu_count())
  out = {}
  for name in cases.REGISTER_CASES:
    oracle_case(name, out)
  np.savez(CACHE, **out)
  print('cached', CACHE)


def stage_track():
  torch.set_num_threads(os.cpu_count())
  out = dict(np.load(CACHE))
  oracle_tracking(out)
  np.savez(CACHE, **out)


def stage_tail(n_seeds=6000):
  from foundationpose_amd import synthetic as S
  from tests import cases
  z = np.load(CACHE)
  names = cases.REGISTER_CASES
  groups = [z[f'{n}/feats'] for n in names] + [f for f in z['trk/multi_feats']]
  base = S.make_score_state_dict(cases.SCORE_SEED)
  res = []
  for seed in range(n_seeds):
    ssd = dict(base)
    ssd.update({k: v for k, v in S.make_score_state_dict(cases.SCORE_SEED, tail_seed=seed, tail_only=True).items()})
    worst = 1e9
    for f in groups:
      m, sp = margin_stats(logits_of(ssd, f))
      worst = min(worst, m / sp)
    res.append((worst, seed))
  res.sort(reverse=True)
  print('best (min over cases of margin / logit std, tail seed):', res[:10])


def stage_write():
  from foundationpose_amd import synthetic as S
  from tests import cases
  z = np.load(CACHE)
  ssd = S.make_score_state_dict(cases.SCORE_SEED)
  out = {}
  for name in cases.REGISTER_CASES:
    lg = logits_of(ssd, z[f'{name}/feats'])
    m, sp = margin_stats(lg)
    out[f'{name}/poses_iter'] = z[f'{name}/poses_iter']
    out[f'{name}/logits'] = lg.astype(np.float32)
    out[f'{name}/argmax'] = np.array(int(lg.argmax()))
    out[f'{name}/margin'] = np.array(m)
    out[f'{name}/spread'] = np.array(sp)
    print(f'{name}: argmax {int(lg.argmax())}, margin {m:.3e}, spread {sp:.3e}, margin/spread {m / sp:.2f}')
  out['c1/feats'] = z['c1/feats']
  for k in ('trk/start', 'trk/one', 'trk/multi_in', 'trk/multi_poses'):
    out[k] = z[k]
  lgs = np.stack([logits_of(ssd, f) for f in z['trk/multi_feats']])
  out['trk/multi_logits'] = lgs.astype(np.float32)
  ms = [margin_stats(l) for l in lgs]
  out['trk/multi_margin'] = np.array([m for m, _ in ms])
  print('tracking margins / spread:', [f'{m / s:.2f}' for m, s in ms], 'argmax', lgs.argmax(1))
  path = os.path.join(HERE, 'fullsize.npz')
  if os.path.exists(path):                       # (the autocast chains of stage 'ac16' do not depend on the tail seed: kept)
    old = np.load(path)
    out.update({k: old[k] for k in old.files if (k.endswith('_ac16') or k.endswith('_d16')) and k not in out})
  np.savez_compressed(path, **out)
  print('wrote', path, f'{os.path.getsize(path) / 1e6:.2f} MB')


AC16_CASES = ('c1', 'c3_1', 'c3_2', 'c3_3')


def stage_d16():
  """The oracle's refine chain with the PRODUCT'S numeric recipe (oracle/nets.py refine_forward_d16: fp16 roundings where the HIP kernels round,
  fp32 accumulation) through the same five full-gain iterations -> `<case>/poses_iter_d16`.  The GPU test compares the HIP chain with THIS
  chain on all 252 hypotheses: what separates them is the order of fp32 sums only, so agreement far inside the fp16-vs-fp32 scatter shows
  that the HIP path differs from the fp32 oracle by precision, not by logic."""
  _stage_chain('d16', dict(recipe='d16'))


def stage_ac16():
  _stage_chain('ac16', dict(autocast=True))


def _stage_chain(tag, kw):
  """The oracle's refine loop with the network in another precision through the same five full-gain iterations, for configs[1] and the three
  other objects of configs[3]; added to the existing fixture as `<case>/poses_iter_<tag>` (the other keys are left as they are).
  ac16: the reference's OWN precision - torch.autocast('cpu', float16), what predict_pose_refine.py:190 runs on the GPU; the full-gain chain
  test measures the HIP path's scatter against THIS chain's scatter around the fp32 chain.  d16: see stage_d16.  ~7 min per case on 8 cores."""
  from foundationpose_amd import synthetic as S
  from oracle import predict as OP
  from tests import cases
  torch.set_num_threads(int(os.environ.get('FP_GEN_THREADS', os.cpu_count())))
  path = os.path.join(HERE, 'fullsize.npz')
  out = dict(np.load(path))
  for name in AC16_CASES:
    c = cases.case(name)
    sc = c['sc']
    rsd = S.make_refine_state_dict(**c['refine_sd_kw'])
    rcfg = dict(OP.DEFAULT_REFINE_CFG, **{k: c['refine_cfg'][k] for k in ('normalize_xyz', 'rot_rep', 'use_BN', 'crop_ratio') if k in c['refine_cfg']})
    t0 = time.time()
    trace = []
    poses = OP.refine_predict(rcfg, rsd, sc['rgb'], c['depth'], sc['K'], c['poses0'], c['xyz_map'], sc['mt'], sc['diameter'],
                              iteration=c['iteration'], chunk=16, trace=trace, **kw)
    per_iter = np.stack([t['poseA'].numpy() for t in trace[1:]] + [poses.numpy()]).astype(np.float32)
    out[f'{name}/poses_iter_{tag}'] = per_iter
    d = np.abs(per_iter - out[f'{name}/poses_iter']).reshape(per_iter.shape[0], per_iter.shape[1], -1).max(2)
    print(f'{name}: {tag} chain in {time.time() - t0:.0f} s; |chain - fp32| per iteration: median ' + ' '.join(f'{np.median(x):.1e}' for x in d) +
          ' max ' + ' '.join(f'{x.max():.1e}' for x in d), flush=True)
  np.savez_compressed(path, **out)
  print('wrote', path, f'{os.path.getsize(path) / 1e6:.2f} MB')


if __name__ == '__main__':
  {'feats': stage_feats, 'track': stage_track, 'tail': stage_tail, 'write': stage_write, 'ac16': stage_ac16, 'd16': stage_d16}[sys.argv[1]]()

"""GPU box: ScoreNet features of the HIP path for every case of tests/cases.py -> gpurun_out/gpu_feats.npz:
  <case>/feats       on the ORACLE's refined poses (fixture): differs from the oracle's fp32 features by the fp16 noise of the
                     scorer alone
  <case>/feats_e2e   chained cases: on the GPU's OWN refined poses (adds what the <= 1e-3 pose differences do to the features)
  trk/multi_feats, trk/multi_feats_e2e   the same for the 64-hypothesis tracking frames
The ScoreNet tail (att_cross + linear) is applied offline: tests/golden/gen_fullsize.py tail --noise gpurun_out/gpu_feats.npz
ranks tail seeds by the worst margin / noise over all cases with the measured features."""
import os
import sys

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, REPO)
from tests import cases, util  # noqa: E402


def main():
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.predict_score import ScorePredictor
  from foundationpose_amd.tracking import tracking_hypotheses
  from foundationpose_amd.Utils import bilateral_filter_depth, depth2xyzmap_batch, erode_depth
  full = np.load(os.path.join(REPO, 'tests', 'golden', 'fullsize.npz'))
  scorer = ScorePredictor(state_dict=S.make_score_state_dict(cases.SCORE_SEED), cfg=SCORE_DEFAULT)
  refiners = {g: PoseRefinePredictor(state_dict=S.make_refine_state_dict(cases.REFINE_SEED, head_gain=g), cfg=REFINE_DEFAULT)
              for g in (cases.GAIN_STEP, cases.GAIN_CHAIN)}
  out = {}
  for name in cases.REGISTER_CASES:
    c = cases.case(name)
    sc = c['sc']
    mt = util.to_dev(sc['mt'])
    kw = dict(mesh_tensors=mt, mesh_diameter=sc['diameter'])
    out[f'{name}/feats'] = scorer.extract_features(sc['rgb'], c['depth'], sc['K'], full[f'{name}/poses_iter'][-1], **kw).cpu().numpy()
    if name not in cases.STEP_CASES:
      got, _ = refiners[c['refine_sd_kw']['head_gain']].predict(rgb=sc['rgb'], depth=c['depth'], K=sc['K'], ob_in_cams=c['poses0'],
                                                                 xyz_map=c['xyz_map'], iteration=c['iteration'], **kw)
      out[f'{name}/feats_e2e'] = scorer.extract_features(sc['rgb'], c['depth'], sc['K'], got, **kw).cpu().numpy()
  sc, frames = cases.tracking_frames(10)
  mt = util.to_dev(sc['mt'])
  kw = dict(mesh_tensors=mt, mesh_diameter=sc['diameter'])
  f_or, f_e2e = [], []
  for fr, poses, hyp in zip(frames, full['trk/multi_poses'], full['trk/multi_in']):
    depth = bilateral_filter_depth(erode_depth(torch.as_tensor(fr['depth']).cuda(), radius=2), radius=2)
    f_or.append(scorer.extract_features(fr['rgb'], depth, fr['K'], poses, **kw).cpu().numpy())
    xyz = depth2xyzmap_batch(depth[None], np.asarray(fr['K'], dtype=np.float32)[None], zfar=np.inf)[0]
    got, _ = refiners[cases.GAIN_CHAIN].predict(rgb=fr['rgb'], depth=depth, K=fr['K'], ob_in_cams=hyp, xyz_map=xyz, iteration=2, **kw)
    f_e2e.append(scorer.extract_features(fr['rgb'], depth, fr['K'], got, **kw).cpu().numpy())
  out['trk/multi_feats'], out['trk/multi_feats_e2e'] = np.stack(f_or), np.stack(f_e2e)
  os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
  np.savez(os.path.join(REPO, 'gpurun_out', 'gpu_feats.npz'), **out)
  print('wrote gpurun_out/gpu_feats.npz', sorted(out))


if __name__ == '__main__':
  main()

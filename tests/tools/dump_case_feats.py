"""GPU box: ScoreNet features of the HIP path on the ORACLE's refined poses for every case of tests/cases.py ->
gpurun_out/gpu_feats.npz.  Together with the oracle's fp32 features (tests/golden/gen_fullsize.py cache) this is the measured
fp16 feature noise the ScoreNet tail seed is chosen against (gen_fullsize.py tail --noise gpurun_out/gpu_feats.npz)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, REPO)
from tests import cases, util  # noqa: E402


def main():
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import SCORE_DEFAULT
  from foundationpose_amd.predict_score import ScorePredictor
  from foundationpose_amd.Utils import bilateral_filter_depth, erode_depth
  full = np.load(os.path.join(REPO, 'tests', 'golden', 'fullsize.npz'))
  scorer = ScorePredictor(state_dict=S.make_score_state_dict(cases.SCORE_SEED), cfg=SCORE_DEFAULT)
  out = {}
  for name in cases.REGISTER_CASES:
    c = cases.case(name)
    sc = c['sc']
    f = scorer.extract_features(sc['rgb'], c['depth'], sc['K'], full[f'{name}/poses_iter'][-1], mesh_tensors=util.to_dev(sc['mt']),
                                mesh_diameter=sc['diameter'])
    out[f'{name}/feats'] = f.cpu().numpy()
  sc, frames = cases.tracking_frames(len(full['trk/multi_poses']))
  mt = util.to_dev(sc['mt'])
  tf = []
  for fr, poses in zip(frames, full['trk/multi_poses']):
    depth = bilateral_filter_depth(erode_depth(fr['depth'], radius=2), radius=2)
    tf.append(scorer.extract_features(fr['rgb'], depth, fr['K'], poses, mesh_tensors=mt, mesh_diameter=sc['diameter']).cpu().numpy())
  out['trk/multi_feats'] = np.stack(tf)
  os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
  np.savez(os.path.join(REPO, 'gpurun_out', 'gpu_feats.npz'), **out)
  print('wrote gpurun_out/gpu_feats.npz')


if __name__ == '__main__':
  main()

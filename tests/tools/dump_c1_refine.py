"""GPU box: c1 refinement, chained (k = 1..5 iterations from the start hypotheses) and one-step (one iteration from the
oracle's poses after k-1 iterations) -> gpurun_out/c1_refine.npz, for offline analysis against tests/golden/fullsize.npz."""
import os
import sys

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, REPO)
from tests import cases, util  # noqa: E402


def main():
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.Utils import compute_crop_window_tf_batch
  full = np.load(os.path.join(REPO, 'tests', 'golden', 'fullsize.npz'))
  gain = float(os.environ.get('HEAD_GAIN', '0'))
  rsd = S.make_refine_state_dict(cases.REFINE_SEED) if gain == 0 else S.make_refine_state_dict(cases.REFINE_SEED, head_gain=gain)
  refiner = PoseRefinePredictor(state_dict=rsd, cfg=REFINE_DEFAULT)
  c = cases.case('c1')
  sc = c['sc']
  kw = dict(rgb=sc['rgb'], depth=c['depth'], K=sc['K'], mesh_tensors=util.to_dev(sc['mt']), mesh_diameter=sc['diameter'], xyz_map=c['xyz_map'])
  want = full['c1/poses_iter']
  out = {}
  chained, onestep, raw = [], [], []
  for it in range(1, 6):
    p, _ = refiner.predict(ob_in_cams=c['poses0'], iteration=it, **kw)
    chained.append(p.cpu().numpy())
    start = c['poses0'] if it == 1 else want[it - 2]
    q, _ = refiner.predict(ob_in_cams=start, iteration=1, **kw)
    onestep.append(q.cpu().numpy())
    raw.append(np.concatenate([refiner.last_trans_update.cpu().numpy(), refiner.last_rot_update.cpu().numpy()], 1))
  out['chained'], out['onestep'], out['raw_onestep'] = np.stack(chained), np.stack(onestep), np.stack(raw)
  out['tf_gpu_chain'] = np.stack([compute_crop_window_tf_batch(poses=p, K=sc['K'], crop_ratio=1.2, out_size=(160, 160), method='box_3d',
                                                               mesh_diameter=sc['diameter']).cpu().numpy() for p in [c['poses0']] + chained[:-1]])
  out['tf_oracle_chain'] = np.stack([compute_crop_window_tf_batch(poses=p, K=sc['K'], crop_ratio=1.2, out_size=(160, 160), method='box_3d',
                                                                  mesh_diameter=sc['diameter']).cpu().numpy() for p in [c['poses0']] + list(want[:-1])])
  np.savez(os.path.join(REPO, 'gpurun_out', f'c1_refine_g{gain:g}.npz'), **out)
  for it in range(5):
    e_c = np.abs(out['chained'][it] - want[it]).reshape(252, -1).max(1)
    e_1 = np.abs(out['onestep'][it] - want[it]).reshape(252, -1).max(1)
    flips = (np.abs(out['tf_gpu_chain'][it] - out['tf_oracle_chain'][it]).reshape(252, -1).max(1) > 1e-6).sum()
    print(f'iteration {it + 1}: chained err median {np.median(e_c):.2e} p90 {np.quantile(e_c, .9):.2e} max {e_c.max():.2e} | one-step err median '
          f'{np.median(e_1):.2e} p90 {np.quantile(e_1, .9):.2e} max {e_1.max():.2e} | crop windows that differ (chained) {flips}')


if __name__ == '__main__':
  main()

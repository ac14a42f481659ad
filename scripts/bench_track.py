"""configs[4]: tracking mode - synthetic sequence, track_one() per frame (reference: 1 hypothesis,
src/estimater.py:250-268) and the 64-hypothesis variant (refine 64 perturbed poses per frame, keep the
best-scoring one).  Prints steady-state frames/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from foundationpose_amd import synthetic as S
from foundationpose_amd.Utils import nvdiffrast_render
from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
from foundationpose_amd.estimater import FoundationPose
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.predict_score import ScorePredictor


def main():
  n_frames = int(os.environ.get('FRAMES', '200'))
  mesh = S.make_mustard_mesh(seed=0)
  refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT)
  scorer = ScorePredictor(state_dict=S.make_score_state_dict(1), cfg=SCORE_DEFAULT)
  np.random.seed(0)
  est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, refiner=refiner, scorer=scorer)
  mt = est.mesh_tensors

  def rf(K, H, W, pose):
    c, d, _ = nvdiffrast_render(K=K, H=H, W=W, ob_in_cams=torch.as_tensor(pose, device='cuda'), mesh_tensors=mt, use_light=True)
    return c[0].cpu().numpy(), d[0].cpu().numpy()
  sc = S.make_scene(rf, mt, seed=0)
  est.rot_grid = est.rot_grid[:32].contiguous()
  est.register(K=sc['K'], rgb=sc['rgb'], depth=sc['depth'], ob_mask=sc['mask'], iteration=2)
  rgb = torch.as_tensor(sc['rgb'], device='cuda', dtype=torch.float)
  depth = torch.as_tensor(sc['depth'], device='cuda')
  for it in (2,):
    for _ in range(5):
      est.track_one(rgb=rgb, depth=depth, K=sc['K'], iteration=it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_frames):
      est.track_one(rgb=rgb, depth=depth, K=sc['K'], iteration=it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'track_one (1 hypothesis, iteration={it}): {n_frames / dt:.1f} frames/s  ({dt / n_frames * 1e3:.2f} ms/frame)')
  # 64-hypothesis tracking: perturb the last pose, refine all, keep the best-scoring
  from foundationpose_amd.Utils import bilateral_filter_depth, depth2xyzmap_batch, erode_depth
  g = torch.Generator(device='cuda').manual_seed(0)
  def track64():
    d = bilateral_filter_depth(erode_depth(depth, radius=2), radius=2)
    xyz = depth2xyzmap_batch(d[None], torch.as_tensor(sc['K'], dtype=torch.float, device='cuda')[None], zfar=np.inf)[0]
    base = est.pose_last.reshape(4, 4)
    hyp = base[None].repeat(64, 1, 1)
    hyp[1:, :3, 3] += torch.randn((63, 3), device='cuda', generator=g) * 0.003
    refined, _ = refiner.predict(rgb=rgb, depth=d, K=sc['K'], ob_in_cams=hyp, xyz_map=xyz, mesh_tensors=mt, mesh_diameter=est.diameter, iteration=2)
    scores, _ = scorer.predict(rgb=rgb, depth=d, K=sc['K'], ob_in_cams=refined, mesh_tensors=mt, mesh_diameter=est.diameter)
    est.pose_last = refined[scores.argmax()]
  for _ in range(3):
    track64()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  n2 = max(20, n_frames // 4)
  for _ in range(n2):
    track64()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  print(f'64-hypothesis tracking (refine x2 + score): {n2 / dt:.1f} frames/s  ({dt / n2 * 1e3:.2f} ms/frame)')


if __name__ == '__main__':
  main()

"""Wall time of the whole FoundationPose.register() call (numpy frame in, numpy pose out: upload, depth filtering, validity /
guess_translation reductions, back-projection, 252 hypotheses x (5 refine + score), sort, download) beside bench.py's core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
  dev = torch.device('cuda', 0)
  est, objects = bench.build_job(dev, n_objects=1, rank=0)
  ob = objects[0]
  est.refiner.ctx.reserve(bench.N_HYP)
  for _ in range(2):
    est.register(K=ob['K'], rgb=ob['rgb_np'], depth=ob['depth_np'], ob_mask=ob['mask'], iteration=5)
  torch.cuda.synchronize()
  n = 10
  t0 = time.perf_counter()
  for _ in range(n):
    pose = est.register(K=ob['K'], rgb=ob['rgb_np'], depth=ob['depth_np'], ob_mask=ob['mask'], iteration=5)
  dt = (time.perf_counter() - t0) / n
  print(f'register(): {dt * 1e3:.2f} ms per call  ({bench.N_HYP / dt:.0f} hypotheses/s through the public API)')


if __name__ == '__main__':
  main()

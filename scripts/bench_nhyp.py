"""Local step (refine x5 + score features, one object) for a batch of n hypotheses: python scripts/bench_nhyp.py 32 48 63 64 96 126"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(bench.N_HYP)
ob = objects[0]
for n in [int(a) for a in sys.argv[1:]] or [32, 63, 126, 252]:
  def one():
    refined = est.refiner.predict_multi([dict(rgb=ob['rgb'], xyz_map=ob['xyz'], K=ob['K'], mesh_tensors=est.mesh_tensors, mesh_diameter=est.diameter,
                                              ob_in_cams=ob['poses'][:n])], iteration=bench.ITER)
    return est.scorer.extract_features_multi([dict(rgb=ob['rgb'], depth=ob['depth'], K=ob['K'], mesh_tensors=est.mesh_tensors, mesh_diameter=est.diameter,
                                                   ob_in_cams=refined)])
  for _ in range(3): one()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(10): one()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 10
  print(f'n {n:4d}: {dt * 1e3:7.3f} ms/step  {dt / n * 1e3:.4f} ms per hypothesis   (FP_TRUNK_MIN={os.environ.get("FP_TRUNK_MIN", "default")})', flush=True)

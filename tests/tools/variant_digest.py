"""Run under an environment that selects kernel variants / schedules (the FP_* knobs are read once per process): the fused passes
(refine x2 + ScoreNet features, one object) at batch sizes that reach the schedules of the product path - 1 and 2 hypotheses (split-K 3x3
layers, tracking), 8 and 40 (the two sides of encodeA as two chains), 56 (trunk cut in two by hypotheses from 48 on), 100 (band kernel for the 128 -> 128 layers) - and a
sha256 of the refined poses and features per size, as one JSON line.  tests/test_gpu_pipeline.py compares the lines of several
environments: schedules and kernel forms that claim bit-identical results must print the same digests."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
import torch
import bench

dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(128)
ob = objects[0]
out = {}
for n in (1, 2, 8, 40, 56, 100):
  refined = est.refiner.predict_multi([dict(rgb=ob['rgb'], xyz_map=ob['xyz'], K=ob['K'], mesh_tensors=est.mesh_tensors, mesh_diameter=est.diameter,
                                            ob_in_cams=ob['poses'][:n], shared_translation=True)], iteration=2)      # (the hypotheses of a registration: FP_NO_SHARED_B=1 must not change a bit)
  feats = est.scorer.extract_features_multi([dict(rgb=ob['rgb'], depth=ob['depth'], K=ob['K'], mesh_tensors=est.mesh_tensors, mesh_diameter=est.diameter,
                                                  ob_in_cams=refined)])
  torch.cuda.synchronize()
  assert bool(torch.isfinite(refined).all()) and bool(torch.isfinite(feats).all())
  out[str(n)] = hashlib.sha256(refined.cpu().numpy().tobytes() + feats.cpu().numpy().tobytes()).hexdigest()[:16]
print('DIGEST ' + json.dumps(out))

"""Where does the host block inside a bench step?  Host time of each part of a step, no synchronisation added."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(252)
for _ in range(3):
  bench.step(est, objects[:1], 1, 0, replicate_tail=True)
torch.cuda.synchronize()
import foundationpose_amd.predict_pose_refine as R, foundationpose_amd.predict_score as S
T = {}
def wrap(obj, name):
  f = getattr(obj, name)
  def g(*a, **k):
    t = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t); return r
  setattr(obj, name, g)
wrap(est.refiner, 'predict_multi'); wrap(est.scorer, 'extract_features_multi'); wrap(est.scorer, 'score_tail')
wrap(bench, 'step_local'); wrap(bench, 'step_finalize')
t0 = time.perf_counter()
for _ in range(10):
  t = time.perf_counter(); bench.step(est, objects[:1], 1, 0, replicate_tail=True); T.setdefault('step', []).append(time.perf_counter() - t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'10 steps: host loop {1e3 * (t1 - t0):.1f} ms, + final sync {1e3 * (t2 - t1):.1f} ms')
for k, v in T.items():
  print(f'{k:28s} ' + ' '.join(f'{1e3 * x:7.2f}' for x in v))

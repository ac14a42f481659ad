"""Runs the per-rank part of a configs[2] step (bench.step_local: rank 0's shard of ONE object) a few times: under
`rocprofv3 --kernel-trace` (scripts/prof_shard.sh) this gives the per-kernel times of a 126- / 63- / 32-hypothesis shard.
WORLD=n: the shard of an n-rank job (default 8), STEPS=k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda', 0)
world = int(os.environ.get('WORLD', '8'))
steps = int(os.environ.get('STEPS', '10'))
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(bench.N_HYP)
one = lambda: bench.step_local(est, objects, world, 0)
for _ in range(3):
  one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
  one()
torch.cuda.synchronize()
print(f'world {world}: {-(-bench.N_HYP // world)} hypotheses, {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step local')

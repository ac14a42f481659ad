"""Per-rank GPU work of `bench.py --gpus W`, emulated on ONE GPU without a process group: W objects, this rank's
(rotated) shard of each through predict_multi / extract_features_multi (the all-gather and the 0.66-GFLOP
tail are left out).  Weak scaling holds if the time does not grow with W.  Then the headline layout of bench.py (configs[2]): ONE
object whose 252 hypotheses are cut into W shards - the local time of rank 0's shard (the longest) bounds the strong-scaling step.
usage: python scripts/bench_rankload.py 1 2 4 8"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
  device = torch.device('cuda', 0)
  worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
  est, objects_all = bench.build_job(device, n_objects=max(worlds), rank=0)
  est.refiner.ctx.reserve(bench.N_HYP + 8)
  for W in worlds:
    objects = objects_all[:W]

    def step():
      return bench.step_local(est, objects, W, 0)           # rank 0's (rotated) shard of every object, as in bench.step
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(3):
      h0 = time.perf_counter()
      step()
      host += time.perf_counter() - h0          # time the host needs to ENQUEUE a step (the calls are asynchronous)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ctx = est.refiner.ctx
    ctx.prof_reset(); ctx.prof_enable(2)
    step(); torch.cuda.synchronize()
    ctx.prof_enable(0)
    cls = {c: ctx.prof_read(c) for c in ('conv3x3_halo', 'conv3x3_s2', 'conv7x7', 'linear', 'attention', 'render')}
    print('   per-class ms (HIP events; overlapping launches on side streams each count their own span): '
          + '  '.join(f"{c} {v['total_ms']:.2f}/{v['launches']}" for c, v in cls.items()))
    print(f'world {W}: {W} objects, 252 hypotheses on this rank: {dt * 1e3:.2f} ms/step  ({bench.N_HYP / dt:.0f} hyp/s per GPU; '
          f'host enqueue {host / 3 * 1e3:.2f} ms/step)')
  print('configs[2] layout: ONE object, rank 0 holds ceil(252 / W) hypotheses')
  for W in worlds:
    one = lambda: bench.step_local(est, objects_all[:1], W, 0)
    one(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
      one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    n = -(-bench.N_HYP // W)
    ctx = est.refiner.ctx
    ctx.prof_reset(); ctx.prof_enable(2)
    one(); torch.cuda.synchronize()
    ctx.prof_enable(0)
    cls = {c: ctx.prof_read(c) for c in ('conv3x3_halo', 'conv3x3_s2', 'conv7x7', 'linear', 'attention', 'heads_wall', 'render')}
    print('   per-class ms: ' + '  '.join(f"{c} {v['total_ms']:.2f}/{v['launches']}" for c, v in cls.items()))
    print(f'world {W}: {n} hypotheses on this rank: {dt * 1e3:.2f} ms/step local -> at most {bench.N_HYP / dt:.0f} hyp/s for the job '
          f'(x{(bench.N_HYP / dt) / 1.0:.0f}); per-hypothesis cost x{dt / n / 1.0 * 1e3:.3f} ms')


if __name__ == '__main__':
  main()

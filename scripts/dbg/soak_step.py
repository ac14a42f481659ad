"""Soak: the bench step (252 hypotheses, two-stream trunk and heads) STEPS times back to back without synchronising in between: every
step's refined poses, features-derived argmax and logits must be bit-identical (a race between the streams would show as a differing step)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda', 0)
n_obj = int(os.environ.get('OBJECTS', '1'))
est, objects = bench.build_job(dev, n_objects=n_obj, rank=0)
est.refiner.ctx.reserve(252 * n_obj)
steps = int(os.environ.get('STEPS', '40'))
outs = []
for _ in range(steps):
  res = bench.step(est, objects[:n_obj], 1, 0, replicate_tail=True)
  outs.append([(am.clone(), poses.clone()) for am, poses in res.values()])
torch.cuda.synchronize()
dig = [hashlib.sha256(b''.join(am.cpu().numpy().tobytes() + p.cpu().numpy().tobytes() for am, p in o)).hexdigest()[:16] for o in outs]
print(f'{steps} steps x {n_obj} object(s): {len(set(dig))} distinct digest(s): {sorted(set(dig))}')
assert len(set(dig)) == 1
print('soak ok')

"""CPU, world_size 2 over gloo: the hypothesis-sharding + single all-gather exchange of
foundationpose_amd/dist.py (the N>1 path of bench.py / FoundationPose(dist_group=...))."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_shard_ranges_cover_exactly():
  from foundationpose_amd.dist import shard_ranges
  for n in (0, 1, 7, 32, 64, 252, 1008):
    for world in (1, 2, 3, 4, 8):
      r = shard_ranges(n, world)
      assert len(r) == world
      assert r[0][0] == 0 and r[-1][1] == n
      assert all(a <= b for a, b in r) and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
      sizes = [b - a for a, b in r]
      assert max(sizes) == (-(-n // world) if n else 0)     # 252/8 -> 32 x7 + 28


def test_pack_unpack_roundtrip_single_process():
  """Emulates the all-gather for 8 ranks in one process: concatenating every rank's padded block and
  unpacking must give back the rows in hypothesis order (252 is not divisible by 8)."""
  from foundationpose_amd.dist import pack_rows, shard_ranges, unpack_rows
  n, world = 252, 8
  g = torch.Generator().manual_seed(0)
  feats, poses = torch.randn((n, 512), generator=g), torch.randn((n, 4, 4), generator=g)
  shard = -(-n // world)
  blocks = [pack_rows(feats[a:b], poses[a:b], shard) for a, b in shard_ranges(n, world)]
  f2, p2 = unpack_rows(torch.cat(blocks, 0), n, world)
  assert torch.equal(f2, feats) and torch.equal(p2, poses)


def test_rotated_shards_balance_8_ranks():
  """8 objects x 252 hypotheses over 8 ranks: with the shard index rotated by the object index every rank handles exactly
  252 hypotheses (contiguous assignment: 256 on seven ranks, 224 on the last), every shard of every object is handled
  exactly once, and gather_order inverts the rotation."""
  from foundationpose_amd.dist import gather_order, rotated_shard, shard_ranges
  n, world = 252, 8
  ranges = shard_ranges(n, world)
  for rank in range(world):
    assert sum(b - a for a, b in (ranges[rotated_shard(o, rank, world)] for o in range(world))) == n
  for o in range(world):
    assert sorted(rotated_shard(o, r, world) for r in range(world)) == list(range(world))
    assert all(rotated_shard(o, gather_order(o, world)[s], world) == s for s in range(world))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, cases, out_q):
  sys.path.insert(0, REPO)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from foundationpose_amd.dist import all_gather_rows, gather_order, pack_rows, rotated_shard, shard_ranges, unpack_rows
  for n, n_obj in cases:
    shard = -(-n // world)
    g = torch.Generator().manual_seed(123)            # every rank can regenerate the full truth
    feats = torch.randn((n_obj, n, 512), generator=g)
    poses = torch.randn((n_obj, n, 4, 4), generator=g)
    ranges = shard_ranges(n, world)
    # bench.py layout: this rank's (rotated) shard of every object, stacked, ONE all-gather
    sl = [ranges[rotated_shard(o, rank, world)] for o in range(n_obj)]
    rows = torch.cat([pack_rows(feats[o, a:b], poses[o, a:b], shard) for o, (a, b) in enumerate(sl)], 0)
    gathered = all_gather_rows(rows).reshape(world, n_obj, shard, -1)
    ok = True
    for o in range(n_obj):
      f2, p2 = unpack_rows(gathered[gather_order(o, world), o].reshape(world * shard, -1), n, world)
      ok = ok and torch.equal(f2, feats[o]) and torch.equal(p2, poses[o])
    # the cross-hypothesis tail sees identical inputs on every rank -> identical argmax
    am = int((gathered[..., :512].sum(-1)).reshape(-1).argmax())
    out_q.put((rank, n, ok, am))
  dist.barrier()
  dist.destroy_process_group()


def test_all_gather_exchange_world2():
  world = 2
  cases = [(252, 2), (7, 1), (1, 1)]     # 252 hypotheses x 2 objects; ragged; fewer hypotheses than ranks
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=180) for _ in range(world * len(cases))]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(ok for _, _, ok, _ in res)
  for n, _ in cases:
    assert len({am for _, nn, _, am in res if nn == n}) == 1


class _FakePredictors:
  """CPU stand-ins with the call signatures bench.step() uses: per-hypothesis 'refinement' and 'features' that depend only on
  the hypothesis' own pose, and a cross-hypothesis tail that needs ALL hypotheses of an object (softmax-weighted mean)."""

  def predict_multi(self, objs, iteration=5):
    return torch.cat([torch.as_tensor(o['ob_in_cams']) * (1 + 0.01 * iteration) + 0.5 for o in objs], 0)

  def extract_features_multi(self, objs):
    poses = torch.cat([torch.as_tensor(o['ob_in_cams']) for o in objs], 0).reshape(-1, 16)
    w = torch.linspace(-1, 1, 16 * 512).reshape(16, 512)
    return torch.tanh(poses @ w)

  def extract_rows_multi(self, objs):
    """[feature 512 | pose 16] rows, as ScorePredictor.extract_rows_multi writes them"""
    poses = torch.cat([torch.as_tensor(o['ob_in_cams']) for o in objs], 0).reshape(-1, 16)
    return torch.cat((self.extract_features_multi(objs), poses), 1)

  def score_tail(self, feats, L=None):
    groups = feats.shape[0] // L
    f = feats[:, :512].reshape(groups, L, 512)         # (rows of 528 are read in place, like the library's tail)
    att = torch.softmax(f @ f.transpose(1, 2) / 512 ** 0.5, -1) @ f
    logits = att.sum(-1)
    return logits, logits.argmax(-1).to(torch.int32)


def _bench_worker(rank, world, port, out_q):
  sys.path.insert(0, REPO)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import types
  import bench
  fake = _FakePredictors()
  est = types.SimpleNamespace(refiner=fake, scorer=fake, mesh_tensors=None, diameter=0.2)
  g = torch.Generator().manual_seed(5)
  objects = [dict(rgb=None, xyz=None, depth=None, K=None, poses=torch.randn((bench.N_HYP, 4, 4), generator=g)) for _ in range(3)]
  # configs[2]: ONE object sharded over the ranks, tail on every rank
  single = bench.step(est, objects[:1], world, rank, replicate_tail=True)
  # the same object unsharded (what one GPU computes)
  truth = bench.step(est, objects[:1], 1, 0)
  ok_single = sorted(single) == [0] and torch.equal(single[0][1], truth[0][1]) and int(single[0][0][0]) == int(truth[0][0][0])
  # 3 objects over the ranks (weak-scaling / configs[3] layout): each object finalised on rank o % world
  multi = bench.step(est, objects, world, rank)
  ok_multi = sorted(multi) == [o for o in range(3) if o % world == rank]
  for o, (am, poses) in multi.items():
    t = bench.step(est, objects[o:o + 1], 1, 0)[0]
    ok_multi = ok_multi and torch.equal(poses, t[1]) and int(am[0]) == int(t[0][0])
  out_q.put((rank, ok_single, ok_multi, int(single[0][0][0])))
  dist.barrier()
  dist.destroy_process_group()


def test_bench_step_layouts_world2():
  """bench.py's step() over 2 gloo ranks with CPU stand-ins for the predictors: the single-object layout of configs[2] (252
  hypotheses cut into 2 shards, one all-gather, tail replicated: every rank ends with the SAME argmax and the unsharded poses)
  and the multi-object layout (rotated shards, object o finalised on rank o % world)."""
  world = 2
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=180) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(a and b for _, a, b, _ in res)
  assert len({am for _, _, _, am in res}) == 1


def _run_bench(args, env_extra=None, timeout=180):
  import subprocess
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'FP_BENCH_REHEARSAL')}
  env.update(env_extra or {})
  return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus 2` without a rendezvous in the environment must start 2 ranks itself (a child torch.distributed.run,
  never an exec) and relay rank 0's JSON line; --launch-check stops after the rendezvous + one all-gather of the rank ids, which
  on this GPU-less container runs over gloo."""
  import json
  r = _run_bench(['--gpus', '2', '--launch-check'])
  assert r.returncode == 0, r.stderr[-2000:]
  line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
  assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['ranks'] == [0, 1]
  assert 'torch.distributed.run' in r.stderr          # the launcher says what it started


def test_bench_refuses_a_mislabelled_run():
  """WORLD_SIZE != --gpus, or --gpus N on a node with fewer GPUs, must fail loudly instead of printing n_gpus: 1."""
  r = _run_bench(['--gpus', '2', '--launch-check'], {'WORLD_SIZE': '3', 'RANK': '0'})
  assert r.returncode != 0 and 'WORLD_SIZE=3' in r.stderr
  if torch.cuda.device_count() < 8:
    r = _run_bench(['--gpus', '8'])
    assert r.returncode != 0 and 'refusing' in r.stderr
    assert not any(l.startswith('{') for l in r.stdout.splitlines())

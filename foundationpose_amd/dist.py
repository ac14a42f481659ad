"""Hypothesis-parallel register over one node: one process per GPU, torch.distributed (backend
'nccl' = RCCL over xGMI on ROCm; 'gloo' on CPU for the tests).

The reference has no multi-GPU code.  The path shards over the hypothesis axis:
refinement and ScoreNet.extract_feat are per-hypothesis; only att_cross + linear + argmax couple
the hypotheses of one object (score_network.py:83-88).  So each object's hypotheses are split into
contiguous shards over ALL ranks, every rank refines and featurises its shards, ONE all-gather moves
[feat(512) | pose(16)] fp32 rows (2112 B / hypothesis, latency-bound: 532 KB at 252 hypotheses), and
the tiny cross-hypothesis tail runs where the object is finalised.  No other collective exists.
"""
import math

import torch
import torch.distributed as dist

ROW = 512 + 16


def shard_ranges(n, world):
  """Contiguous shards of ceil(n/world): [(start, stop)] per rank (the last ones may be short or empty)."""
  s = math.ceil(n / world) if n > 0 else 0
  return [(min(r * s, n), min((r + 1) * s, n)) for r in range(world)]


def rotated_shard(obj, rank, world):
  """Shard index of object `obj` that `rank` processes in a multi-object job.  Contiguous shards of ceil(n/world) leave
  the last shard short (252 = 7 x 32 + 28); rotating the assignment by the object index gives every rank of an
  8-object job exactly 252 hypotheses instead of 256 on seven ranks and 224 on the last."""
  return (rank + obj) % world


def gather_order(obj, world):
  """Rank that holds shard s of object `obj` (inverse of rotated_shard), for s = 0..world-1."""
  return [(s - obj) % world for s in range(world)]


_ORDER_INDEX = {}


def gather_order_index(obj, world, device):
  """gather_order as an index tensor on `device`, built once: indexing with a Python list uploads an index tensor on every call, and
  that host-to-device copy makes the host wait for the stream (the GPU then idles between the small launches that follow)."""
  key = (obj % world, world, str(device))
  if key not in _ORDER_INDEX:
    _ORDER_INDEX[key] = torch.as_tensor(gather_order(obj, world), dtype=torch.long, device=device)
  return _ORDER_INDEX[key]


def pack_rows(feats, poses, shard_size):
  """(k,512) feats + (k,4,4) poses -> (shard_size, 528) rows, zero padded (k <= shard_size)."""
  k = feats.shape[0]
  if k == shard_size:                      # a full shard: one copy, no padding pass
    return torch.cat((feats, poses.reshape(k, 16)), 1)
  rows = torch.zeros((shard_size, ROW), dtype=torch.float32, device=feats.device)
  if k:
    rows[:k, :512] = feats
    rows[:k, 512:] = poses.reshape(k, 16)
  return rows


def unpack_rows(gathered, n, world):
  """(world*shard_size, 528) all-gathered rows -> feats (n,512), poses (n,4,4) in hypothesis order."""
  shard = gathered.shape[0] // world
  if n == world * shard:                   # every shard full: the gathered block is the row list
    rows = gathered
  else:
    rows = torch.cat([gathered[r * shard:r * shard + (b - a)] for r, (a, b) in enumerate(shard_ranges(n, world))], 0)
  return rows[:, :512].contiguous(), rows[:, 512:].reshape(-1, 4, 4).contiguous()


def all_gather_rows(rows, group=None):
  """One all-gather of equally sized row blocks.  rows: (R, 528) -> (world*R, 528)."""
  world = dist.get_world_size(group)
  out = torch.empty((world * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
  dist.all_gather_into_tensor(out, rows.contiguous(), group=group)
  return out


def sharded_refine_and_score(est, K, rgb, depth, xyz_map, poses, iteration, group=None):
  """The refine + score section of FoundationPose.register (src/estimater.py:215-219) for ONE object,
  hypothesis-sharded over the ranks of `group`.  Every rank passes the same inputs and gets the same
  (poses (N,4,4), scores (N,)) back."""
  group = group if group is not None else est.dist_group
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  n = len(poses)
  a, b = shard_ranges(n, world)[rank]
  shard = math.ceil(n / world)
  mine = poses[a:b]
  if b > a:
    refined, _ = est.refiner.predict(mesh=est.mesh, mesh_tensors=est.mesh_tensors, rgb=rgb, depth=depth, K=K,
                                     ob_in_cams=mine, xyz_map=xyz_map, glctx=est.glctx,
                                     mesh_diameter=est.diameter, iteration=iteration, shared_translation=True)     # (a shard of register's rotation grid: one centre)
    feats = est.scorer.extract_features(rgb, depth, K, refined, mesh=est.mesh, mesh_tensors=est.mesh_tensors, glctx=est.glctx,
                                        mesh_diameter=est.diameter)
  else:
    refined = poses[:0]
    feats = torch.zeros((0, 512), dtype=torch.float32, device=poses.device)
  gathered = all_gather_rows(pack_rows(feats, refined, shard), group)
  feats_all, poses_all = unpack_rows(gathered, n, world)
  _, _, scores = est.scorer.score_tail(feats_all, L=n, score_offset=100.0)      # replicated on every rank: cheaper than a broadcast
  return poses_all, scores.reshape(-1)

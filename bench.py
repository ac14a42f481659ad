"""Benchmark: pose-hypotheses/sec of the render-and-compare hot path (render + refine x5 + score +
argmax) on 252 hypotheses x 160x160 crops per object - BASELINE.json's metric on configs[1].

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one register-core pass: est_refine_iter=5 x (crop window, render,
observed crop, RefineNet, pose update) + 1 x (crop window, render, observed crop, ScoreNet features)
+ cross-hypothesis tail + argmax, with the RGB-D frame, mesh, weights and hypotheses already resident
in HBM.  With N GPUs the job is N objects x 252 hypotheses (weak scaling); every object's hypotheses
are sharded over all ranks and one RCCL all-gather of [feature|pose] rows precedes the per-object
tails (foundationpose_amd/dist.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_HYP = 252
ITER = 5
PEAK_F16_TFLOPS = 2500.0     # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


def build_job(device, n_objects, rank):
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.Utils import nvdiffrast_render
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  from foundationpose_amd.estimater import FoundationPose
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
  from foundationpose_amd.predict_score import ScorePredictor
  from foundationpose_amd.Utils import bilateral_filter_depth, depth2xyzmap, erode_depth
  mesh = S.make_mustard_mesh(seed=0)
  refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT, device=device)
  scorer = ScorePredictor(state_dict=S.make_score_state_dict(1), cfg=SCORE_DEFAULT, device=device)
  np.random.seed(0)
  est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, refiner=refiner, scorer=scorer)
  assert est.rot_grid.shape[0] == N_HYP
  mt = est.mesh_tensors

  def rf(K, H, W, pose):
    c, d, _ = nvdiffrast_render(K=K, H=H, W=W, ob_in_cams=torch.as_tensor(pose, device=device), mesh_tensors=mt, use_light=True)
    return c[0].cpu().numpy(), d[0].cpu().numpy()
  objects = []
  for o in range(n_objects):
    sc = S.make_scene(rf, mt, seed=o)
    depth = bilateral_filter_depth(erode_depth(sc['depth'], radius=2), radius=2)
    center = est.guess_translation(depth=depth, mask=sc['mask'], K=sc['K'])
    poses = est.rot_grid.clone()
    poses[:, :3, 3] = torch.as_tensor(center, device=device, dtype=torch.float)
    objects.append(dict(K=sc['K'], rgb=torch.as_tensor(sc['rgb'], device=device, dtype=torch.float).contiguous(),
                        depth=torch.as_tensor(depth, device=device, dtype=torch.float).contiguous(),
                        xyz=torch.as_tensor(depth2xyzmap(depth, sc['K']), device=device).contiguous(), poses=poses,
                        rgb_np=sc['rgb'], depth_np=depth, mask=sc['mask']))
  return est, objects


def step_local(est, objects, world, rank):
  """The per-rank part of a step: this rank's shard of EVERY object through refinement and feature extraction as one batch.
  Returns the [feat | pose] row block this rank contributes to the all-gather, (n_objects * shard, 528)."""
  from foundationpose_amd.dist import pack_rows, rotated_shard, shard_ranges
  shard = math.ceil(N_HYP / world)
  ranges = shard_ranges(N_HYP, world)
  # the shard index is rotated by the object index so that the short last shard (252 = 7 x 32 + 28) lands on a different
  # rank for every object: each rank of an 8-GPU job handles exactly 252 hypotheses
  sl = [ranges[rotated_shard(o, rank, world)] for o in range(len(objects))]
  refined = est.refiner.predict_multi([dict(rgb=ob['rgb'], xyz_map=ob['xyz'], K=ob['K'], mesh_tensors=est.mesh_tensors,
                                            mesh_diameter=est.diameter, ob_in_cams=ob['poses'][a:b])
                                       for ob, (a, b) in zip(objects, sl)], iteration=ITER)
  offs = [0]
  for a, b in sl:
    offs.append(offs[-1] + (b - a))
  feats = est.scorer.extract_features_multi([dict(rgb=ob['rgb'], depth=ob['depth'], K=ob['K'], mesh_tensors=est.mesh_tensors,
                                                  mesh_diameter=est.diameter, ob_in_cams=refined[offs[o]:offs[o + 1]])
                                             for o, ob in enumerate(objects)])
  return torch.cat([pack_rows(feats[offs[o]:offs[o + 1]], refined[offs[o]:offs[o + 1]], shard) for o in range(len(objects))], 0)


def step_finalize(est, objects, world, rank, gathered):
  """The cross-hypothesis tail of the objects this rank finalises (object o: rank o % world).  gathered: every rank's row
  block, (world * n_objects * shard, 528) in rank order.  Returns {object: (argmax, poses (252,4,4))}."""
  from foundationpose_amd.dist import gather_order, unpack_rows
  shard = math.ceil(N_HYP / world)
  gathered = gathered.reshape(world, len(objects), shard, -1)
  results = {}
  for o in range(len(objects)):
    if o % world != rank:
      continue
    feats_all, poses_all = unpack_rows(gathered[gather_order(o, world), o].reshape(world * shard, -1), N_HYP, world)
    logits, am = est.scorer.score_tail(feats_all, L=N_HYP)
    results[o] = (am, poses_all)
  return results


def step(est, objects, world, rank):
  """One register-core pass over all objects: local part, ONE all-gather of [feat | pose] rows, tail on the owning rank."""
  from foundationpose_amd.dist import all_gather_rows
  rows = step_local(est, objects, world, rank)
  return step_finalize(est, objects, world, rank, all_gather_rows(rows) if world > 1 else rows)


def pmc_traffic():
  """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
  (profiles/r01_halo_traffic.json; FETCH_SIZE / WRITE_SIZE collected in separate passes and corrected
  as MI355X_MICROARCH.md prescribes).  None when no such profile has been committed."""
  path = os.path.join(REPO, 'profiles', 'r01_halo_traffic.json')
  try:
    with open(path) as f:
      return json.load(f)
  except OSError:
    return None


def cpu_baseline():
  """The CPU oracle timed on this host's cores on a bounded sample of the same workload."""
  from tests import util
  from oracle.predict import OracleFoundationPose
  from foundationpose_amd import synthetic as S
  from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
  # the GPU box exposes 256 logical CPUs but one GPU's share is 16: more threads only oversubscribe
  cores = min(os.cpu_count() or 1, 16)
  os.environ['OMP_NUM_THREADS'] = str(cores)
  torch.set_num_threads(cores)
  sc = util.scene(0)
  n_s = 126            # half of the 252-hypothesis workload: 10-15 s on 16 threads
  orc = OracleFoundationPose(sc['mt'], sc['diameter'], sc['center'], sc['grid'][:n_s], S.make_refine_state_dict(0),
                             S.make_score_state_dict(1), refine_cfg=dict(REFINE_DEFAULT), score_cfg=dict(SCORE_DEFAULT))
  t0 = time.time()
  orc.register(sc['K'], sc['rgb'], sc['depth'], sc['mask'], iteration=ITER, chunk=16)
  dt = time.time() - t0
  return dict(value=n_s / dt, unit='pose-hypotheses/sec', cores=cores, kind='port',
              sample=f'{n_s} hypotheses of the same scene, est_refine_iter={ITER} + score, oracle/ (torch-CPU fp32 nets + C/OpenMP '
                     f'rasteriser), {dt:.1f} s wall incl. depth filtering')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
  # FP_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend - rehearses the multi-process flow (rendezvous, barriers,
  # the all-gather, finalisation) on a one-GPU box; the numbers it prints mean nothing
  rehearsal = os.environ.get('FP_BENCH_REHEARSAL') == '1'
  if rehearsal:
    local_rank = 0
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if rehearsal:
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=device)

  est, objects = build_job(device, n_objects=world, rank=rank)
  ctx = est.refiner.ctx
  ctx.reserve(N_HYP)

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    step(est, objects, world, rank)
  barrier()
  ctx.prof_reset()
  ctx.prof_enable(True)          # HIP events around every dominant-kernel launch, on the launch stream
  t0 = time.perf_counter()
  for _ in range(args.steps):
    res = step(est, objects, world, rank)
  barrier()
  dt = time.perf_counter() - t0
  ctx.prof_enable(False)
  conv = ctx.prof_read('conv3x3_halo')
  tmax = torch.tensor([dt], device=device, dtype=torch.float64)
  if world > 1:
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
  dt = float(tmax.item())

  if rank == 0:
    total_hyp = N_HYP * world * args.steps
    achieved = conv['flops'] / (conv['total_ms'] * 1e-3) / 1e12 if conv['total_ms'] > 0 else 0.0
    out = {
      'metric': 'pose-hypotheses/sec (render+refine+score), 252 hyp x 160x160',
      'value': total_hyp / dt, 'unit': 'pose-hypotheses/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f16', 'data': 'synthetic',
      'config': {'workload': 'configs[1]: single mesh (8066 v / 16128 f), 252 hypotheses, est_refine_iter=5, 160x160 crops, '
                             '480x640 RGB-D frame; one such object per GPU, hypotheses sharded over all ranks',
                 'hypotheses_per_object': N_HYP, 'objects': world, 'est_refine_iter': ITER, 'parallelism': f'hyp-shard x{world}',
                 'weights': 'seeded random (reference state_dict layout)'},
      'roofline': {'bound': 'mfma', 'kernel': 'conv3x3_halo_dma_kernel (3x3 stride-1 convolutions: 93 % of the conv FLOPs)',
                   'achieved': achieved, 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F16_TFLOPS,
                   'avg_launch_ms': conv['total_ms'] / max(conv['launches'], 1), 'launches': conv['launches'],
                   'flops_per_launch': conv['flops'] / max(conv['launches'], 1),
                   'traffic': (pmc_traffic() or {}).get('total'), 'traffic_detail': pmc_traffic()},
    }
    classes = {}
    for c in ('conv3x3_halo', 'conv3x3_s2', 'conv7x7', 'linear', 'attention', 'render'):
      r = ctx.prof_read(c)
      if r['launches']:
        classes[c] = {'ms_per_step': r['total_ms'] / args.steps, 'launches_per_step': r['launches'] / args.steps,
                      'tflops': (r['flops'] / (r['total_ms'] * 1e-3) / 1e12) if r['flops'] else None}
    out['kernel_classes'] = classes       # HIP-event time per kernel class (same events as the roofline figure)
    out['kernel_classes_note'] = ('linear / attention: RefineNet runs its two transformer heads on two streams, their launches overlap '
                                  'and each counts its own span (sum > wall time); the convolution classes and render run alone')
    if not args.no_cpu_baseline and world == 1:        # timed on rank 0 of the single-GPU run only
      out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()

"""Runs tracking frames (under rocprofv3 --kernel-trace: scripts/prof_track2.sh).  MODE=one|multi, FRAMES=n, GRAPH=0|1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from foundationpose_amd import synthetic as S
from foundationpose_amd.Utils import nvdiffrast_render
from foundationpose_amd.synthetic import trajectory

dev = torch.device('cuda', 0)
est, objects = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(64)
n = int(os.environ.get('FRAMES', '60'))
mode = os.environ.get('MODE', 'one')
K = S.YCB_K
poses = torch.as_tensor(trajectory(n), device=dev)
c, d, _ = nvdiffrast_render(K=K, H=480, W=640, ob_in_cams=poses, mesh_tensors=est.mesh_tensors, use_light=True)
rgbs = (c * 255).clamp(0, 255).to(torch.uint8)
depths = torch.where(d > 0, d, torch.full_like(d, 1.2))
# a frame as ONE buffer [depth float32 | rgb uint8], as bench.tracking_fps hands it over: one upload copy per frame
nb_d, nb_c = 480 * 640 * 4, 480 * 640 * 3
packed = torch.empty((n, nb_d + nb_c), dtype=torch.uint8, device=dev)
packed[:, :nb_d] = depths.reshape(n, -1).contiguous().view(torch.uint8)
packed[:, nb_d:] = rgbs.reshape(n, -1)
depths = [packed[f, :nb_d].view(torch.float).reshape(480, 640) for f in range(n)]
rgbs = [packed[f, nb_d:].reshape(480, 640, 3) for f in range(n)]
est.enable_track_graph(os.environ.get('GRAPH', '0') == '1')
fn = (lambda f: est.track_one(rgbs[f], depths[f], K, iteration=2)) if mode == 'one' else (lambda f: est.track_multi(rgbs[f], depths[f], K, iteration=2, n_hypotheses=64))
starts = [poses[max(f - 1, 0)].clone() for f in range(n)]      # teacher-forced start poses (bench.tracking_fps)
def step(f):
  est.pose_last = starts[f]
  fn(f)
for f in range(10): step(f)
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(n): step(f)
torch.cuda.synchronize()
print(f'{mode}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame over {n} frames')

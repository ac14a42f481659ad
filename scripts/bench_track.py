"""configs[4]: tracking mode on a synthetic sequence (bench.tracking_fps): track_one (reference: 1 hypothesis,
src/estimater.py:250-268) and the 64-hypothesis mode (FoundationPose.track_multi), eager and as one hipGraph per frame.
FRAMES=n sets the sequence length.  Under `rocprofv3 --kernel-trace` this gives the per-kernel times of a B=1 frame."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda', 0)
est, _ = bench.build_job(dev, n_objects=1, rank=0)
est.refiner.ctx.reserve(int(os.environ.get('RESERVE', '64')))
print(json.dumps(bench.tracking_fps(est, dev, n_frames=int(os.environ.get('FRAMES', '200'))), indent=1))

"""Render 12 crops of the test scene through the nvdiffrast_render API and save them (used by test_render_in_sub_batches: run once
with the default scratch limit and once with FP_RENDER_SCRATCH_MAX lowered so that the batch is rendered in sub-batches)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
import numpy as np
import torch
from foundationpose_amd import Utils as U
from tests import util
from oracle import geometry as G
s = util.scene(0)
poses = util.hypotheses(s, 12, jitter_seed=5)
tf = G.compute_crop_window_tf_batch(torch.from_numpy(poses), s['K'], 1.2, (160, 160), s['diameter'])
bbox = G.crop_bbox2d_ori(tf, (160, 160))
e = {}
c, d, n = U.nvdiffrast_render(K=s['K'], H=480, W=640, ob_in_cams=torch.from_numpy(poses).cuda(), mesh_tensors=util.to_dev(s['mt']), bbox2d=bbox.cuda(),
                              output_size=(160, 160), use_light=True, get_normal=True, extra=e)
np.savez(sys.argv[1], c=c.cpu().numpy(), d=d.cpu().numpy(), n=n.cpu().numpy(), x=e['xyz_map'].cpu().numpy())
print('dumped', float((d > 0).float().mean()))

"""Render N_CROPS (default 12) crops of the test scene through the nvdiffrast_render API, run one fused refinement pass on them, and save
everything (used by test_render_in_sub_batches: run once with the default scratch limit and once with FP_RENDER_SCRATCH_MAX lowered so
that the standalone render AND the render inside the fused pass go out in sub-batches)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
import numpy as np
import torch
from foundationpose_amd import Utils as U
from tests import util
from oracle import geometry as G
s = util.scene(0)
n_crops = int(os.environ.get('N_CROPS', '12'))
poses = util.hypotheses(s, n_crops, jitter_seed=5)
tf = G.compute_crop_window_tf_batch(torch.from_numpy(poses), s['K'], 1.2, (160, 160), s['diameter'])
bbox = G.crop_bbox2d_ori(tf, (160, 160))
e = {}
c, d, n = U.nvdiffrast_render(K=s['K'], H=480, W=640, ob_in_cams=torch.from_numpy(poses).cuda(), mesh_tensors=util.to_dev(s['mt']), bbox2d=bbox.cuda(),
                              output_size=(160, 160), use_light=True, get_normal=True, extra=e)
from foundationpose_amd import synthetic as S
from foundationpose_amd.config import REFINE_DEFAULT
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT, device=torch.device('cuda', 0))
d_f = G.bilateral_filter_depth(G.erode_depth(s['depth']))
refined, _ = refiner.predict(rgb=s['rgb'], depth=d_f, K=s['K'], ob_in_cams=poses, xyz_map=G.depth2xyzmap(d_f, s['K']), mesh_tensors=util.to_dev(s['mt']),
                             mesh_diameter=s['diameter'], iteration=1)
np.savez(sys.argv[1], c=c.cpu().numpy(), d=d.cpu().numpy(), n=n.cpu().numpy(), x=e['xyz_map'].cpu().numpy(), r=refined.cpu().numpy())
print('dumped', float((d > 0).float().mean()))

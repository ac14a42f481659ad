"""Tool: run-to-run determinism of the refiner (two encoder heads on two streams) and of the attention core alone."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from foundationpose_amd import synthetic as S
from foundationpose_amd.config import REFINE_DEFAULT
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
from foundationpose_amd._lib import check, lib, ptr, stream_ptr, Context
from oracle import geometry as G
sc = util.scene(0)
poses = util.hypotheses(sc, 8, jitter_seed=7)
depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
xyz_map = G.depth2xyzmap(depth, sc['K'])
mt = util.to_dev(sc['mt'])
refiner = PoseRefinePredictor(state_dict=S.make_refine_state_dict(0), cfg=REFINE_DEFAULT)
pd = make_crop_data_batch((160, 160), poses, None, sc['rgb'], depth, sc['K'], 1.2, xyz_map, mesh_diameter=sc['diameter'], cfg=refiner.cfg, mesh_tensors=mt)
outs = [refiner.forward(pd) for _ in range(4)]
for o in outs[1:]:
  print('forward repeat: trans diff', float((o['trans'] - outs[0]['trans']).abs().max()), 'rot diff', float((o['rot'] - outs[0]['rot']).abs().max()))
ctx = Context.get('cuda:0')
g = torch.Generator().manual_seed(5)
B, T = 8, 400
qk = (torch.randn((B * T, 1024), generator=g) * 1.5).half().cuda()
vt = torch.zeros((B, 4, 128, 416), dtype=torch.float16)
vt[..., :T] = torch.randn((B, 4, 128, T), generator=g).half()
vt = vt.cuda()
res = []
for i in range(4):
  out = torch.full((B * T, 512), float('nan'), dtype=torch.float16, device='cuda')
  check(lib().fp_attention_f16(ctx.handle, ptr(qk), ptr(vt), B, T, ptr(out), stream_ptr()))
  torch.cuda.synchronize()
  res.append(out.clone())
for r in res[1:]:
  print('attention repeat: equal', torch.equal(r, res[0]), 'nan', int(torch.isnan(r).sum()))

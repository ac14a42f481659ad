import numpy as np, torch, sys
sys.path.insert(0, '.')
from tests import util
from foundationpose_amd import synthetic as S, _lib
from foundationpose_amd.config import REFINE_DEFAULT, SCORE_DEFAULT
from foundationpose_amd.predict_score import ScorePredictor
from oracle import predict as OP, geometry as G, nets
sc = util.scene(0)
ssd = S.make_score_state_dict(1)
scorer = ScorePredictor(state_dict=ssd, cfg=SCORE_DEFAULT)
depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
poses = util.hypotheses(sc, 32)
mt = util.to_dev(sc['mt'])
fg = scorer.extract_features(sc['rgb'], depth, sc['K'], poses, mesh_tensors=mt, mesh_diameter=sc['diameter']).cpu()
cfg = dict(OP.DEFAULT_SCORE_CFG, **SCORE_DEFAULT)
pd = OP.make_crop_data_batch_score(cfg, poses, sc['mt'], torch.as_tensor(sc['rgb'], dtype=torch.float32), torch.from_numpy(depth), sc['K'], sc['diameter'])
A = torch.cat([pd['rgbAs'], pd['xyz_mapAs']], 1); B = torch.cat([pd['rgbBs'], pd['xyz_mapBs']], 1)
fo = torch.cat([nets.score_extract_feat(ssd, A[i:i+16], B[i:i+16]) for i in range(0,32,16)])
print('feat scale', fo.abs().mean().item(), 'std across hyps', fo.std(0).mean().item())
d = fg - fo
print('feat err: max', d.abs().max().item(), 'mean abs', d.abs().mean().item(), 'common-mode', d.mean(0).abs().mean().item(), 'differential', (d - d.mean(0)).abs().mean().item())
# oracle nets on fp16-rounded inputs (A,B) -> isolates the input quantisation
fo16 = torch.cat([nets.score_extract_feat(ssd, A[i:i+16].half().float(), B[i:i+16].half().float()) for i in range(0,32,16)])
print('oracle feats, fp16-rounded inputs vs fp32 inputs: mean abs', (fo16-fo).abs().mean().item())
# oracle with B border row/col zeroed vs not
B2 = B.clone(); B2[:, 3:, 0, :] = 0; B2[:, 3:, :, 0] = 0
fo_b = torch.cat([nets.score_extract_feat(ssd, A[i:i+16], B2[i:i+16]) for i in range(0,32,16)])
print('oracle feats, border row/col of xyzB zeroed: mean abs diff', (fo_b-fo).abs().mean().item())
lo = nets.score_tail(ssd, fo, 32).reshape(-1); lg = nets.score_tail(ssd, fg, 32).reshape(-1)
print('logit diff (oracle tail on gpu feats vs oracle feats): common', (lg-lo).mean().item(), 'diff', ((lg-lg.mean())-(lo-lo.mean())).abs().max().item(), 'spread', lo.std().item())
lgg, _ = scorer.score_tail(fo.cuda(), 32)
print('gpu tail on oracle feats vs oracle logits: max', (lgg.cpu().reshape(-1)-lo).abs().max().item())

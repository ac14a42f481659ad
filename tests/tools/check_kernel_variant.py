"""Run under an environment that selects a non-default kernel variant (the FP_* knobs are read once per process): RefineNet and
ScoreNet on the golden inputs against the reference modules' outputs (tests/golden/reference_outputs.npz), same rules as
tests/test_gpu_pipeline.py.  Exit code 0 = the variant reproduces the goldens.  Used by test_kernel_variants_reproduce_the_goldens."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
import numpy as np
import torch
from foundationpose_amd import _lib, synthetic as S
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
from tests.test_gpu_pipeline import assert_tracks_input, to_net_tensor
from tests.util import net_inputs

golden = np.load(os.path.join(os.path.dirname(__file__), '..', 'golden', 'reference_outputs.npz'))
ctx = _lib.Context.get('cuda:0')
rnet = _lib.DeviceNet(ctx, _lib.FP_NET_REFINE, S.make_refine_state_dict(0), True)
snet = _lib.DeviceNet(ctx, _lib.FP_NET_SCORE, S.make_score_state_dict(1), True)
A, B = net_inputs(11, 8)
trans, rot = torch.empty((8, 3), device='cuda'), torch.empty((8, 3), device='cuda')
check(lib().fp_refine_forward(ctx.handle, rnet.handle, ptr(to_net_tensor(A, B)), 8, ptr(trans), ptr(rot), stream_ptr()))
assert_tracks_input(trans.cpu().numpy(), golden['refine_trans'], 0.1, 'trans head')
assert_tracks_input(rot.cpu().numpy(), golden['refine_rot'], 0.1, 'rot head')
np.testing.assert_allclose(trans.cpu().numpy(), golden['refine_trans'], atol=2e-3)
np.testing.assert_allclose(rot.cpu().numpy(), golden['refine_rot'], atol=2e-3)
A3, B3 = net_inputs(13, 8)
feats = torch.empty((8, 512), device='cuda')
check(lib().fp_score_features(ctx.handle, snet.handle, ptr(to_net_tensor(A3, B3)), 8, ptr(feats), stream_ptr()))
assert_tracks_input(feats.cpu().numpy(), golden['score_feats'], 0.1, 'ScoreNet features')
rms = lambda d: float(np.sqrt((np.asarray(d, dtype=np.float64) ** 2).mean()))
for what, got, key in (('trans', trans.cpu().numpy(), 'refine_trans'), ('rot', rot.cpu().numpy(), 'refine_rot'), ('feats', feats.cpu().numpy(), 'score_feats')):
  ours, theirs = rms(got - golden[key]), rms(golden[key + '_ac16'] - golden[key])
  print(f'{what}: rms |hip - ref fp32| {ours:.2e}  |ref fp16 - ref fp32| {theirs:.2e}')
  assert ours <= 2.0 * theirs
print('variant ok:', {k: v for k, v in os.environ.items() if k.startswith('FP_')})

"""Study for DESIGN.md section 9 (CPU only, no kernel): what a Winograd F(2x2,3x3) form of the 3x3 stride-1 layers would do to
RefineNet's outputs, emulated in torch on the oracle's network with the product's precision recipe
  D: every conv / linear with fp16-rounded operands and fp32 accumulation (the HIP kernels' numerics, BN not folded here),
  W: as D, but each 3x3 stride-1 convolution as  Y = At [ (G g Gt) . (Bt d B) ] A  with the transformed weights and the
     transformed input tiles rounded to fp16 (input transform in two fp16 stages, as v_pk_add_f16 would compute it),
against R: the fp32 oracle.  Inputs: real crops of the test scene (tests/util.scene).  Prints max |trans|, |rot| deviations
and the resulting pose deviation after one update."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from tests import util
from oracle import geometry as G, nets, predict as OP
from foundationpose_amd import synthetic as S

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
GM = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
h16 = lambda t: t.half().float()
_conv2d = F.conv2d


def wino_conv3x3(x, w, b):
  N, C, H, W = x.shape
  t = F.pad(h16(x), (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)            # (N,C,H/2,W/2,4,4)
  V = h16(h16(BT @ t) @ BT.T)
  U = h16(GM @ w @ GM.T)                                                      # (O,C,4,4), from the fp32 weights
  out = []
  for n in range(N):                                                          # per image: bounded memory
    M = torch.einsum('chwij,ocij->ohwij', V[n], U)
    Y = AT @ M @ AT.T                                                         # (O,H/2,W/2,2,2)
    out.append(Y.permute(0, 1, 3, 2, 4).reshape(w.shape[0], H, W))
  y = torch.stack(out)
  return y if b is None else y + b.reshape(1, -1, 1, 1)


def wino_conv3x3_rows(x, w, b):
  """F(2,3) ALONG ROWS only (the form of csrc/conv_wino.hip): per output pair (x0, x0+1) and kernel row ky, v = Bt d over the four input
  columns x0-1 .. x0+2 (ONE fp16 add each, as v_pk_add_f16 computes it), u = G g rounded to fp16 from the fp32 weights, products
  accumulated in fp32 over (ky, ci), y = At m in fp32."""
  N, C, H, W = x.shape
  t = F.pad(h16(x), (1, 1, 1, 1)).unfold(3, 4, 2)                                # (N,C,H+2,W/2,4)
  V = h16(torch.einsum('ij,nchwj->nchwi', BT, t))                                 # (N,C,H+2,W/2,4)
  U = h16(torch.einsum('ij,ockj->ocki', GM, w))                                   # (O,C,3,4)
  out = []
  for n in range(N):
    M = sum(torch.einsum('chwi,oci->ohwi', V[n][:, ky:ky + H], U[:, :, ky]) for ky in range(3))      # (O,H,W/2,4)
    Y = torch.einsum('ji,ohwi->ohwj', AT, M)                                       # (O,H,W/2,2)
    out.append(Y.reshape(w.shape[0], H, W))
  y = torch.stack(out)
  return y if b is None else y + b.reshape(1, -1, 1, 1)


def make_conv(mode):
  def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    if mode == 'W' and w.shape[-1] == 3 and stride == 1 and x.shape[-1] % 2 == 0:
      return wino_conv3x3(x, w, b)
    if mode == 'W1' and w.shape[-1] == 3 and stride == 1 and x.shape[-1] % 2 == 0:
      return wino_conv3x3_rows(x, w, b)
    return _conv2d(h16(x), h16(w), b, stride, padding, *a, **k)
  return conv2d


def run(mode, sd, A, B):
  if mode == 'R':
    return nets.refine_forward(sd, A, B, True)
  lin = F.linear
  F.conv2d = make_conv(mode)
  F.linear = lambda x, w, b=None: lin(h16(x), h16(w), b)
  try:
    return nets.refine_forward(sd, A, B, True)
  finally:
    F.conv2d, F.linear = _conv2d, lin


def main():
  torch.set_num_threads(8)
  sc = util.scene(0)
  n = 6
  poses = util.hypotheses(sc, n, jitter_seed=7)
  depth = G.bilateral_filter_depth(G.erode_depth(sc['depth']))
  xyz_map = torch.from_numpy(G.depth2xyzmap(depth, sc['K']))
  cfg = dict(OP.DEFAULT_REFINE_CFG)
  pd = OP.make_crop_data_batch_refine(cfg, poses, sc['mt'], torch.as_tensor(sc['rgb'], dtype=torch.float32), torch.from_numpy(depth), sc['K'],
                                      xyz_map, sc['diameter'])
  A = torch.cat([pd['rgbAs'], pd['xyz_mapAs']], 1).float()
  B = torch.cat([pd['rgbBs'], pd['xyz_mapBs']], 1).float()
  sd = S.make_refine_state_dict(0)
  with torch.no_grad():
    out = {m: run(m, sd, A, B) for m in ('R', 'D', 'W', 'W1')}
  pose = {m: OP.pose_update(cfg, pd['poseA'], out[m]['trans'], out[m]['rot'], sc['diameter'])[0] for m in out}
  for a, b in (('D', 'R'), ('W', 'R'), ('W', 'D'), ('W1', 'R'), ('W1', 'D')):
    dt = float((out[a]['trans'] - out[b]['trans']).abs().max()); dr = float((out[a]['rot'] - out[b]['rot']).abs().max())
    dp = float((pose[a] - pose[b]).abs().max())
    print(f'{a} vs {b}: max |d trans| {dt:.2e}  max |d rot| {dr:.2e}  (outputs are O({float(out["R"]["trans"].abs().max()):.2f}))  pose after one update {dp:.2e}')


if __name__ == '__main__':
  main()

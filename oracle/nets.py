"""Oracle: RefineNet / ScoreNetMultiPair forward passes as plain torch-CPU fp32 functions of a
reference-layout state_dict.

TEST INFRASTRUCTURE - see oracle/__init__.py.  PINNED: tests/golden/gen_golden.py loads the same
state_dict into the reference's own nn.Modules (learning/models/refine_network.py:27-93,
score_network.py:28-90, network_modules.py:37-137) and stores their outputs in tests/golden/.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-5      # nn.BatchNorm2d default
LN_EPS = 1e-5      # nn.TransformerEncoderLayer layer_norm_eps default
N_HEAD = 4
D_MODEL = 512


def _conv_bn_relu(sd, pre, x, stride, use_bn, relu=True):
  """network_modules.py:37-50 ConvBNReLU: conv(pad=(k-1)//2) -> BN(eval) -> ReLU.  keys <pre>.net.{0,1}.*"""
  w = sd[f'{pre}.net.0.weight']
  x = F.conv2d(x, w, sd[f'{pre}.net.0.bias'], stride=stride, padding=(w.shape[-1] - 1) // 2)
  if use_bn:
    x = F.batch_norm(x, sd[f'{pre}.net.1.running_mean'], sd[f'{pre}.net.1.running_var'],
                     sd[f'{pre}.net.1.weight'], sd[f'{pre}.net.1.bias'], training=False, eps=BN_EPS)
  return F.relu(x) if relu else x


def _res_block(sd, pre, x, use_bn):
  """network_modules.py:73-111 ResnetBasicBlock (bias=True, no downsample)."""
  out = F.conv2d(x, sd[f'{pre}.conv1.weight'], sd[f'{pre}.conv1.bias'], padding=1)
  if use_bn:
    out = F.batch_norm(out, sd[f'{pre}.bn1.running_mean'], sd[f'{pre}.bn1.running_var'],
                       sd[f'{pre}.bn1.weight'], sd[f'{pre}.bn1.bias'], training=False, eps=BN_EPS)
  out = F.relu(out)
  out = F.conv2d(out, sd[f'{pre}.conv2.weight'], sd[f'{pre}.conv2.bias'], padding=1)
  if use_bn:
    out = F.batch_norm(out, sd[f'{pre}.bn2.running_mean'], sd[f'{pre}.bn2.running_var'],
                       sd[f'{pre}.bn2.weight'], sd[f'{pre}.bn2.bias'], training=False, eps=BN_EPS)
  return F.relu(out + x)


def encode(sd, encA, encAB, A, B, use_bn, taps=None):
  """Shared trunk: refine_network.py:79-88 / score_network.py:66-72 -> tokens (bs,400,512) + pe."""
  bs = len(A)
  x = torch.cat([A, B], dim=0)
  x = _conv_bn_relu(sd, f'{encA}.0', x, 2, use_bn)
  if taps is not None: taps['encA0'] = x
  x = _conv_bn_relu(sd, f'{encA}.1', x, 2, use_bn)
  if taps is not None: taps['encA1'] = x
  x = _res_block(sd, f'{encA}.2', x, use_bn)
  x = _res_block(sd, f'{encA}.3', x, use_bn)
  if taps is not None: taps['encA3'] = x
  ab = torch.cat((x[:bs], x[bs:]), 1)
  ab = _res_block(sd, f'{encAB}.0', ab, use_bn)
  ab = _res_block(sd, f'{encAB}.1', ab, use_bn)
  if taps is not None: taps['encAB1'] = ab
  ab = _conv_bn_relu(sd, f'{encAB}.2', ab, 2, use_bn)
  ab = _res_block(sd, f'{encAB}.3', ab, use_bn)
  ab = _res_block(sd, f'{encAB}.4', ab, use_bn)
  if taps is not None: taps['encAB4'] = ab
  tok = ab.reshape(bs, ab.shape[1], -1).permute(0, 2, 1)
  tok = tok + sd['pos_embed.pe'][:, :tok.shape[1]]
  if taps is not None: taps['tokens'] = tok
  return tok


def mha(sd, pre, x):
  """nn.MultiheadAttention(512,4,batch_first=True) self-attention, eval."""
  Bn, L, D = x.shape
  dh = D // N_HEAD
  qkv = x @ sd[f'{pre}.in_proj_weight'].T + sd[f'{pre}.in_proj_bias']
  q, k, v = qkv.split(D, dim=-1)
  q = q.reshape(Bn, L, N_HEAD, dh).transpose(1, 2)
  k = k.reshape(Bn, L, N_HEAD, dh).transpose(1, 2)
  v = v.reshape(Bn, L, N_HEAD, dh).transpose(1, 2)
  att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
  o = (att @ v).transpose(1, 2).reshape(Bn, L, D)
  return o @ sd[f'{pre}.out_proj.weight'].T + sd[f'{pre}.out_proj.bias']


def encoder_layer(sd, pre, x):
  """nn.TransformerEncoderLayer(512,4,dim_feedforward=512,batch_first=True): post-norm, ReLU."""
  x = F.layer_norm(x + mha(sd, f'{pre}.self_attn', x), (D_MODEL,), sd[f'{pre}.norm1.weight'], sd[f'{pre}.norm1.bias'], LN_EPS)
  ff = F.relu(x @ sd[f'{pre}.linear1.weight'].T + sd[f'{pre}.linear1.bias']) @ sd[f'{pre}.linear2.weight'].T + sd[f'{pre}.linear2.bias']
  return F.layer_norm(x + ff, (D_MODEL,), sd[f'{pre}.norm2.weight'], sd[f'{pre}.norm2.bias'], LN_EPS)


@torch.no_grad()
def refine_forward(sd, A, B, use_bn=True, taps=None):
  """refine_network.py:73-93 -> {'trans': (bs,3), 'rot': (bs,3|6)}"""
  tok = encode(sd, 'encodeA', 'encodeAB', A, B, use_bn, taps)
  out = {}
  for name in ('trans', 'rot'):
    h = encoder_layer(sd, f'{name}_head.0', tok)
    h = h @ sd[f'{name}_head.1.weight'].T + sd[f'{name}_head.1.bias']
    out[name] = h.mean(dim=1)
  return out


@torch.no_grad()
def score_extract_feat(sd, A, B, use_bn=True, taps=None):
  """score_network.py:60-74 -> (bs,512)"""
  tok = encode(sd, 'encoderA', 'encoderAB', A, B, use_bn, taps)
  return mha(sd, 'att', tok).mean(dim=1).reshape(len(A), -1)


@torch.no_grad()
def score_tail(sd, feats, L):
  """score_network.py:82-88 -> (bs,L) logits"""
  bs = feats.shape[0] // L
  x = feats.reshape(bs, L, -1)
  x = mha(sd, 'att_cross', x)
  return (x @ sd['linear.weight'].T + sd['linear.bias']).reshape(bs, L)


@torch.no_grad()
def score_forward(sd, A, B, L, use_bn=True, taps=None, chunk=None):
  """score_network.py:77-90.  `chunk` only bounds CPU memory; the math is per-hypothesis up to feats."""
  if chunk is None:
    feats = score_extract_feat(sd, A, B, use_bn, taps)
  else:
    feats = torch.cat([score_extract_feat(sd, A[i:i + chunk], B[i:i + chunk], use_bn) for i in range(0, len(A), chunk)], 0)
  return {'score_logit': score_tail(sd, feats, L), 'feats': feats}

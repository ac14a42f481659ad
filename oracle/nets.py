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


# ------------------------------------------------------------------------------------------------------------------------
# RefineNet with the PRODUCT'S NUMERIC RECIPE ("d16"), still on the CPU in plain torch: the same roundings to fp16 at the same
# points as the HIP kernels, fp32 everywhere else.  Products of fp16 values are exact in fp32, so what is
# left between this chain and the HIP chain is the ORDER of the fp32 sums (and exp2 to an ulp): if the two agree far inside the
# fp16-vs-fp32 scatter, the HIP path differs from the fp32 oracle by precision only, not by logic.
# Rounding points (kernel that rounds): network input (raster.hip / crop.hip write the fp16 net tensor); BN folded in fp32, then
# the weights ONCE (net.hip make_conv); every conv output after bias + residual + ReLU (+ pe) in fp32 (conv*.hip epilogues);
# q, k, v rows (tok_qkv.hip); the un-normalised softmax numerators per 64-key block against the running maximum, and the
# attention output (attn.hip); x1 = LayerNorm1(..) and ReLU(linear1) (head_mlp.hip).  LayerNorm2, token mean and the output
# Linear stay fp32 (attn.hip refine_tail_kernel).
# ------------------------------------------------------------------------------------------------------------------------
def _h(t):
  return t.half().float()


def fold_trunk_d16(sd, encA, encAB, use_bn):
  """-> {prefix: (w fp16-rounded (as fp32), b fp32)} with eval-BatchNorm folded in fp32 BEFORE the one rounding of the weights."""
  out = {}

  def fold(wkey, bnkey):
    w, b = sd[f'{wkey}.weight'].float(), sd[f'{wkey}.bias'].float()
    if use_bn:
      scale = sd[f'{bnkey}.weight'] / torch.sqrt(sd[f'{bnkey}.running_var'] + BN_EPS)
      shift = sd[f'{bnkey}.bias'] - sd[f'{bnkey}.running_mean'] * scale
      w, b = w * scale.reshape(-1, 1, 1, 1), b * scale + shift
    out[wkey] = (_h(w), b)
  for pre in (f'{encA}.0', f'{encA}.1', f'{encAB}.2'):
    fold(f'{pre}.net.0', f'{pre}.net.1')
  for pre in (f'{encA}.2', f'{encA}.3', f'{encAB}.0', f'{encAB}.1', f'{encAB}.3', f'{encAB}.4'):
    fold(f'{pre}.conv1', f'{pre}.bn1')
    fold(f'{pre}.conv2', f'{pre}.bn2')
  return out


def encode_d16(sd, fw, encA, encAB, A, B):
  bs = len(A)

  def conv(key, x, stride, res=None, post=None):
    w, b = fw[key]
    y = F.conv2d(x, w, b, stride=stride, padding=(w.shape[-1] - 1) // 2)
    if res is not None:
      y = y + res
    y = F.relu(y)
    if post is not None:
      y = y + post
    return _h(y)

  def block(pre, x, post=None):
    return conv(f'{pre}.conv2', conv(f'{pre}.conv1', x, 1), 1, res=x, post=post)
  x = _h(torch.cat([A, B], dim=0))
  x = conv(f'{encA}.0.net.0', x, 2)
  x = conv(f'{encA}.1.net.0', x, 2)
  x = block(f'{encA}.2', x)
  x = block(f'{encA}.3', x)
  ab = torch.cat((x[:bs], x[bs:]), 1)
  ab = block(f'{encAB}.0', ab)
  ab = block(f'{encAB}.1', ab)
  ab = conv(f'{encAB}.2.net.0', ab, 2)
  ab = block(f'{encAB}.3', ab)
  pe = sd['pos_embed.pe'][0, :400].T.reshape(1, D_MODEL, 20, 20)          # the last conv adds pos_embed.pe in its epilogue, then rounds
  ab = block(f'{encAB}.4', ab, post=pe)
  return ab.reshape(bs, D_MODEL, -1).permute(0, 2, 1)


def mha_core_d16(q, k, v, key_block=64):
  """attn.hip: per 64-key block, scores in fp32, running maximum, numerators exp((s - m_run) / sqrt(128)) rounded to fp16 for the
  PV product, their fp32 sum unrounded; O rescaled when the maximum moves; O / l rounded to fp16.  q, k, v: (B, H, L, dh) fp16-valued."""
  Bn, Hh, L, dh = q.shape
  c = 1.0 / math.sqrt(dh)
  m = torch.full((Bn, Hh, L, 1), -1e30)
  l = torch.zeros((Bn, Hh, L, 1))
  o = torch.zeros((Bn, Hh, L, dh))
  for k0 in range(0, L, key_block):
    s = q @ k[:, :, k0:k0 + key_block].transpose(-1, -2)
    m_new = torch.maximum(m, s.max(-1, keepdim=True).values)
    alpha = torch.exp((m - m_new) * c)
    e = torch.exp((s - m_new) * c)
    l = l * alpha + e.sum(-1, keepdim=True)
    o = o * alpha + _h(e) @ v[:, :, k0:k0 + key_block]
    m = m_new
  return _h(o / l)


def head_d16(sd, pre, out_pre, tok):
  """One RefineNet head on fp16-valued tokens (bs, 400, 512) -> (bs, out_dim)."""
  Bn, L, D = tok.shape
  dh = D // N_HEAD
  W = lambda k: _h(sd[k].float())
  qkv = _h(tok @ W(f'{pre}.self_attn.in_proj_weight').T + sd[f'{pre}.self_attn.in_proj_bias'])
  q, k, v = (t.reshape(Bn, L, N_HEAD, dh).transpose(1, 2) for t in qkv.split(D, dim=-1))
  att = mha_core_d16(q, k, v).transpose(1, 2).reshape(Bn, L, D)
  x1 = _h(F.layer_norm(tok + att @ W(f'{pre}.self_attn.out_proj.weight').T + sd[f'{pre}.self_attn.out_proj.bias'], (D_MODEL,),
                       sd[f'{pre}.norm1.weight'], sd[f'{pre}.norm1.bias'], LN_EPS))
  ff = _h(F.relu(x1 @ W(f'{pre}.linear1.weight').T + sd[f'{pre}.linear1.bias']))
  y = F.layer_norm(x1 + ff @ W(f'{pre}.linear2.weight').T + sd[f'{pre}.linear2.bias'], (D_MODEL,), sd[f'{pre}.norm2.weight'],
                   sd[f'{pre}.norm2.bias'], LN_EPS)
  return y.mean(dim=1) @ sd[f'{out_pre}.weight'].T + sd[f'{out_pre}.bias']


@torch.no_grad()
def refine_forward_d16(sd, A, B, use_bn=True, folded=None):
  """refine_network.py:73-93 with the HIP kernels' precision recipe (see the block comment above)."""
  fw = folded if folded is not None else fold_trunk_d16(sd, 'encodeA', 'encodeAB', use_bn)
  tok = encode_d16(sd, fw, 'encodeA', 'encodeAB', A, B)
  return {'trans': head_d16(sd, 'trans_head.0', 'trans_head.1', tok), 'rot': head_d16(sd, 'rot_head.0', 'rot_head.1', tok)}

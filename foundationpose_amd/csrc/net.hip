// Network weights (state_dict -> folded, packed device buffers) and the forward schedules of
// RefineNet (learning/models/refine_network.py:73-93) and ScoreNetMultiPair
// (learning/models/score_network.py:60-90) as sequences of the gfx950 kernels in conv.hip / attn.hip.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>

struct ConvW {
  f16 *w = nullptr;
  f16 *wpk = nullptr;     // 3x3 stride-2 layers: the same weights in the fragment order of conv_s2.hip
  f16 *wwino = nullptr;   // 3x3 stride-1 layers from 128 channels on: the Winograd image of the fp32 weights (conv_wino.hip)
  f16 *wsm = nullptr;     // 3x3 stride-1 layers with 128 / 256 / 512 input channels: fragment order of conv_small.hip (launches of a few images)
  float *bias = nullptr;
  int Cin = 0, Cout = 0, K = 1, Kpad = 0, stride = 1;
};

struct LinP {   // 512 rows of an nn.Linear(512, .) in the fragment order tok_gemm.hip reads (pack_tok_weights)
  f16 *w = nullptr;
  float *bias = nullptr;
};

struct HeadW {  // one nn.TransformerEncoderLayer + Linear
  LinP q, k, v, out, ff1, ff2;
  float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr, *hw = nullptr, *hb = nullptr;
  int out_dim = 0;
};

struct LinF32 {
  float *w = nullptr, *b = nullptr, *wt = nullptr;      // w: (N, K) row-major; wt: its transpose (K, N) for the column-per-lane kernel
  int N = 0, K = 0;
};

struct fp_net {
  int kind = 0, use_bn = 1;
  ConvW trunk[15];
  float *pe = nullptr;  // 400 x 512
  HeadW heads[2];       // refine: trans, rot
  LinP att_q, att_k, att_v;  // score: self.att in_proj on tokens
  LinF32 att_out;
  // score tail (score_tail.hip): q | k of att_cross transposed (512, 1024) fp32 + bias (1024); the value path folded in float64 at load
  // time: u (4, 512), c (4), b_eff
  float *tail_wqk_t = nullptr, *tail_bqk = nullptr;
  double *tail_u = nullptr, *tail_c = nullptr, tail_b_eff = 0.0;
  std::vector<void *> allocs;
};

// Hypotheses per network pass: the whole batch.  FP_CHUNK=n (environment, experiments only) splits it.
int fp_hyp_chunk(int n_total) {
  static int env = getenv("FP_CHUNK") ? atoi(getenv("FP_CHUNK")) : -1;
  int ch = env >= 0 ? env : 0;
  if (ch <= 0 || ch > n_total) ch = n_total;
  return ch;
}

int fp_trunk_split_min() {
  // from 48 hypotheses on (round 4, refine x5 + score of one object: 40 hypotheses 6.55 ms as one batch / 7.08 split, 48: 7.82 / 7.55, 56: 8.87 / 8.17,
  // 63: 9.85 / 9.23, 64: 9.95 / 9.32, 96: 13.97 / 13.71; scripts/bench_nhyp.py): below, the halves' tiles get too small
  static const int n_min = getenv("FP_TRUNK_MIN") ? atoi(getenv("FP_TRUNK_MIN")) : 48;
  static const int n_streams = getenv("FP_TRUNK_STREAMS") ? atoi(getenv("FP_TRUNK_STREAMS")) : 2;
  return n_streams < 2 ? 0x7fffffff : n_min;
}

namespace {

struct SD {
  std::map<std::string, const fp_tensor *> m;
  const fp_tensor *get(const std::string &k) const {
    auto it = m.find(k);
    return it == m.end() ? nullptr : it->second;
  }
};

int need(const SD &sd, const std::string &k, int ndim, std::initializer_list<int64_t> shape, const fp_tensor **out) {
  const fp_tensor *t = sd.get(k);
  if (!t) {
    fp_set_error("state_dict: missing key '%s'", k.c_str());
    return FP_EKEY;
  }
  if (t->ndim != ndim) {
    fp_set_error("state_dict: '%s' has ndim %d, expected %d", k.c_str(), t->ndim, ndim);
    return FP_EKEY;
  }
  int i = 0;
  for (int64_t s : shape) {
    if (s >= 0 && t->shape[i] != s) {
      fp_set_error("state_dict: '%s' dim %d is %lld, expected %lld", k.c_str(), i, (long long)t->shape[i], (long long)s);
      return FP_EKEY;
    }
    ++i;
  }
  *out = t;
  return FP_OK;
}

template <typename T>
int upload(fp_net *net, const std::vector<T> &h, T **d) {
  void *p = nullptr;
  FP_CHECK_HIP(hipMalloc(&p, h.size() * sizeof(T)));
  net->allocs.push_back(p);
  FP_CHECK_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *d = (T *)p;
  return FP_OK;
}

// conv (+ optional BatchNorm eval fold): y = (conv(x,w)+b - mean) * gamma/sqrt(var+eps) + beta
int make_conv(fp_net *net, const SD &sd, const std::string &wkey, const std::string &bnkey, bool use_bn, int stride, ConvW *out) {
  const fp_tensor *w, *b;
  FP_TRY(need(sd, wkey + ".weight", 4, {-1, -1, -1, -1}, &w));
  const int Cout = (int)w->shape[0], Cin = (int)w->shape[1], K = (int)w->shape[2];
  FP_REQUIRE(w->shape[3] == K, "conv '%s' is not square", wkey.c_str());
  FP_TRY(need(sd, wkey + ".bias", 1, {Cout}, &b));
  std::vector<float> scale(Cout, 1.f), shift(Cout, 0.f);
  if (use_bn) {
    const fp_tensor *g, *be, *mu, *var;
    FP_TRY(need(sd, bnkey + ".weight", 1, {Cout}, &g));
    FP_TRY(need(sd, bnkey + ".bias", 1, {Cout}, &be));
    FP_TRY(need(sd, bnkey + ".running_mean", 1, {Cout}, &mu));
    FP_TRY(need(sd, bnkey + ".running_var", 1, {Cout}, &var));
    for (int c = 0; c < Cout; ++c) {
      scale[c] = g->data[c] / std::sqrt(var->data[c] + 1e-5f);
      shift[c] = be->data[c] - mu->data[c] * scale[c];
    }
  }
  const int CinP = (Cin < 8) ? 8 : Cin;
  FP_REQUIRE(CinP == 8 || CinP % 32 == 0, "conv '%s': Cin=%d unsupported", wkey.c_str(), Cin);
  const int Kraw = K * K * CinP, Kpad = (Kraw + 31) / 32 * 32;
  std::vector<f16> hw((size_t)Cout * Kpad, (f16)0.f);
  std::vector<float> hb(Cout);
  for (int co = 0; co < Cout; ++co) {
    for (int ci = 0; ci < Cin; ++ci)
      for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx)
          hw[(size_t)co * Kpad + (ky * K + kx) * CinP + ci] = (f16)(w->data[(((size_t)co * Cin + ci) * K + ky) * K + kx] * scale[co]);
    hb[co] = b->data[co] * scale[co] + shift[co];
  }
  out->Cin = CinP;
  out->Cout = Cout;
  out->K = K;
  out->Kpad = Kpad;
  out->stride = stride;
  FP_TRY(upload(net, hw, &out->w));
  FP_TRY(upload(net, hb, &out->bias));
  if (K == 3 && stride == 1 && CinP >= 128 && CinP % 32 == 0 && Cout % 64 == 0 && fp_wino_mode() != 0) {
    // Winograd F(2,3) along rows: u = G g from the fp32 BN-folded weights, rounded to fp16 once
    std::vector<f16> hu(wino_packed_halfs(Cout, CinP));
    wino_pack_weights(w->data, scale.data(), Cout, Cin, hu.data());
    FP_TRY(upload(net, hu, &out->wwino));
  }
  // fragment-ordered copy for the band kernels: the 3x3 stride-2 layers (conv_s2.hip) and the 128 -> 128 / 256 -> 256 stride-1 layers, which run on 40x40 maps (conv_s1b.hip)
  if (K == 3 && (stride == 2 || (stride == 1 && CinP == Cout && (Cout == 128 || Cout == 256))) && CinP % 16 == 0 && s2_ct_for(Cout) != 0) {
    void *pk = nullptr;
    FP_CHECK_HIP(hipMalloc(&pk, s2_packed_halfs(Cout, CinP) * sizeof(f16)));
    net->allocs.push_back(pk);
    out->wpk = (f16 *)pk;
    FP_TRY(s2_pack_weights(out->w, Cout, CinP, Kpad, out->wpk, nullptr, stride == 1 ? 2 : 0, stride == 1 ? 1 : 0));      // (conv_s1b.hip: 64 couts per wave group)
    FP_CHECK_HIP(hipStreamSynchronize(nullptr));
  }
  if (K == 3 && ((stride == 1 && (CinP == 128 || CinP == 256 || CinP == 512)) || (stride == 2 && (CinP == 256 || CinP == 64))) && Cout % 32 == 0 && Kpad == 9 * CinP) {
    void *pk = nullptr;
    FP_CHECK_HIP(hipMalloc(&pk, small_packed_halfs(Cout, CinP) * sizeof(f16)));
    net->allocs.push_back(pk);
    out->wsm = (f16 *)pk;
    FP_TRY(small_pack_weights(out->w, Cout, CinP, Kpad, out->wsm, nullptr));
    FP_CHECK_HIP(hipStreamSynchronize(nullptr));
  }
  return FP_OK;
}

// 512 x 512 row-major fp32 -> fp16 in the fragment order of tok_gemm.hip
void pack_tok_weights(const float *w, f16 *out) {
  for (int wave = 0; wave < 4; ++wave)
    for (int k16 = 0; k16 < 32; ++k16)
      for (int i = 0; i < 4; ++i)
        for (int lane = 0; lane < 64; ++lane) {
          const int lr = lane & 31, lh = lane >> 5;
          const float *src = w + (size_t)(wave * 128 + i * 32 + lr) * 512 + k16 * 16 + lh * 8;
          f16 *dst = out + ((((size_t)wave * 32 + k16) * 4 + i) * 64 + lane) * 8;
          for (int e = 0; e < 8; ++e) dst[e] = (f16)src[e];
        }
}

// rows [r0, r0 + 512) of an (R, 512) linear weight in the MFMA-fragment order of tok_gemm.hip: [wave 4][k16 32][i 4][lane 64][8],
// lane (lh*32 + lr) of fragment (wave, k16, i) holds W[r0 + wave*128 + i*32 + lr][k16*16 + lh*8 .. + 8] - one coalesced
// 1-KB load per fragment
int make_linear(fp_net *net, const SD &sd, const std::string &wkey, const std::string &bkey, int r0, LinP *out) {
  const fp_tensor *w, *b;
  FP_TRY(need(sd, wkey, 2, {-1, 512}, &w));
  FP_TRY(need(sd, bkey, 1, {w->shape[0]}, &b));
  FP_REQUIRE(r0 + 512 <= w->shape[0], "linear '%s': rows out of range", wkey.c_str());
  std::vector<f16> hw((size_t)512 * 512);
  std::vector<float> hb(512);
  pack_tok_weights(w->data + (size_t)r0 * 512, hw.data());
  for (int n = 0; n < 512; ++n) hb[n] = b->data[r0 + n];
  FP_TRY(upload(net, hw, &out->w));
  FP_TRY(upload(net, hb, &out->bias));
  return FP_OK;
}

int make_vec(fp_net *net, const SD &sd, const std::string &key, int n, float **out) {
  const fp_tensor *t = sd.get(key);
  if (!t) {
    fp_set_error("state_dict: missing key '%s'", key.c_str());
    return FP_EKEY;
  }
  int64_t tot = 1;
  for (int i = 0; i < t->ndim; ++i) tot *= t->shape[i];
  FP_REQUIRE(tot == n, "state_dict: '%s' has %lld elements, expected %d", key.c_str(), (long long)tot, n);
  std::vector<float> h(t->data, t->data + n);
  return upload(net, h, out);
}

int make_linf32(fp_net *net, const SD &sd, const std::string &wkey, const std::string &bkey, int N, int K, LinF32 *out) {
  out->N = N;
  out->K = K;
  FP_TRY(make_vec(net, sd, wkey, N * K, &out->w));
  FP_TRY(make_vec(net, sd, bkey, N, &out->b));
  const fp_tensor *t = sd.get(wkey);
  std::vector<float> tr((size_t)N * K);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) tr[(size_t)k * N + n] = t->data[(size_t)n * K + k];
  return upload(net, tr, &out->wt);
}

// att_cross + linear (score_network.py:83-85) for score_tail.hip: q | k rows of in_proj transposed; the value path folded in float64:
//   w_eff = W_o^T lin.w, b_eff = lin.w . b_o + lin.b, u_h = Wv_h^T w_eff^h, c_h = bv_h . w_eff^h   (derivation: score_tail.hip)
int make_score_tail(fp_net *net, const SD &sd) {
  const fp_tensor *wi, *bi, *wo, *bo, *lw, *lb;
  FP_TRY(need(sd, "att_cross.in_proj_weight", 2, {1536, 512}, &wi));
  FP_TRY(need(sd, "att_cross.in_proj_bias", 1, {1536}, &bi));
  FP_TRY(need(sd, "att_cross.out_proj.weight", 2, {512, 512}, &wo));
  FP_TRY(need(sd, "att_cross.out_proj.bias", 1, {512}, &bo));
  FP_TRY(need(sd, "linear.weight", 2, {1, 512}, &lw));
  FP_TRY(need(sd, "linear.bias", 1, {1}, &lb));
  std::vector<float> wt((size_t)512 * 1024), bqk(1024);
  for (int n = 0; n < 1024; ++n) {
    bqk[n] = bi->data[n];
    for (int k = 0; k < 512; ++k) wt[(size_t)k * 1024 + n] = wi->data[(size_t)n * 512 + k];
  }
  std::vector<double> weff(512, 0.0), u((size_t)4 * 512, 0.0), c(4, 0.0);
  double beff = (double)lb->data[0];
  for (int n = 0; n < 512; ++n) {
    const double l = (double)lw->data[n];
    beff += l * (double)bo->data[n];
    for (int k = 0; k < 512; ++k) weff[k] += l * (double)wo->data[(size_t)n * 512 + k];
  }
  for (int h = 0; h < 4; ++h)
    for (int d = 0; d < 128; ++d) {
      const double w = weff[h * 128 + d];
      const float *vr = wi->data + (size_t)(1024 + h * 128 + d) * 512;
      c[h] += w * (double)bi->data[1024 + h * 128 + d];
      for (int k = 0; k < 512; ++k) u[(size_t)h * 512 + k] += w * (double)vr[k];
    }
  net->tail_b_eff = beff;
  FP_TRY(upload(net, wt, &net->tail_wqk_t));
  FP_TRY(upload(net, bqk, &net->tail_bqk));
  FP_TRY(upload(net, u, &net->tail_u));
  FP_TRY(upload(net, c, &net->tail_c));
  return FP_OK;
}

int make_trunk(fp_net *net, const SD &sd, const std::string &eA, const std::string &eAB, bool bn) {
  auto cbr = [&](const std::string &pre, int stride, ConvW *o) { return make_conv(net, sd, pre + ".net.0", pre + ".net.1", bn, stride, o); };
  auto res = [&](const std::string &pre, ConvW *o1, ConvW *o2) {
    FP_TRY(make_conv(net, sd, pre + ".conv1", pre + ".bn1", bn, 1, o1));
    return make_conv(net, sd, pre + ".conv2", pre + ".bn2", bn, 1, o2);
  };
  ConvW *t = net->trunk;
  FP_TRY(cbr(eA + ".0", 2, &t[0]));
  FP_TRY(cbr(eA + ".1", 2, &t[1]));
  FP_TRY(res(eA + ".2", &t[2], &t[3]));
  FP_TRY(res(eA + ".3", &t[4], &t[5]));
  FP_TRY(res(eAB + ".0", &t[6], &t[7]));
  FP_TRY(res(eAB + ".1", &t[8], &t[9]));
  FP_TRY(cbr(eAB + ".2", 2, &t[10]));
  FP_TRY(res(eAB + ".3", &t[11], &t[12]));
  FP_TRY(res(eAB + ".4", &t[13], &t[14]));
  FP_REQUIRE(t[0].Cin == 8 && t[0].Cout == 64 && t[0].K == 7 && t[1].Cout == 128 && t[6].Cin == 256 && t[10].Cout == 512,
             "trunk: unexpected layer shapes (c_in must be <= 8 -> padded to 8)");
  FP_TRY(make_vec(net, sd, "pos_embed.pe", 400 * 512, &net->pe));
  return FP_OK;
}

}  // namespace

extern "C" int fp_net_create(fp_ctx *ctx, int kind, const fp_tensor *tensors, int n_tensors, int use_bn, fp_net **out) {
  FP_REQUIRE(ctx && tensors && out, "fp_net_create: null argument");
  FP_REQUIRE(kind == FP_NET_REFINE || kind == FP_NET_SCORE, "fp_net_create: kind must be FP_NET_REFINE or FP_NET_SCORE");
  FP_CHECK_HIP(hipSetDevice(ctx->device));
  SD sd;
  for (int i = 0; i < n_tensors; ++i) sd.m[tensors[i].name] = &tensors[i];
  std::unique_ptr<fp_net> net(new fp_net);
  net->kind = kind;
  net->use_bn = use_bn;
  int rc = FP_OK;
  auto run = [&]() -> int {
    if (kind == FP_NET_REFINE) {
      FP_TRY(make_trunk(net.get(), sd, "encodeA", "encodeAB", use_bn));
      const char *names[2] = {"trans_head", "rot_head"};
      for (int h = 0; h < 2; ++h) {
        HeadW &H = net->heads[h];
        std::string p = std::string(names[h]) + ".0";
        FP_TRY(make_linear(net.get(), sd, p + ".self_attn.in_proj_weight", p + ".self_attn.in_proj_bias", 0, &H.q));
        FP_TRY(make_linear(net.get(), sd, p + ".self_attn.in_proj_weight", p + ".self_attn.in_proj_bias", 512, &H.k));
        FP_TRY(make_linear(net.get(), sd, p + ".self_attn.in_proj_weight", p + ".self_attn.in_proj_bias", 1024, &H.v));
        FP_TRY(make_linear(net.get(), sd, p + ".self_attn.out_proj.weight", p + ".self_attn.out_proj.bias", 0, &H.out));
        FP_TRY(make_linear(net.get(), sd, p + ".linear1.weight", p + ".linear1.bias", 0, &H.ff1));
        FP_TRY(make_linear(net.get(), sd, p + ".linear2.weight", p + ".linear2.bias", 0, &H.ff2));
        FP_TRY(make_vec(net.get(), sd, p + ".norm1.weight", 512, &H.ln1g));
        FP_TRY(make_vec(net.get(), sd, p + ".norm1.bias", 512, &H.ln1b));
        FP_TRY(make_vec(net.get(), sd, p + ".norm2.weight", 512, &H.ln2g));
        FP_TRY(make_vec(net.get(), sd, p + ".norm2.bias", 512, &H.ln2b));
        const fp_tensor *hw;
        FP_TRY(need(sd, std::string(names[h]) + ".1.weight", 2, {-1, 512}, &hw));
        H.out_dim = (int)hw->shape[0];
        FP_REQUIRE(H.out_dim >= 1 && H.out_dim <= 6, "head '%s': out_dim %d unsupported", names[h], H.out_dim);
        FP_TRY(make_vec(net.get(), sd, std::string(names[h]) + ".1.weight", H.out_dim * 512, &H.hw));
        FP_TRY(make_vec(net.get(), sd, std::string(names[h]) + ".1.bias", H.out_dim, &H.hb));
      }
    } else {
      FP_TRY(make_trunk(net.get(), sd, "encoderA", "encoderAB", use_bn));
      FP_TRY(make_linear(net.get(), sd, "att.in_proj_weight", "att.in_proj_bias", 0, &net->att_q));
      FP_TRY(make_linear(net.get(), sd, "att.in_proj_weight", "att.in_proj_bias", 512, &net->att_k));
      FP_TRY(make_linear(net.get(), sd, "att.in_proj_weight", "att.in_proj_bias", 1024, &net->att_v));
      FP_TRY(make_linf32(net.get(), sd, "att.out_proj.weight", "att.out_proj.bias", 512, 512, &net->att_out));
      FP_TRY(make_score_tail(net.get(), sd));
    }
    return FP_OK;
  };
  rc = run();
  if (rc != FP_OK) {
    for (void *p : net->allocs) (void)hipFree(p);
    return rc;
  }
  *out = net.release();
  return FP_OK;
}

extern "C" int fp_net_destroy(fp_net *net) {
  if (!net) return FP_OK;
  for (void *p : net->allocs) (void)hipFree(p);
  delete net;
  return FP_OK;
}

extern "C" int fp_net_rot_dim(const fp_net *net) { return (net && net->kind == FP_NET_REFINE) ? net->heads[1].out_dim : 0; }

// ---------------------------------------------------------------------------------------------
// forward schedules
// ---------------------------------------------------------------------------------------------
int conv_ksplit(const ConvArgs &a, int num_cu);      // conv.hip

namespace {

struct Conv2dCall {
  const f16 *in;
  int Nimg, H, W;
  const ConvW *cw;
  const f16 *res = nullptr;
  int relu = 1;
  void *out = nullptr;
  int out_mode = 0;
  int out_ld = 0, split_m = 0x7fffffff, coff_hi = 0;
  const float *post_add = nullptr;
  int post_period = 1;
  int tokens = 400;
};

int run_conv(fp_ctx *ctx, const Conv2dCall &c, hipStream_t s, float *splitk_scratch = nullptr, int hyp = 0) {
  ConvArgs a;
  const ConvW &w = *c.cw;
  a.in = c.in;
  a.w = w.w;
  a.wpk = w.wpk;
  a.wwino = w.wwino;
  a.wsm = w.wsm;
  a.hyp = hyp;
  a.bias = w.bias;
  a.res = c.res;
  a.post_add = c.post_add;
  a.out = c.out;
  a.Nimg = c.Nimg;
  a.H = c.H;
  a.W = c.W;
  a.Cin = w.Cin;
  a.KH = a.KW = w.K;
  a.stride = w.stride;
  a.pad = (w.K - 1) / 2;
  a.Ho = (c.H + 2 * a.pad - w.K) / w.stride + 1;
  a.Wo = (c.W + 2 * a.pad - w.K) / w.stride + 1;
  a.Cout = w.Cout;
  a.Kpad = w.Kpad;
  a.M = c.Nimg * a.Ho * a.Wo;
  a.relu = c.relu;
  a.out_mode = c.out_mode;
  a.out_ld = c.out_ld ? c.out_ld : w.Cout;
  a.split_m = c.split_m;
  a.coff_hi = c.coff_hi;
  a.post_period = c.post_period;
  a.tokens = c.tokens;
  if (splitk_scratch) {
    a.ksplit = conv_ksplit(a, ctx->num_cu);
    a.splitk = a.ksplit > 1 ? splitk_scratch : nullptr;
  }
  return launch_conv(ctx, a, s);
}

#define TAKE(ptr, type, count)                                                            \
  type *ptr = (type *)ctx->arena.take((size_t)(count) * sizeof(type));                    \
  if (!ptr) {                                                                             \
    fp_set_error("arena exhausted (%s); call fp_ctx_reserve with a larger max_hyp", #ptr); \
    return FP_ENOMEM;                                                                     \
  }

// ab0[n][p][128 .. 256) = feat[object of hypothesis h0 + n][p][0 .. 128): the B half of cat((a, b), 1) from one encoded crop per object
__global__ __launch_bounds__(256) void broadcast_side_b_kernel(const f16 *__restrict__ feat, SharedB sb, int h0, int N, f16 *__restrict__ ab0) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one 16-byte piece: 16 per pixel
  if (i >= (long long)N * 1600 * 16) return;
  const int n = (int)(i / (1600 * 16)), r = (int)(i - (long long)n * (1600 * 16)), px = r >> 4, c8 = r & 15;
  int g = 0;
  while (g + 1 < sb.n_groups && h0 + n >= sb.start[g + 1]) ++g;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u32x4 *>(ab0 + ((size_t)n * 1600 + px) * 256 + 128 + c8 * 8) =
      *reinterpret_cast<const u32x4 *>(feat + ((size_t)g * 1600 + px) * 128 + c8 * 8);
}

// shared trunk -> tokens (N*400, 512) fp16, positional embedding added.
// `ab` (optional): the two sides of encodeA as two chains on two streams.  encodeA runs on cat([A, B], 0) with shared weights
// (refine_network.py:74-78): nothing couples the rendered side A and the observed side B before the channel concat, and in a fused
// pass side A waits for the rasteriser while side B (observed crop -> stem -> ...) does not.  The caller has forked `ab` (two chains),
// side B's input is produced on ab->stream_for(0), side A's on `s`; the chains join in front of encodeAB.  Each side's launches are the
// 2N-image launches cut in two: a pixel's arithmetic does not depend on its batch, so the tokens are those of the single chain bit for bit.
int run_trunk(fp_ctx *ctx, const fp_net *net, const f16 *xA, const f16 *xB, int N, f16 **tokens_out, hipStream_t s, f16 *tok_dst = nullptr,
              StreamFanout *ab = nullptr, const SharedB *sb = nullptr, int h0 = 0) {
  const ConvW *t = net->trunk;
  const size_t n2 = (size_t)2 * N;
  TAKE(a0, f16, n2 * 80 * 80 * 64);
  TAKE(a1, f16, n2 * 40 * 40 * 128);
  TAKE(tA, f16, n2 * 40 * 40 * 128);
  TAKE(a2, f16, n2 * 40 * 40 * 128);
  TAKE(ab0, f16, (size_t)N * 40 * 40 * 256);
  TAKE(tB, f16, (size_t)N * 40 * 40 * 256);
  TAKE(ab1, f16, (size_t)N * 40 * 40 * 256);
  TAKE(c0, f16, (size_t)N * 400 * 512);
  TAKE(tC, f16, (size_t)N * 400 * 512);
  TAKE(c1, f16, (size_t)N * 400 * 512);
  f16 *tok = tok_dst;                  // (given by the caller when the batch is cut in two: both halves write into one token tensor)
  if (!tok) {
    TAKE(tok_, f16, (size_t)N * 400 * 512);
    tok = tok_;
  }
  // split-K scratch of the 3x3 stride-1 layers for 1 .. 4 hypotheses: 4 shares x the largest fp32 output (256 channels at 40x40)
  float *sk = nullptr;
  if (N <= 4) {
    TAKE(sk_, float, (size_t)4 * N * 1600 * 256 + (size_t)2 * n2 * 1600 * 128);
    sk = sk_;
  }
  Conv2dCall c;
  if (sb) {
    // side A alone; side B = the objects' encoded crops (SharedB)
    if (ab) FP_TRY(ab->join());                       // (the crops were encoded on the side stream)
    c = Conv2dCall{xA, N, 160, 160, &t[0]}; c.out = a0; FP_TRY(run_conv(ctx, c, s, sk, N));
    c = Conv2dCall{a0, N, 80, 80, &t[1]}; c.out = a1; FP_TRY(run_conv(ctx, c, s, sk, N));
    c = Conv2dCall{a1, N, 40, 40, &t[2]}; c.out = tA; FP_TRY(run_conv(ctx, c, s, sk, N));
    c = Conv2dCall{tA, N, 40, 40, &t[3]}; c.res = a1; c.out = a2; FP_TRY(run_conv(ctx, c, s, sk, N));
    c = Conv2dCall{a2, N, 40, 40, &t[4]}; c.out = tA; FP_TRY(run_conv(ctx, c, s, sk, N));
    c = Conv2dCall{tA, N, 40, 40, &t[5]}; c.res = a2; c.out = ab0; c.out_ld = 256; FP_TRY(run_conv(ctx, c, s, sk, N));
    hipLaunchKernelGGL(broadcast_side_b_kernel, dim3((unsigned)(((long long)N * 1600 * 16 + 255) / 256)), dim3(256), 0, s, sb->feat, *sb, h0, N, ab0);
    FP_CHECK_HIP(hipGetLastError());
  } else if (ab && ab->fan) {
    // encodeA / encoderA, one chain per side; side `h` owns images [h N, h N + N) of every buffer (and its own half of the split-K scratch)
    for (int h = 0; h < 2; ++h) {
      hipStream_t st = h == 0 ? s : ab->stream_for(0);
      float *skh = sk ? sk + (size_t)h * 4 * N * 1600 * 128 : nullptr;
      const size_t o80 = (size_t)h * N * 80 * 80 * 64, o40 = (size_t)h * N * 1600 * 128;
      c = Conv2dCall{h == 0 ? xA : xB, N, 160, 160, &t[0]}; c.out = a0 + o80; FP_TRY(run_conv(ctx, c, st, skh, N));
      c = Conv2dCall{a0 + o80, N, 80, 80, &t[1]}; c.out = a1 + o40; FP_TRY(run_conv(ctx, c, st, skh, N));
      c = Conv2dCall{a1 + o40, N, 40, 40, &t[2]}; c.out = tA + o40; FP_TRY(run_conv(ctx, c, st, skh, N));
      c = Conv2dCall{tA + o40, N, 40, 40, &t[3]}; c.res = a1 + o40; c.out = a2 + o40; FP_TRY(run_conv(ctx, c, st, skh, N));
      c = Conv2dCall{a2 + o40, N, 40, 40, &t[4]}; c.out = tA + o40; FP_TRY(run_conv(ctx, c, st, skh, N));
      // the channel concat cat((a,b),1): side A -> channels [0,128), side B -> [128,256) of the same rows
      c = Conv2dCall{tA + o40, N, 40, 40, &t[5]}; c.res = a2 + o40; c.out = ab0; c.out_ld = 256;
      if (h == 1) c.split_m = 0, c.coff_hi = 128;
      FP_TRY(run_conv(ctx, c, st, skh, N));
    }
    FP_TRY(ab->join());
  } else {
  // encodeA / encoderA on cat([A,B],0)
  // (A and B halves of the net tensor may come from different places when the batch is processed in chunks)
  c = Conv2dCall{xA, N, 160, 160, &t[0]}; c.out = a0; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{xB, N, 160, 160, &t[0]}; c.out = a0 + (size_t)N * 80 * 80 * 64; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{a0, (int)n2, 80, 80, &t[1]}; c.out = a1; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{a1, (int)n2, 40, 40, &t[2]}; c.out = tA; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{tA, (int)n2, 40, 40, &t[3]}; c.res = a1; c.out = a2; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{a2, (int)n2, 40, 40, &t[4]}; c.out = tA; FP_TRY(run_conv(ctx, c, s, sk, N));
  // last conv of encodeA writes the channel-concat cat((a,b),1) directly: image n<N -> channels [0,128), n>=N -> [128,256)
  c = Conv2dCall{tA, (int)n2, 40, 40, &t[5]}; c.res = a2; c.out = ab0; c.out_ld = 256; c.split_m = N * 1600; c.coff_hi = 128;
  FP_TRY(run_conv(ctx, c, s, sk, N));
  }
  // encodeAB
  c = Conv2dCall{ab0, N, 40, 40, &t[6]}; c.out = tB; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{tB, N, 40, 40, &t[7]}; c.res = ab0; c.out = ab1; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{ab1, N, 40, 40, &t[8]}; c.out = tB; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{tB, N, 40, 40, &t[9]}; c.res = ab1; c.out = ab0; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{ab0, N, 40, 40, &t[10]}; c.out = c0; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{c0, N, 20, 20, &t[11]}; c.out = tC; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{tC, N, 20, 20, &t[12]}; c.res = c0; c.out = c1; FP_TRY(run_conv(ctx, c, s, sk, N));
  c = Conv2dCall{c1, N, 20, 20, &t[13]}; c.out = tC; FP_TRY(run_conv(ctx, c, s, sk, N));
  // reshape(bs,C,-1).permute(0,2,1) is the NHWC tensor itself; pos_embed.pe added in the epilogue
  c = Conv2dCall{tC, N, 20, 20, &t[14]}; c.res = c1; c.out = tok; c.post_add = net->pe; c.post_period = 400;
  FP_TRY(run_conv(ctx, c, s, sk, N));
  *tokens_out = tok;
  return FP_OK;
}

// The trunk of a batch as TWO half batches on two streams (from 48 hypotheses on; FP_TRUNK_STREAMS=1: one).  The halves are
// independent (nothing in the trunk couples hypotheses) and every 3x3 launch fills the chip alone - one workgroup per CU - so two
// streams do not run side by side: what they buy is the LAST ROUND of each launch.  At 252 hypotheses a launch is 788 or 1575 tiles on
// 256 CUs, 3.08 or 6.15 rounds, and the partial last round costs 3 - 8 % of it (as quarter tiles, round 1); with two half-batch launches
// queued on two streams the workgroups of the other half start on the CUs the last round leaves idle.  Results are those of the single
// batch bit for bit (a hypothesis' arithmetic does not depend on its batch).
int run_trunk_maybe_split(fp_ctx *ctx, const fp_net *net, const f16 *in, int s0, int NT, int N, f16 **tokens_out, hipStream_t s, StreamFanout *ab = nullptr,
                          const SharedB *sb = nullptr) {
  static const int n_streams = getenv("FP_TRUNK_STREAMS") ? atoi(getenv("FP_TRUNK_STREAMS")) : 2;
  const size_t img = (size_t)160 * 160 * 8;
  if (N < fp_trunk_split_min()) return run_trunk(ctx, net, in + s0 * img, in + ((size_t)NT + s0) * img, N, tokens_out, s, nullptr, ab, sb, s0);
  if (ab) FP_TRY(ab->join());          // (a batch that is cut in two by hypotheses keeps both sides of a half on one stream)
  TAKE(tok, f16, (size_t)N * 400 * 512);
  const int n_parts = std::min(n_streams, std::min(fp_ctx::NSIDE, std::max(2, N / 32)));
  StreamFanout fo(ctx, s, n_parts);
  f16 *t = nullptr;
  int rc = FP_OK;
  for (int k = 0, a0 = 0; k < n_parts && rc == FP_OK; ++k) {
    const int a1 = (int)((long long)N * (k + 1) / n_parts);
    rc = run_trunk(ctx, net, in + (s0 + a0) * img, in + ((size_t)NT + s0 + a0) * img, a1 - a0, &t, k == 0 ? s : fo.stream_for(k - 1), tok + (size_t)a0 * 400 * 512,
                   nullptr, sb, s0 + a0);
    a0 = a1;
  }
  const int rj = fo.join();            // on every path: the caller resets the arena, which the side streams may still be writing
  FP_TRY(rc);
  FP_TRY(rj);
  *tokens_out = tok;
  return FP_OK;
}

// q|k and transposed-V in-projections of up to two attention layers that read the SAME tokens (RefineNet's two heads;
// one for ScoreNet): one launch for every q / k block (fp16 rows [M][1024] = q | k per layer), one for the V^T images
int run_qkv(fp_ctx *ctx, const LinP *const *q, const LinP *const *k, const LinP *const *v, int n_layers, const f16 *tok, int N, f16 *const *qk,
            f16 *const *vt, hipStream_t s) {
  TokGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.in = tok;
  a.M = N * 400;
  a.tokens = 400;
  static const bool old_form = getenv("FP_QKV64") != nullptr;          // A/B knob: the 64-token kernel, one launch for q | k, one for v (bit-identical)
  if (old_form && N > tok_qkv_small_max()) {          // (the few-image form of one and two hypotheses has its own bits: the knob compares the two batch kernels)
    a.nblk = 2 * n_layers;
    for (int l = 0; l < n_layers; ++l) {
      a.blk[2 * l] = TokGemmBlock{q[l]->w, q[l]->bias, qk[l], 1024, 0, 0};
      a.blk[2 * l + 1] = TokGemmBlock{k[l]->w, k[l]->bias, qk[l], 1024, 512, 0};
    }
    FP_TRY(launch_tok_gemm(ctx, a, TG_EPI_ROWS, s));
    a.nblk = n_layers;
    for (int l = 0; l < n_layers; ++l) a.blk[l] = TokGemmBlock{v[l]->w, v[l]->bias, vt[l], 0, 0, 0};
    return launch_tok_gemm(ctx, a, TG_EPI_VT, s);
  }
  // ONE launch: every block of every layer over a resident 128-token tile (tok_qkv.hip)
  a.nblk = 3 * n_layers;
  for (int l = 0; l < n_layers; ++l) {
    a.blk[3 * l] = TokGemmBlock{q[l]->w, q[l]->bias, qk[l], 1024, 0, 0, 0};
    a.blk[3 * l + 1] = TokGemmBlock{k[l]->w, k[l]->bias, qk[l], 1024, 512, 0, 0};
    a.blk[3 * l + 2] = TokGemmBlock{v[l]->w, v[l]->bias, vt[l], 0, 0, 0, 1};
  }
  return launch_tok_qkv(ctx, a, s, N);
}

}  // namespace

extern "C" int fp_refine_forward(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_trans, float *d_rot,
                                 void *stream) {
  return fp_refine_forward_ab(ctx, net, d_net_in, N, d_trans, d_rot, (hipStream_t)stream, nullptr);
}

// `ab`: side B of the network input is being produced on ab->stream_for(0), side A on `s` (the fused passes of api.hip): run_trunk
// encodeA of `n_groups` observed crops (one per object), with the kernels the batch of `hyp` hypotheses takes for these layers (run_conv's
// `hyp`: the few-image forms are chosen by the pass, not by the launch) -> feat [n_groups][1600][128]
int fp_encode_side_b(fp_ctx *ctx, const fp_net *net, const f16 *xB, int n_groups, int hyp, f16 *feat, hipStream_t s) {
  const ConvW *t = net->trunk;
  const int G = n_groups;
  TAKE(a0, f16, (size_t)G * 80 * 80 * 64);
  TAKE(a1, f16, (size_t)G * 1600 * 128);
  TAKE(tA, f16, (size_t)G * 1600 * 128);
  TAKE(a2, f16, (size_t)G * 1600 * 128);
  Conv2dCall c;
  c = Conv2dCall{xB, G, 160, 160, &t[0]}; c.out = a0; FP_TRY(run_conv(ctx, c, s, nullptr, hyp));
  c = Conv2dCall{a0, G, 80, 80, &t[1]}; c.out = a1; FP_TRY(run_conv(ctx, c, s, nullptr, hyp));
  c = Conv2dCall{a1, G, 40, 40, &t[2]}; c.out = tA; FP_TRY(run_conv(ctx, c, s, nullptr, hyp));
  c = Conv2dCall{tA, G, 40, 40, &t[3]}; c.res = a1; c.out = a2; FP_TRY(run_conv(ctx, c, s, nullptr, hyp));
  c = Conv2dCall{a2, G, 40, 40, &t[4]}; c.out = tA; FP_TRY(run_conv(ctx, c, s, nullptr, hyp));
  c = Conv2dCall{tA, G, 40, 40, &t[5]}; c.res = a2; c.out = feat; FP_TRY(run_conv(ctx, c, s, nullptr, hyp));
  return FP_OK;
}

int fp_refine_forward_ab(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_trans, float *d_rot, hipStream_t s, StreamFanout *ab,
                         RefineTailArgs *tail, const SharedB *sb) {
  FP_REQUIRE(ctx && net && d_net_in && d_trans && d_rot, "fp_refine_forward: null argument");
  FP_REQUIRE(net->kind == FP_NET_REFINE, "fp_refine_forward: not a RefineNet");
  FP_REQUIRE(N >= 0, "fp_refine_forward: N<0");
  if (N == 0) return ab ? ab->join() : FP_OK;
  const int NT = N;                                   // whole batch
  const int CH = fp_hyp_chunk(NT);
  FP_TRY(fp_arena_ensure(ctx, fp_arena_inner_bytes(CH)));
  const size_t mark = ctx->arena.off;
  if (CH != NT) tail = nullptr;                       // (hypothesis chunks: the building-block launches)
  auto body = [&](int s0, int N) -> int {
    f16 *tok = nullptr;
    const f16 *in = (const f16 *)d_net_in;
    FP_TRY(run_trunk_maybe_split(ctx, net, in, s0, NT, N, &tok, s, CH == NT ? ab : nullptr, sb));
    const int M = N * 400;
    // The translation and rotation heads are independent transformer layers on the same tokens: each gets its own
    // buffers and its own stream, so the tail of one head's kernels overlaps the other's (their launches are 3.08 rounds of
    // workgroups each)
    float *outs[2] = {d_trans + (size_t)s0 * 3, d_rot + (size_t)s0 * net->heads[1].out_dim};
    f16 *qk[2], *vt[2], *att[2];
    float *gsum[2];
    for (int h = 0; h < 2; ++h) {
      TAKE(qk_, f16, (size_t)M * 1024);
      TAKE(vt_, f16, (size_t)N * 4 * 128 * 416);
      TAKE(att_, f16, (size_t)M * 512);
      TAKE(gsum_, float, (size_t)(M / 16) * 512);
      qk[h] = qk_, vt[h] = vt_, att[h] = att_, gsum[h] = gsum_;
    }
    // in-projections of BOTH heads: the tokens are staged once per workgroup for 512 output columns of either head
    const LinP *q_[2] = {&net->heads[0].q, &net->heads[1].q}, *k_[2] = {&net->heads[0].k, &net->heads[1].k}, *v_[2] = {&net->heads[0].v, &net->heads[1].v};
    ProfScope wall(ctx, s, "heads_wall", 0.0);      // first in-projection .. join of both heads, on the main stream: the heads' share of wall time
    FP_TRY(run_qkv(ctx, q_, k_, v_, 2, tok, N, qk, vt, s));
    static const bool serial_heads = getenv("FP_HEADS_SERIAL") != nullptr;      // A/B timing knob
    StreamFanout fo(ctx, s, serial_heads ? 1 : 2);
    for (int h = 0; h < 2; ++h) {
      const HeadW &H = net->heads[h];
      hipStream_t sh = h == 0 ? s : fo.stream_for(0);
      FP_TRY(launch_attention(ctx, qk[h], vt[h], N, 400, att[h], sh));
      // x1 = norm1(tok + out_proj(att)); ff = relu(linear1(x1)); norm2(x1 + linear2(ff)) summed over groups of 16 tokens - one launch,
      // x1 and ff stay in LDS (head_mlp.hip); then the mean over the 400 tokens + Linear(512, out_dim)
      HeadMlpArgs a;
      a.att = att[h];
      a.tok = tok;
      a.M = M;
      a.w_out = H.out.w, a.b_out = H.out.bias;
      a.w1 = H.ff1.w, a.b1 = H.ff1.bias;
      a.w2 = H.ff2.w, a.b2 = H.ff2.bias;
      a.g1 = H.ln1g, a.be1 = H.ln1b;
      a.gsum = gsum[h];
      FP_TRY(launch_head_mlp(ctx, a, sh));
      if (!tail) FP_TRY(launch_mean_head(gsum[h], 25, H.ln2g, H.ln2b, N, 400, H.hw, H.hb, H.out_dim, outs[h], sh));
    }
    FP_TRY(fo.join());
    if (tail) {
      for (int h = 0; h < 2; ++h) {
        const HeadW &H = net->heads[h];
        tail->partial[h] = gsum[h], tail->gam[h] = H.ln2g, tail->bet[h] = H.ln2b, tail->hw[h] = H.hw, tail->hb[h] = H.hb;
      }
      tail->nparts = 25, tail->T = 400, tail->rot_dim = net->heads[1].out_dim;
      tail->trans = outs[0], tail->rot = outs[1];
      FP_REQUIRE(net->heads[0].out_dim == 3, "refine tail: translation head of %d outputs", net->heads[0].out_dim);
      FP_TRY(launch_refine_tail(*tail, N, s));
    }
    return FP_OK;
  };
  // hypothesis chunks (FP_CHUNK, default: the whole batch in one pass) reuse the SAME arena addresses.  Chunking to keep
  // producer -> consumer tensors inside the 256 MiB Infinity Cache was measured and is slower (smaller grids, same traffic)
  int rc = FP_OK;
  if (ab && CH != NT) rc = ab->join();          // (hypothesis chunks: one chain)
  for (int s0 = 0; s0 < NT && rc == FP_OK; s0 += CH) {
    rc = body(s0, std::min(CH, NT - s0));
    ctx->arena.off = mark;
  }
  if (ab && rc != FP_OK) (void)ab->join();      // never leave the side stream unjoined
  return rc;
}

// Building block for the parity tests: one 512 -> 512 token Linear with one of the fused epilogues of tok_gemm.hip.
extern "C" int fp_token_linear_f16(fp_ctx *ctx, const void *d_in, int M, const float *h_weight, const float *h_bias, int epilogue, int relu,
                                   const void *d_res, const float *h_gamma, const float *h_beta, int tokens, void *d_out, void *stream) {
  FP_REQUIRE(ctx && d_in && h_weight && h_bias && d_out, "fp_token_linear_f16: null argument");
  FP_REQUIRE(epilogue >= TG_EPI_ROWS && epilogue <= TG_EPI_LNSUM + 4, "fp_token_linear_f16: epilogue %d unknown", epilogue);
  FP_CHECK_HIP(hipSetDevice(ctx->device));
  std::vector<f16> hw((size_t)512 * 512);
  pack_tok_weights(h_weight, hw.data());
  char *dev = nullptr;
  FP_CHECK_HIP(hipMalloc((void **)&dev, hw.size() * 2 + 3 * 512 * 4));
  float *d_bias = (float *)(dev + hw.size() * 2), *d_g = d_bias + 512, *d_b = d_g + 512;
  int rc = FP_OK;
  auto run = [&]() -> int {
    FP_CHECK_HIP(hipMemcpy(dev, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    FP_CHECK_HIP(hipMemcpy(d_bias, h_bias, 512 * 4, hipMemcpyHostToDevice));
    if (h_gamma) FP_CHECK_HIP(hipMemcpy(d_g, h_gamma, 512 * 4, hipMemcpyHostToDevice));
    if (h_beta) FP_CHECK_HIP(hipMemcpy(d_b, h_beta, 512 * 4, hipMemcpyHostToDevice));
    TokGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const f16 *)d_in;
    a.M = M;
    a.nblk = 1;
    a.tokens = tokens;
    a.blk[0] = TokGemmBlock{(const f16 *)dev, d_bias, d_out, 512, 0, relu};
    a.res = (const f16 *)d_res;
    a.gamma = h_gamma ? d_g : nullptr;
    a.beta = h_beta ? d_b : nullptr;
    a.gsum = (float *)d_out;
    if (epilogue > TG_EPI_LNSUM) {          // 4 / 5: rows / transposed V image through the 128-token kernel of the in-projections (tok_qkv.hip); 6 / 7: its few-image form
      a.blk[0].vt = ((epilogue - TG_EPI_LNSUM) & 1) == 0;
      FP_TRY(launch_tok_qkv(ctx, a, (hipStream_t)stream, epilogue > TG_EPI_LNSUM + 2 ? 1 : 0));
    } else {
      FP_TRY(launch_tok_gemm(ctx, a, epilogue, (hipStream_t)stream));
    }
    FP_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FP_OK;
  };
  rc = run();
  (void)hipFree(dev);
  return rc;
}

// Building block for the parity tests: the fused post-attention part of one transformer head (head_mlp.hip).
extern "C" int fp_head_mlp_f16(fp_ctx *ctx, const void *d_att, const void *d_tok, int M, const float *h_w_out, const float *h_b_out,
                               const float *h_gamma1, const float *h_beta1, const float *h_w1, const float *h_b1, const float *h_w2,
                               const float *h_b2, float *d_gsum, void *stream) {
  FP_REQUIRE(ctx && d_att && d_tok && h_w_out && h_b_out && h_gamma1 && h_beta1 && h_w1 && h_b1 && h_w2 && h_b2 && d_gsum, "fp_head_mlp_f16: null argument");
  FP_CHECK_HIP(hipSetDevice(ctx->device));
  const size_t wh = (size_t)512 * 512;
  std::vector<f16> hw(3 * wh);
  pack_tok_weights(h_w_out, hw.data());
  pack_tok_weights(h_w1, hw.data() + wh);
  pack_tok_weights(h_w2, hw.data() + 2 * wh);
  std::vector<float> hv(5 * 512);
  const float *src[5] = {h_b_out, h_b1, h_b2, h_gamma1, h_beta1};
  for (int i = 0; i < 5; ++i) memcpy(hv.data() + i * 512, src[i], 512 * sizeof(float));
  char *dev = nullptr;
  FP_CHECK_HIP(hipMalloc((void **)&dev, hw.size() * 2 + hv.size() * 4));
  auto run = [&]() -> int {
    FP_CHECK_HIP(hipMemcpy(dev, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    float *dv = (float *)(dev + hw.size() * 2);
    FP_CHECK_HIP(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    HeadMlpArgs a;
    a.att = (const f16 *)d_att;
    a.tok = (const f16 *)d_tok;
    a.M = M;
    a.w_out = (const f16 *)dev, a.w1 = (const f16 *)dev + wh, a.w2 = (const f16 *)dev + 2 * wh;
    a.b_out = dv, a.b1 = dv + 512, a.b2 = dv + 1024, a.g1 = dv + 1536, a.be1 = dv + 2048;
    a.gsum = d_gsum;
    FP_TRY(launch_head_mlp(ctx, a, (hipStream_t)stream));
    FP_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FP_OK;
  };
  const int rc = run();
  (void)hipFree(dev);
  return rc;
}

extern "C" int fp_net_tokens(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, void *d_tokens, void *stream) {
  FP_REQUIRE(ctx && net && d_net_in && d_tokens, "fp_net_tokens: null argument");
  FP_REQUIRE(N >= 0, "fp_net_tokens: N<0");
  if (N == 0) return FP_OK;
  hipStream_t s = (hipStream_t)stream;
  FP_TRY(fp_arena_ensure(ctx, fp_arena_inner_bytes(N)));
  const size_t mark = ctx->arena.off;
  f16 *tok = nullptr;
  const f16 *in = (const f16 *)d_net_in;
  int rc = run_trunk(ctx, net, in, in + (size_t)N * 160 * 160 * 8, N, &tok, s);
  if (rc == FP_OK && hipMemcpyAsync(d_tokens, tok, (size_t)N * 400 * 512 * sizeof(f16), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    fp_set_error("fp_net_tokens: copy failed");
    rc = FP_EHIP;
  }
  ctx->arena.off = mark;
  return rc;
}

extern "C" int fp_score_features(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_feats, void *stream) {
  return fp_score_features_ab(ctx, net, d_net_in, N, d_feats, (hipStream_t)stream, nullptr);
}

int fp_score_features_ab(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_feats, hipStream_t s, StreamFanout *ab, int feat_ld,
                         const float *d_poses) {
  FP_REQUIRE(ctx && net && d_net_in && d_feats, "fp_score_features: null argument");
  FP_REQUIRE(net->kind == FP_NET_SCORE, "fp_score_features: not a ScoreNetMultiPair");
  FP_REQUIRE(N >= 0, "fp_score_features: N<0");
  if (N == 0) return ab ? ab->join() : FP_OK;
  const int NT = N;
  const int CH = fp_hyp_chunk(NT);
  FP_TRY(fp_arena_ensure(ctx, fp_arena_inner_bytes(CH)));
  const size_t mark = ctx->arena.off;
  auto body = [&](int s0, int N) -> int {
    f16 *tok = nullptr;
    const f16 *in = (const f16 *)d_net_in;
    FP_TRY(run_trunk_maybe_split(ctx, net, in, s0, NT, N, &tok, s, CH == NT ? ab : nullptr));
    const int M = N * 400;
    TAKE(qk, f16, (size_t)M * 1024);
    TAKE(vt, f16, (size_t)N * 4 * 128 * 416);
    TAKE(att, f16, (size_t)M * 512);
    const LinP *q_[1] = {&net->att_q}, *k_[1] = {&net->att_k}, *v_[1] = {&net->att_v};
    f16 *qk_[1] = {qk}, *vt_[1] = {vt};
    ProfScope wall(ctx, s, "heads_wall", 0.0);
    FP_TRY(run_qkv(ctx, q_, k_, v_, 1, tok, N, qk_, vt_, s));
    FP_TRY(launch_attention(ctx, qk, vt, N, 400, att, s));
    // mean over tokens commutes with out_proj (score_network.py:73-74): both in one launch
    FP_TRY(launch_score_feat(att, N, 400, net->att_out.wt, net->att_out.b, d_feats + (size_t)s0 * feat_ld, feat_ld, d_poses ? d_poses + (size_t)s0 * 16 : nullptr, s));
    return FP_OK;
  };
  int rc = FP_OK;
  if (ab && CH != NT) rc = ab->join();
  for (int s0 = 0; s0 < NT && rc == FP_OK; s0 += CH) {
    rc = body(s0, std::min(CH, NT - s0));
    ctx->arena.off = mark;
  }
  if (ab && rc != FP_OK) (void)ab->join();
  return rc;
}

int fp_score_tail_impl(fp_ctx *ctx, const fp_net *net, const float *d_feats, int feat_ld, int groups, int L, const ScoreTailOut &o, hipStream_t s) {
  FP_REQUIRE(ctx && net && d_feats && o.logits, "fp_score_tail: null argument");
  FP_REQUIRE(net->kind == FP_NET_SCORE, "fp_score_tail: not a ScoreNetMultiPair");
  FP_REQUIRE(groups >= 0 && L >= 1, "fp_score_tail: bad groups/L");
  if (groups == 0) return FP_OK;
  const int M = groups * L;
  FP_REQUIRE(groups <= FP_TAIL_MAX_GROUPS, "fp_score_tail: %d groups (at most %d)", groups, FP_TAIL_MAX_GROUPS);
  FP_TRY(fp_arena_ensure(ctx, (size_t)M * (1024 * 4 + 4 * 8) + (1 << 20)));
  const size_t mark = ctx->arena.off;
  auto body = [&]() -> int {
    TAKE(qk, float, (size_t)M * 1024);
    TAKE(sv, double, (size_t)M * 4);
    // two launches: q | k rows + the folded value scalars; logits (+ scores, argmax, the winner's pose) - score_tail.hip
    return launch_score_tail(d_feats, feat_ld, net->tail_wqk_t, net->tail_bqk, net->tail_u, net->tail_c, net->tail_b_eff, groups, L, qk, sv, ctx->tail_counter, o, s);
  };
  int rc = body();
  ctx->arena.off = mark;
  return rc;
}

extern "C" int fp_score_tail(fp_ctx *ctx, const fp_net *net, const float *d_feats, int groups, int L, float *d_logits,
                             int32_t *d_argmax, void *stream) {
  ScoreTailOut o;
  o.logits = d_logits;
  o.argmax = d_argmax;
  return fp_score_tail_impl(ctx, net, d_feats, 512, groups, L, o, (hipStream_t)stream);
}

extern "C" int fp_score_tail_scores(fp_ctx *ctx, const fp_net *net, const float *d_feats, int feat_ld, int groups, int L, float score_offset,
                                    float *d_logits, float *d_scores, int32_t *d_argmax, void *stream) {
  ScoreTailOut o;
  o.logits = d_logits;
  o.argmax = d_argmax;
  o.scores = d_scores;
  o.score_offset = score_offset;
  return fp_score_tail_impl(ctx, net, d_feats, feat_ld, groups, L, o, (hipStream_t)stream);
}

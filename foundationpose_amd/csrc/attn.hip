// Transformer-head kernels for gfx950: fused 400-token multi-head attention on MFMA, LayerNorm,
// token-mean + output heads, the cross-hypothesis score tail, argmax and the pose update.
//
// Reference: nn.TransformerEncoderLayer / nn.MultiheadAttention as used by
// learning/models/refine_network.py:56-70,88-91 and score_network.py:53-54,73-88;
// pose update: learning/training/predict_pose_refine.py:195-231 + pytorch3d so3_exp_map +
// src/Utils.py:848-855.
#include "common.h"

#define AT_WAVES 10
#define AT_THREADS (AT_WAVES * 64)
#define AT_DH 128
#define AT_KLD 136   // halfs per K row in LDS (128 + 8 pad)  -> 272 B
#define AT_TP 416    // padded token count of the transposed V image
#define AT_VLD 424   // halfs per Vt row in LDS (416 + 8 pad)  -> 848 B
#define AT_MAXT 400

// One workgroup = one (hypothesis, head, 160-query block); each of the 10 waves owns 16 queries.
//   staging: K (400 x 128, 100 KB) and the low half of V^T (64 x 416, 52 KB) arrive by LDS-DMA
//            (global_load_lds_dwordx4: every load of the tile in flight at once, no VGPRs); both images are
//            lane-linear in LDS with the XOR swizzle applied on the SOURCE address (K: chunk ^ (key&15),
//            V^T: chunk ^ ((d>>2)&3)) so the fragment reads are bank-conflict free.
//   phase 1: S^T = K Q^T with v_mfma_f32_16x16x32_f16 (A = K tile from LDS, B = Q fragments held in
//            registers) -> the lane owning query column q holds 4 keys per 16-key tile; the whole
//            400-key row (100 fp32) stays in registers, softmax needs two xor-shuffles.
//   phase 2: O = P V.  The S^T accumulator layout IS the A-operand layout of the next MFMA when two
//            key tiles are paired per k-step with the k order (tile0: 4g+0..3, tile1: 4g+0..3); V is
//            consumed from the transposed image so its B fragments are two 8-byte reads.  The high half
//            of V^T is DMA'd over the (dead) K image while the low half is being multiplied.
#define AT_VROW 416                    // halfs per V^T row (= AT_TP), 52 16-byte chunks
#define AT_KBYTES (400 * 256)          // K image
#define AT_VHALF (64 * AT_VROW * 2)    // one half of V^T

__device__ __forceinline__ void at_glds16(const f16 *g, f16 *l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

__global__ __launch_bounds__(AT_THREADS) void attention_kernel(const f16 *__restrict__ qk, const f16 *__restrict__ vt, int T,
                                                               f16 *__restrict__ out, const f16 *__restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  f16 *ks = smem;                        // [400][128] swizzled; later V^T high half [64][416]
  f16 *vlo = smem + AT_KBYTES / 2;       // V^T low half [64][416]
  // 1-D grid, XCD-aware order: the query blocks of one (hypothesis, head) share K and V -> same XCD L2
  const int nqb = (T + AT_WAVES * 16 - 1) / (AT_WAVES * 16);
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = L % nqb, h = (L / nqb) & 3, b = L / (nqb * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, g = lane >> 4;
  const int ntile = (T + 15) / 16;  // key tiles (25)
  const size_t rowbase = (size_t)b * T;
  const f16 *vsrc = vt + ((size_t)b * 4 + h) * AT_DH * AT_TP;

  // ---- issue every K and V^T(low) DMA ----
  for (int c0 = wave * 64; c0 < ntile * 256; c0 += AT_THREADS) {      // 16 chunks per key row
    const int c = c0 + lane, key = c >> 4, chp = c & 15;
    const f16 *src = key < T ? qk + (rowbase + key) * 1024 + 512 + h * AT_DH + ((chp ^ (key & 15)) * 8) : zero_page;
    at_glds16(src, ks + (size_t)c0 * 8);
  }
  auto vstage = [&](int half, f16 *dst) {
    for (int c0 = wave * 64; c0 < 64 * 52; c0 += AT_THREADS) {         // 52 chunks per row, 64 rows = 3328 = 52 x 64
      const int c = c0 + lane, dl = c / 52, chp = c - dl * 52, d = half * 64 + dl;
      at_glds16(vsrc + (size_t)d * AT_TP + ((chp ^ ((d >> 2) & 3)) * 8), dst + (size_t)c0 * 8);
    }
  };
  vstage(0, vlo);
  // ---- Q fragments: B operand, lane (q = lq, g) holds Q[q][32s + 8g .. +7] ----
  const int q = qb * (AT_WAVES * 16) + wave * 16 + lq;
  const bool qvalid = q < T;
  half8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qvalid) v = *reinterpret_cast<const uint4 *>(qk + (rowbase + q) * 1024 + h * AT_DH + s * 32 + g * 8);
    qf[s] = *reinterpret_cast<half8 *>(&v);
  }
  __syncthreads();   // hipcc drains vmcnt(0) before the barrier: K, V^T(low) and Q have landed

  // ---- phase 1: S^T tiles ----
  floatx4 st[26];
#pragma unroll
  for (int t = 0; t < 26; ++t) st[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    if (t < ntile) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        half8 kf = *reinterpret_cast<const half8 *>(&ks[(t * 16 + lq) * 128 + (((s * 4 + g) ^ lq) * 8)]);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], st[t], 0, 0, 0);
      }
    }
  }
  // ---- softmax over keys for query column lq (values spread over the 4 lane groups) ----
  const float scale = 0.08838834764831845f;  // 1/sqrt(128)
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < 25; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int key = t * 16 + g * 4 + r;
      if (key < T) mx = fmaxf(mx, st[t][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 25; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int key = t * 16 + g * 4 + r;
      float e = (key < T) ? __expf((st[t][r] - mx) * scale) : 0.f;
      st[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.f / sum;
  __syncthreads();  // everyone is done reading K
  vstage(1, ks);    // V^T high half over the K image, in flight under phase 2a

  // ---- phase 2: O = P V, 13 k-steps of 32 keys (two key tiles each), 8 n-tiles of 16 dims ----
  floatx4 oacc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) oacc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
  half8 pfr[13];
#pragma unroll
  for (int s = 0; s < 13; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pfr[s][r] = (f16)st[2 * s][r];
      pfr[s][4 + r] = (f16)st[2 * s + 1][r];
    }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const f16 *vb = half ? ks : vlo;
    if (half) __syncthreads();       // V^T(high) landed (vmcnt drained before the barrier)
#pragma unroll
    for (int s = 0; s < 13; ++s) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int d = half * 64 + n * 16 + lq, sw = (d >> 2) & 3;
        const f16 *vrow = vb + (n * 16 + lq) * AT_VROW + (g & 1) * 4;
        half4 v0 = *reinterpret_cast<const half4 *>(vrow + (((4 * s + (g >> 1)) ^ sw) * 8));
        half4 v1 = *reinterpret_cast<const half4 *>(vrow + (((4 * s + 2 + (g >> 1)) ^ sw) * 8));
        half8 vf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vf[r] = v0[r];
          vf[4 + r] = v1[r];
        }
        oacc[half * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfr[s], vf, oacc[half * 4 + n], 0, 0, 0);
      }
    }
  }
  // O accumulator: col = lq -> dim n*16+lq, row = 4g + r -> query (wave*16 + 4g + r).  inv belongs to
  // the query on column lq of phase 1, i.e. query index lq; fetch the right one per row.
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = 4 * g + r;
    const float invq = __shfl(inv, qrow);  // lane qrow (g=0 copy) holds 1/sum of query qrow
    const int qq = qb * (AT_WAVES * 16) + wave * 16 + qrow;
    if (qq < T) {
#pragma unroll
      for (int n = 0; n < 8; ++n) out[(rowbase + qq) * 512 + h * AT_DH + n * 16 + lq] = (f16)(oacc[n][r] * invq);
    }
  }
}

int launch_attention(fp_ctx *ctx, const f16 *qk, const f16 *vt, int B, int T, f16 *out, hipStream_t s) {
  FP_REQUIRE(T > 0 && T <= AT_MAXT, "attention: T=%d must be in [1,%d]", T, AT_MAXT);
  if (B == 0) return FP_OK;
  const size_t lds = AT_KBYTES + AT_VHALF;
  FP_CHECK_HIP(hipFuncSetAttribute((const void *)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(((T + AT_WAVES * 16 - 1) / (AT_WAVES * 16)) * 4 * B);
  ProfScope ps(ctx, s, "attention", 4.0 * B * 4 * (double)T * T * AT_DH);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(AT_THREADS), lds, s, qk, vt, T, out, (const f16 *)ctx->zero_page);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LayerNorm over 512 features, one wave per row (8 values per lane), fp32 in -> fp16 out.
__device__ __forceinline__ void load8(const float *p, float *v) {
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const f16 *p, float *v) {
  const half8 h = *reinterpret_cast<const half8 *>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}

template <typename TI>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI *__restrict__ x, const float *__restrict__ gam,
                                                        const float *__restrict__ bet, int M, f16 *__restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float v[8];
  load8(x + (size_t)row * 512 + lane * 8, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.f / 512.f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] -= mean;
    sq += v[i] * v[i];
  }
  const float rstd = rsqrtf(wave_sum(sq) * (1.f / 512.f) + 1e-5f);
  half8 hv;
#pragma unroll
  for (int i = 0; i < 8; ++i) hv[i] = (f16)(v[i] * rstd * gam[lane * 8 + i] + bet[lane * 8 + i]);
  *reinterpret_cast<half8 *>(out + (size_t)row * 512 + lane * 8) = hv;
}

int launch_layernorm(const float *x, const float *g, const float *b, int M, f16 *out, hipStream_t s) {
  if (M == 0) return FP_OK;
  hipLaunchKernelGGL(layernorm_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, M, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int launch_layernorm_h(const f16 *x, const float *g, const float *b, int M, f16 *out, hipStream_t s) {
  if (M == 0) return FP_OK;
  hipLaunchKernelGGL(layernorm_kernel<f16>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, M, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// Final LayerNorm + mean over the T tokens of one hypothesis + Linear(512 -> out_dim<=6):
// mean_t(Linear(LN(x_t))) == Linear(mean_t LN(x_t))  (refine_network.py:90-91).
// Two launches so that 252 hypotheses fill the chip: (1) LNP_SPLIT workgroups per hypothesis sum the
// normalised rows of their token range -> partial[b][part][512]; (2) one small workgroup per hypothesis
// adds the partials in a fixed order (deterministic), applies gamma/beta and the output Linear.
#define LNP_SPLIT 8
template <typename TI>
__global__ __launch_bounds__(256) void ln_partial_kernel(const TI *__restrict__ x, int T, float *__restrict__ partial) {
  __shared__ float part[4][512];
  const int b = blockIdx.x / LNP_SPLIT, q = blockIdx.x % LNP_SPLIT, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (T + LNP_SPLIT - 1) / LNP_SPLIT, t0 = q * per, t1 = min(T, t0 + per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = t0 + wave; t < t1; t += 4) {
    float v[8];
    load8(x + ((size_t)b * T + t) * 512 + lane * 8, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.f / 512.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] -= mean;
      sq += v[i] * v[i];
    }
    const float rstd = rsqrtf(wave_sum(sq) * (1.f / 512.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += v[i] * rstd;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[wave][lane * 8 + i] = acc[i];
  __syncthreads();
  for (int f = threadIdx.x; f < 512; f += 256)
    partial[((size_t)b * LNP_SPLIT + q) * 512 + f] = (part[0][f] + part[1][f]) + (part[2][f] + part[3][f]);
}

__global__ __launch_bounds__(128) void mean_head_kernel(const float *__restrict__ partial, const float *__restrict__ gam,
                                                        const float *__restrict__ bet, int T, const float *__restrict__ hw,
                                                        const float *__restrict__ hb, int out_dim, float *__restrict__ out) {
  __shared__ float meanv[512];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int f = threadIdx.x; f < 512; f += 128) {
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < LNP_SPLIT; ++q) m += partial[((size_t)b * LNP_SPLIT + q) * 512 + f];
    meanv[f] = m * (1.f / (float)T) * gam[f] + bet[f];
  }
  __syncthreads();
  for (int o = wave; o < out_dim; o += 2) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += meanv[lane * 8 + i] * hw[(size_t)o * 512 + lane * 8 + i];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)b * out_dim + o] = s + hb[o];
  }
}

int launch_ln_mean_head(const float *x, const float *g, const float *b, int Bn, int T, const float *hw, const float *hb, int out_dim,
                        float *out, float *scratch, hipStream_t s) {
  if (Bn == 0) return FP_OK;
  hipLaunchKernelGGL(ln_partial_kernel<float>, dim3(Bn * LNP_SPLIT), dim3(256), 0, s, x, T, scratch);
  hipLaunchKernelGGL(mean_head_kernel, dim3(Bn), dim3(128), 0, s, scratch, g, b, T, hw, hb, out_dim, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int launch_ln_mean_head_h(const f16 *x, const float *g, const float *b, int Bn, int T, const float *hw, const float *hb, int out_dim,
                          float *out, float *scratch, hipStream_t s) {
  if (Bn == 0) return FP_OK;
  hipLaunchKernelGGL(ln_partial_kernel<f16>, dim3(Bn * LNP_SPLIT), dim3(256), 0, s, x, T, scratch);
  hipLaunchKernelGGL(mean_head_kernel, dim3(Bn), dim3(128), 0, s, scratch, g, b, T, hw, hb, out_dim, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// mean over tokens of an fp16 (Bn*T, 512) tensor -> fp32 (Bn, 512)   (score_network.py:74)
__global__ __launch_bounds__(256) void token_mean_kernel(const f16 *__restrict__ x, int T, float *__restrict__ out) {
  __shared__ float part[4][512];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = wave; t < T; t += 4) {
    half8 v = *reinterpret_cast<const half8 *>(x + ((size_t)b * T + t) * 512 + lane * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[wave][lane * 8 + i] = acc[i];
  __syncthreads();
  for (int f = threadIdx.x; f < 512; f += 256)
    out[(size_t)b * 512 + f] = ((part[0][f] + part[1][f]) + (part[2][f] + part[3][f])) * (1.f / (float)T);
}

int launch_token_mean(const f16 *x, int Bn, int T, float *out, hipStream_t s) {
  if (Bn == 0) return FP_OK;
  hipLaunchKernelGGL(token_mean_kernel, dim3(Bn), dim3(256), 0, s, x, T, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// y[m][n] = sum_k x[m][k] w[n][k] + b[n], fp32, one wave per output element group (tiny matrices:
// the per-object score tail, 0.66 GFLOP per 252 hypotheses).
__global__ __launch_bounds__(256) void small_linear_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias, int M, int K, int N,
                                                           float *__restrict__ out) {
  const int m = blockIdx.y;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  const float *xr = x + (size_t)m * K, *wr = w + (size_t)n * K;
  float s = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 a = *reinterpret_cast<const float4 *>(xr + k);
    const float4 c = *reinterpret_cast<const float4 *>(wr + k);
    s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  s = wave_sum(s);
  if (lane == 0) out[(size_t)m * N + n] = s + (bias ? bias[n] : 0.f);
}

int launch_small_linear(const float *x, const float *w, const float *b, int M, int K, int N, float *out, hipStream_t s) {
  FP_REQUIRE(K % 4 == 0, "small_linear: K=%d must be a multiple of 4", K);
  if (M == 0 || N == 0) return FP_OK;
  hipLaunchKernelGGL(small_linear_kernel, dim3((N + 3) / 4, M), dim3(256), 0, s, x, w, b, M, K, N, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// Cross-hypothesis self-attention (score_network.py:83): qkv (groups*L, 1536) fp32 -> out (groups*L, 512).
// One workgroup per (query, group); 4 waves = 4 heads; scores held in LDS (L <= 4096).
#define CA_MAXL 4096
__global__ __launch_bounds__(256) void cross_attention_kernel(const float *__restrict__ qkv, int L, float *__restrict__ out) {
  __shared__ float sc[4][CA_MAXL];
  const int i = blockIdx.x, grp = blockIdx.y, hd = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *base = qkv + (size_t)grp * L * 1536;
  const float *qr = base + (size_t)i * 1536 + hd * 128;
  const float q0 = qr[lane * 2], q1 = qr[lane * 2 + 1];
  const float scale = 0.08838834764831845f;
  float mx = -3.0e38f;
  for (int j = 0; j < L; ++j) {
    const float *kr = base + (size_t)j * 1536 + 512 + hd * 128;
    float s = wave_sum(q0 * kr[lane * 2] + q1 * kr[lane * 2 + 1]) * scale;
    if (lane == 0) sc[hd][j] = s;
    mx = fmaxf(mx, s);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) {
    float e = __expf(sc[hd][j] - mx);
    sc[hd][j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  float o0 = 0.f, o1 = 0.f;
  for (int j = 0; j < L; ++j) {
    const float *vr = base + (size_t)j * 1536 + 1024 + hd * 128;
    const float pj = sc[hd][j];
    o0 += pj * vr[lane * 2];
    o1 += pj * vr[lane * 2 + 1];
  }
  const float inv = 1.f / sum;
  float *orow = out + ((size_t)grp * L + i) * 512 + hd * 128;
  orow[lane * 2] = o0 * inv;
  orow[lane * 2 + 1] = o1 * inv;
}

int launch_cross_attention(const float *qkv, int groups, int L, float *out, hipStream_t s) {
  FP_REQUIRE(L >= 1 && L <= CA_MAXL, "score tail: L=%d must be in [1,%d]", L, CA_MAXL);
  if (groups == 0) return FP_OK;
  hipLaunchKernelGGL(cross_attention_kernel, dim3(L, groups), dim3(256), 0, s, qkv, L, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// per-group argmax, first maximum wins (torch.argmax tie rule on a 1-D tensor)
__global__ __launch_bounds__(64) void argmax_kernel(const float *__restrict__ logits, int L, int32_t *__restrict__ out) {
  const int grp = blockIdx.x, lane = threadIdx.x;
  float best = -3.0e38f;
  int bi = 0x7fffffff;
  for (int j = lane; j < L; j += 64) {
    float v = logits[(size_t)grp * L + j];
    if (v > best || (v == best && j < bi)) {
      best = v;
      bi = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o);
    int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (lane == 0) out[grp] = bi;
}

int launch_argmax(const float *logits, int groups, int L, int32_t *out, hipStream_t s) {
  if (groups == 0) return FP_OK;
  hipLaunchKernelGGL(argmax_kernel, dim3(groups), dim3(64), 0, s, logits, L, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// pose update (predict_pose_refine.py:195-231): float32, mirrors oracle/predict.py:pose_update
__global__ void pose_update_kernel(const float *__restrict__ poseA, const float *__restrict__ trans, const float *__restrict__ rot,
                                   int N, int rot_dim, int trans_tanh, float tn0, float tn1, float tn2, float rot_normalizer,
                                   float trans_scale, float *__restrict__ outp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  const float *A = poseA + (size_t)b * 16;
  float td[3] = {trans[b * 3], trans[b * 3 + 1], trans[b * 3 + 2]};
  if (trans_tanh) {
    td[0] = tanhf(td[0]) * tn0;
    td[1] = tanhf(td[1]) * tn1;
    td[2] = tanhf(td[2]) * tn2;
  }
  td[0] *= trans_scale;
  td[1] *= trans_scale;
  td[2] *= trans_scale;
  float R[9];  // rot_mat_delta (already transposed as in the reference)
  if (rot_dim == 3) {
    const float x = tanhf(rot[b * 3]) * rot_normalizer, y = tanhf(rot[b * 3 + 1]) * rot_normalizer,
                z = tanhf(rot[b * 3 + 2]) * rot_normalizer;
    const float nrm = fmaxf(x * x + y * y + z * z, 1e-4f);
    const float th = sqrtf(nrm), ith = 1.f / th;
    const float f1 = ith * sinf(th), f2 = ith * ith * (1.f - cosf(th));
    const float Kx[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float K2[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K2[r * 3 + c] = Kx[r * 3] * Kx[c] + Kx[r * 3 + 1] * Kx[3 + c] + Kx[r * 3 + 2] * Kx[6 + c];
    float E[9];
    for (int i = 0; i < 9; ++i) E[i] = f1 * Kx[i] + f2 * K2[i] + ((i % 4 == 0) ? 1.f : 0.f);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = E[c * 3 + r];  // .permute(0,2,1)
  } else {
    const float *d6 = rot + (size_t)b * 6;
    float a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
    float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float dp = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float b2[3] = {a2[0] - dp * b1[0], a2[1] - dp * b1[1], a2[2] - dp * b1[2]};
    float n2 = fmaxf(sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]), 1e-12f);
    b2[0] /= n2;
    b2[1] /= n2;
    b2[2] /= n2;
    float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    // rows (b1,b2,b3) then transposed
    for (int c = 0; c < 3; ++c) {
      R[c * 3 + 0] = b1[c];
      R[c * 3 + 1] = b2[c];
      R[c * 3 + 2] = b3[c];
    }
  }
  float *O = outp + (size_t)b * 16;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) O[r * 4 + c] = R[r * 3] * A[c] + R[r * 3 + 1] * A[4 + c] + R[r * 3 + 2] * A[8 + c];
    O[r * 4 + 3] = A[r * 4 + 3] + td[r];
  }
  O[12] = 0.f;
  O[13] = 0.f;
  O[14] = 0.f;
  O[15] = 1.f;
}

int launch_pose_update(const float *poseA, const float *trans, const float *rot, int N, int rot_dim, int trans_tanh, float tn0,
                       float tn1, float tn2, float rot_normalizer, float trans_scale, float *out, hipStream_t s) {
  FP_REQUIRE(rot_dim == 3 || rot_dim == 6, "pose_update: rot_dim must be 3 or 6");
  if (N == 0) return FP_OK;
  hipLaunchKernelGGL(pose_update_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poseA, trans, rot, N, rot_dim, trans_tanh, tn0, tn1, tn2,
                     rot_normalizer, trans_scale, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// The cross-hypothesis end of ScoreNetMultiPair (learning/models/score_network.py:73-90) in three launches, two of them the tail:
//
//   score_feat_kernel      per hypothesis: mean over the 400 tokens of the self-attention output + att.out_proj -> feature (512)
//                          (the mean commutes with the projection, score_network.py:73-74; this is what a rank all-gathers)
//   cross_qk_kernel        per row of the gathered features: q, k of att_cross (fp32 rows) and FOUR SCALARS s_j^h
//   cross_logit_kernel     per query: softmax_j(q_i^h . k_j^h / sqrt(128)) . s_j^h summed over the heads + b_eff -> logit; the workgroup
//                          that finishes a group last takes the argmax (first maximum wins, torch.argmax)
//
// The value path is folded at load time, in float64: behind the attention only `linear(out_proj(ctx))` follows (score_network.py:84-85),
// so   logit_i = lin.w . (W_o ctx_i + b_o) + lin.b = w_eff . ctx_i + b_eff,   w_eff = W_o^T lin.w,
// and  w_eff . ctx_i = sum_h sum_j p_ij^h (w_eff^h . v_j^h) = sum_h sum_j p_ij^h s_j^h,   s_j^h = u_h . f_j + c_h,  u_h = Wv_h^T w_eff^h:
// V (252 x 512), the out-projection and the final Linear never run - a 512-vector per head does.  All sums are float64 (the logits
// of a group differ by 1e-4 .. 1e-3 on top of O(1) common parts: fp32 summation noise would be a visible part of that).
// Until round 4 this was seven launches (token mean, out-projection, q/k/v projection, attention with one workgroup per query,
// out-projection, Linear, argmax): 0.18 - 0.3 ms replicated on every rank of a multi-GPU job.
#include "common.h"
#include "pose_math.h"

__device__ __forceinline__ double st_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double st_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// ---- feature: token mean (fp32 sums of fp16 values, eight waves x 50 tokens, fixed order) + out-projection (float64 sums) ----
// out: rows of `ld` floats; with `poses` the row is the all-gather record [feature 512 | pose 16] (dist.py) and the pose rides along
__global__ __launch_bounds__(512) void score_feat_kernel(const f16 *__restrict__ x, int T, const float *__restrict__ wt, const float *__restrict__ bias,
                                                         float *__restrict__ out, int ld, const float *__restrict__ poses) {
  __shared__ float part[8][512];
  __shared__ float mean[512];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const f16 *xb = x + (size_t)b * T * 512 + lane * 8;
  for (int t0 = wave; t0 < T; t0 += 32) {                 // four tokens in flight per wave
    half8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8 *>(xb + (size_t)min(t0 + 8 * u, T - 1) * 512);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (t0 + 8 * u < T) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (float)v[u][i];
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[wave][lane * 8 + i] = acc[i];
  __syncthreads();
  {
    const int f = tid;
    mean[f] = (((part[0][f] + part[1][f]) + (part[2][f] + part[3][f])) + ((part[4][f] + part[5][f]) + (part[6][f] + part[7][f]))) * (1.f / (float)T);
  }
  __syncthreads();
  // out[n] = sum_k wt[k][n] mean[k] + b[n]: a lane per output column (coalesced rows of the transposed weights), 16 rows in flight
  const int n = tid;
  double s = 0.0;
  float wv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) wv[u] = wt[(size_t)u * 512 + n];
  for (int k0 = 0; k0 < 512; k0 += 16) {
    float nx[16];
    const int k1 = min(k0 + 16, 512 - 16);
#pragma unroll
    for (int u = 0; u < 16; ++u) nx[u] = wt[(size_t)(k1 + u) * 512 + n];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (double)mean[k0 + u] * (double)wv[u];
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = nx[u];
  }
  out[(size_t)b * ld + n] = (float)(s + (double)bias[n]);
  if (poses && tid < 16) out[(size_t)b * ld + 512 + tid] = poses[(size_t)b * 16 + tid];
}

int launch_score_feat(const f16 *att, int Bn, int T, const float *wt, const float *bias, float *out, int ld, const float *poses, hipStream_t s) {
  if (Bn == 0) return FP_OK;
  FP_REQUIRE(ld >= 512 && (!poses || ld >= 528), "score features: row stride %d", ld);
  hipLaunchKernelGGL(score_feat_kernel, dim3(Bn), dim3(512), 0, s, att, T, wt, bias, out, ld, poses);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// ---- q | k rows (fp32) and the four value scalars (float64) of every feature row ----
// blockIdx.x < 4: 256 of the 1024 q | k columns, a lane per column, 4 rows per workgroup staged in LDS (broadcast reads), float64 sums,
// the next 16 weight rows requested while these 16 multiply; blockIdx.x == 4: s[row][h] = u_h . f_row + c_h, one wave per row.
#define CQ_ROWS 4
__global__ __launch_bounds__(256) void cross_qk_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ wt, const float *__restrict__ bias,
                                                       const double *__restrict__ u, const double *__restrict__ c, int M, float *__restrict__ qk,
                                                       double *__restrict__ sv) {
  __shared__ double xs[CQ_ROWS][512];               // the rows as float64 once: the inner loop converts only the weight
  const int m0 = blockIdx.y * CQ_ROWS, tid = threadIdx.x;
  for (int i = tid; i < CQ_ROWS * 512; i += 256) xs[i >> 9][i & 511] = (double)x[(size_t)min(m0 + (i >> 9), M - 1) * ldx + (i & 511)];
  __syncthreads();
  if (blockIdx.x == 4) {
    const int r = tid >> 6, lane = tid & 63;
    if (m0 + r >= M) return;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += u[h * 512 + k * 64 + lane] * xs[r][k * 64 + lane];
      s = st_wave_sum(s);
      if (lane == 0) sv[(size_t)(m0 + r) * 4 + h] = s + c[h];
    }
    return;
  }
  const int n = blockIdx.x * 256 + tid;
  double acc[CQ_ROWS] = {0.0, 0.0, 0.0, 0.0};
  float wv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) wv[q] = wt[(size_t)q * 1024 + n];
  for (int k0 = 0; k0 < 512; k0 += 16) {
    float nx[16];
    const int k1 = min(k0 + 16, 512 - 16);
#pragma unroll
    for (int q = 0; q < 16; ++q) nx[q] = wt[(size_t)(k1 + q) * 1024 + n];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const double w = (double)wv[q];
#pragma unroll
      for (int r = 0; r < CQ_ROWS; ++r) acc[r] += xs[r][k0 + q] * w;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) wv[q] = nx[q];
  }
  const double b = (double)bias[n];
#pragma unroll
  for (int r = 0; r < CQ_ROWS; ++r)
    if (m0 + r < M) qk[(size_t)(m0 + r) * 1024 + n] = (float)(acc[r] + b);
}

// ---- logits of CL_QB queries per workgroup (4 waves = 4 heads) + the group's argmax by the workgroup that finishes last ----
// A lane owns keys lane, lane + 64, ... (four in flight), reads each key row ONCE for the CL_QB queries and keeps per query a running
// (maximum, sum of exponentials, sum of exponentials x s_j); the 64 lanes' triples are merged at the end (exact up to float64 rounding,
// fixed order) - no score buffer, any L.
#define CL_QB 4
__global__ __launch_bounds__(256) void cross_logit_kernel(const float *__restrict__ qk, const double *__restrict__ sv, double b_eff, int L,
                                                          int *__restrict__ counter, ScoreTailOut o) {
  float *__restrict__ logits = o.logits;
  int32_t *__restrict__ argmax_out = o.argmax;
  __shared__ double qs[4][CL_QB][128];
  __shared__ double head_out[4][CL_QB];
  __shared__ int is_last;
  const int grp = blockIdx.y, i0 = blockIdx.x * CL_QB, hd = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *base = qk + (size_t)grp * L * 1024;
  const double *sb = sv + (size_t)grp * L * 4;
#pragma unroll
  for (int q = 0; q < CL_QB; ++q) {
    const float *qr = base + (size_t)min(i0 + q, L - 1) * 1024 + hd * 128;
    qs[hd][q][lane * 2] = (double)qr[lane * 2];
    qs[hd][q][lane * 2 + 1] = (double)qr[lane * 2 + 1];
  }
  __syncthreads();
  const double scale = 0.08838834764831845;      // 1 / sqrt(128)
  double mx[CL_QB], z[CL_QB], wsum[CL_QB];
#pragma unroll
  for (int q = 0; q < CL_QB; ++q) mx[q] = -1.0e300, z[q] = 0.0, wsum[q] = 0.0;
  for (int j0 = lane; j0 < L; j0 += 256) {
    const float4 *kr[4];
    double acc[4][CL_QB];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kr[u] = reinterpret_cast<const float4 *>(base + (size_t)min(j0 + 64 * u, L - 1) * 1024 + 512 + hd * 128);
#pragma unroll
      for (int q = 0; q < CL_QB; ++q) acc[u][q] = 0.0;
    }
#pragma unroll 2
    for (int d4 = 0; d4 < 32; ++d4) {
      float4 kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[u] = kr[u][d4];
#pragma unroll
      for (int q = 0; q < CL_QB; ++q) {
        const double qa = qs[hd][q][4 * d4], qb = qs[hd][q][4 * d4 + 1], qc = qs[hd][q][4 * d4 + 2], qd = qs[hd][q][4 * d4 + 3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u][q] += qa * (double)kv[u].x;
          acc[u][q] += qb * (double)kv[u].y;
          acc[u][q] += qc * (double)kv[u].z;
          acc[u][q] += qd * (double)kv[u].w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + 64 * u < L) {
        const double sj = sb[(size_t)(j0 + 64 * u) * 4 + hd];
#pragma unroll
        for (int q = 0; q < CL_QB; ++q) {
          const double s = acc[u][q] * scale;
          if (s > mx[q]) {                          // rescale the running sums to the new maximum
            const double r = exp(mx[q] - s);
            z[q] *= r, wsum[q] *= r, mx[q] = s;
          }
          const double e = exp(s - mx[q]);
          z[q] += e, wsum[q] += e * sj;
        }
      }
  }
#pragma unroll
  for (int q = 0; q < CL_QB; ++q) {
    const double m = st_wave_max(mx[q]);
    const double r = exp(mx[q] - m);              // (a lane without keys: exp(-1e300 - m) = 0)
    const double zt = st_wave_sum(z[q] * r), wt = st_wave_sum(wsum[q] * r);
    if (lane == 0) head_out[hd][q] = wt / zt;
  }
  __syncthreads();
  if (threadIdx.x < CL_QB && i0 + (int)threadIdx.x < L) {
    const int q = threadIdx.x;
    const float lg = (float)((((head_out[0][q] + head_out[1][q]) + head_out[2][q]) + head_out[3][q]) + b_eff);
    logits[(size_t)grp * L + i0 + q] = lg;
    if (o.scores) o.scores[(size_t)grp * L + i0 + q] = lg + o.score_offset;
  }
  if (!argmax_out) return;
  // the workgroup that arrives last at the group's counter sees every logit of the group (agent-scope release / acquire around the
  // counter, MI355X_MICROARCH.md "inter-workgroup visibility") and takes the argmax; it leaves the counter at 0 for the next call
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(counter + grp, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = done == (int)gridDim.x - 1;
    if (is_last) __hip_atomic_store(counter + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!is_last || hd != 0) return;
  float best = -3.0e38f;
  int bi = 0x7fffffff;
  for (int j = lane; j < L; j += 64) {
    const float v = __hip_atomic_load(logits + (size_t)grp * L + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v > best || (v == best && j < bi)) best = v, bi = j;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi;
  }
  if (lane == 0) {
    argmax_out[grp] = bi;
    if (o.poses) {                                   // tracking: the winner's pose and its pose @ get_tf_to_centered_mesh()
      const float *pp = o.poses + ((size_t)grp * L + bi) * 16;
      if (o.best_pose)
        for (int e = 0; e < 16; ++e) o.best_pose[(size_t)grp * 16 + e] = pp[e];
      if (o.best_centered) pose_of_mesh_one(pp, o.cneg, o.best_centered + (size_t)grp * 16);
    }
  }
}

int launch_score_tail(const float *feats, int feat_ld, const float *wqk_t, const float *bqk, const double *u, const double *c, double b_eff, int groups, int L,
                      float *qk, double *sv, int *counter, const ScoreTailOut &o, hipStream_t s) {
  FP_REQUIRE(o.logits, "score tail: no logits buffer");
  FP_REQUIRE(feat_ld >= 512, "score tail: feature row stride %d", feat_ld);
  FP_REQUIRE(!o.poses || o.argmax, "score tail: the winner's pose needs the argmax output");
  const int M = groups * L;
  if (M == 0) return FP_OK;
  hipLaunchKernelGGL(cross_qk_kernel, dim3(5, (M + CQ_ROWS - 1) / CQ_ROWS), dim3(256), 0, s, feats, feat_ld, wqk_t, bqk, u, c, M, qk, sv);
  hipLaunchKernelGGL(cross_logit_kernel, dim3((L + CL_QB - 1) / CL_QB, groups), dim3(256), 0, s, qk, sv, b_eff, L, counter, o);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

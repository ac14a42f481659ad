// The part of nn.TransformerEncoderLayer behind the attention core, for the two RefineNet heads (learning/models/
// refine_network.py:56-70,88-91; post-norm layer, dim_feedforward 512, ReLU) in ONE launch per head:
//
//   x1   = LayerNorm1(tok + att @ Wout^T + b_out)
//   ff   = ReLU(x1 @ W1^T + b1)
//   y    = LayerNorm2(x1 + ff @ W2^T + b2)           (gamma / beta of LayerNorm2 are applied behind the token mean)
//   gsum = sums of y over groups of 16 tokens         (norm2 feeds nothing but the token mean, refine_network.py:90-91)
//
// Until round 3 these were three launches of tok_gemm.hip (out-projection + LN, linear1, linear2 + LN sums): x1 was written once
// and read twice, ff written and read once - 515 MB of the 734 MB the three launches moved per head and pass at 252 hypotheses.
// Here a workgroup (8 waves) owns 64 tokens for the whole chain and the intermediates never leave the CU:
//   * two 64-KB tile buffers P and Q in the layout of tok_gemm.hip's activation tile (k segments of 128, 256-byte row segments,
//     16-byte chunk c of a row stored at c ^ (row & 15): conflict-free ds_read_b128).  At the start the attention output goes to P and
//     the residual tokens to Q by LDS-DMA, together; the first K loop starts when P has landed;
//   * K loop as in tok_gemm.hip: weights packed in MFMA-fragment order stream L2 -> registers (each wave its own 64 output columns,
//     2 fragments per k-step of 16, prefetched 4 k-steps ahead), token fragments from LDS, no barrier and no LDS write inside;
//   * the accumulator layout (lane = token, 4 consecutive channels per register quad) IS an 8-byte half chunk of the tile layout, so
//     LayerNorm1 reads its residual from Q and writes x1 (fp16) back IN PLACE - Q is then the A tile of linear1 and the residual of
//     LayerNorm2 -, and ReLU(linear1) is written into P, dead since the first K loop, as the A tile of linear2;
//   * LayerNorm statistics: two passes in fp32 on the accumulators, lane -> partner lane (xor 32) -> the 8 waves through LDS in a
//     fixed order; the residual stream stays fp32 inside a LayerNorm as in the unfused form (x1 is rounded to fp16 once, as before).
// One workgroup per CU (132 KB of LDS); what the second workgroup per CU bought the unfused kernels - one's loads beside the other's
// K loop - is bought here by having no loads behind the first 128 KB.
#include "common.h"

#define HM_THREADS 512
#define HM_TILE 65536
#define HM_RED_OFF (2 * HM_TILE)                 // [2][8 waves][64 tokens] floats
#define HM_LDS_BYTES (2 * HM_TILE + 4096)

typedef unsigned int hm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void hm_glds16(const f16 *sbase, unsigned voff_bytes, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(lds_addr) : "memory");
}

// sum over the 16 lanes of a DPP row, result in every lane; fixed order (as tok_gemm.hip)
__device__ __forceinline__ float hm_row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
  return x;
}

// 64 rows x 512 fp16 of `src` (rows past M repeat the last one) -> LDS tile; 8 DMA instructions (1 KB each) per wave
__device__ __forceinline__ void hm_tile_dma(const f16 *src, int m0, int M, int wave, int lane, unsigned lds0) {
#pragma unroll
  for (int seg = 0; seg < 4; ++seg)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r4 = wave * 2 + u, row = r4 * 4 + (lane >> 4);
      const int m = min(m0 + row, M - 1);
      const unsigned voff = (unsigned)(((size_t)m * 512 + seg * 128 + (((lane & 15) ^ (row & 15)) * 8)) * 2);
      hm_glds16(src, voff, lds0 + seg * 16384 + r4 * 1024);
    }
}

struct HmCtx {
  int lane, wave, lr, lh;
  unsigned xo[8];
};

// acc[i][j] (channels wave*64 + i*32 + ..., tokens j*32 + lr) = bias + tile (64 x 512, LDS) @ W^T
__device__ __forceinline__ void hm_gemm(const HmCtx &c, const f16 *w, const float *bias, const unsigned char *tile, floatx16 (&acc)[2][2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 bv = *reinterpret_cast<const float4 *>(bias + c.wave * 64 + i * 32 + rg * 8 + c.lh * 4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j][rg * 4 + 0] = bv.x;
        acc[i][j][rg * 4 + 1] = bv.y;
        acc[i][j][rg * 4 + 2] = bv.z;
        acc[i][j][rg * 4 + 3] = bv.w;
      }
    }
  // packed [wave4][k16 32][i4 4][lane 64][8 halfs] (pack_tok_weights): this wave's fragments are (wave >> 1, k16, (wave & 1) * 2 + i)
  const hm_u32x4 *wp = reinterpret_cast<const hm_u32x4 *>(w) + (size_t)(c.wave >> 1) * (32 * 4 * 64) + ((c.wave & 1) * 2) * 64 + c.lane;
  constexpr int D = 4;
  hm_u32x4 wr[D][2];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) wr[d][i] = wp[(d * 4 + i) * 64];
  const unsigned char *xb = tile + c.lr * 256;
  half8 bf[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bf[0][j] = *reinterpret_cast<const half8 *>(xb + j * 8192 + c.xo[0]);
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int cur = k & 1, slot = k % D;
    half8 af[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<half8 *>(&wr[slot][i]);
    __builtin_amdgcn_sched_barrier(0);      // pin the prefetch (hipcc otherwise sinks the loads next to their use)
    if (k + D < 32) {
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[slot][i] = wp[((k + D) * 4 + i) * 64];
    }
    if (k + 1 < 32) {
      const int kn = k + 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[cur ^ 1][j] = *reinterpret_cast<const half8 *>(xb + (kn >> 3) * 16384 + j * 8192 + c.xo[kn & 7]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[cur][j], acc[i][j], 0, 0, 0);
  }
}

// byte offset inside a tile of the 8-byte half chunk that holds channels wave*64 + i*32 + rg*8 + lh*4 .. +3 of token row
__device__ __forceinline__ unsigned hm_off(const HmCtx &c, int row, int i, int rg) {
  return (unsigned)((c.wave >> 1) * 16384 + row * 256 + ((((c.wave & 1) * 8 + i * 4 + rg) ^ (row & 15)) * 16) + c.lh * 8);
}

// acc += residual tile (fp16, fp32 add); then the LayerNorm statistics of every token over the 512 columns -> acc = (acc - mean), rstd
__device__ __forceinline__ void hm_residual_stats(const HmCtx &c, const unsigned char *res_tile, float *red, floatx16 (&acc)[2][2], float (&rstd)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = j * 32 + c.lr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const half4 rq = *reinterpret_cast<const half4 *>(res_tile + hm_off(c, row, i, rg));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][rg * 4 + e] += (float)rq[e];
      }
  }
  float mean[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    s += __shfl_xor(s, 32);
    if (c.lh == 0) red[c.wave * 64 + j * 32 + c.lr] = s;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w * 64 + j * 32 + c.lr];
    mean[j] = s * (1.f / 512.f);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] -= mean[j];
        s += acc[i][j][e] * acc[i][j][e];
      }
    s += __shfl_xor(s, 32);
    if (c.lh == 0) red[512 + c.wave * 64 + j * 32 + c.lr] = s;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[512 + w * 64 + j * 32 + c.lr];
    rstd[j] = rsqrtf(s * (1.f / 512.f) + 1e-5f);
  }
}

__global__ __launch_bounds__(HM_THREADS, 1) void head_mlp_kernel(HeadMlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hm_smem[];
  HmCtx c;
  const int tid = threadIdx.x;
  c.lane = tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.lr = c.lane & 31;
  c.lh = c.lane >> 5;
#pragma unroll
  for (int s = 0; s < 8; ++s) c.xo[s] = (unsigned)(((2 * s + c.lh) ^ (c.lr & 15)) * 16);
  const int m0 = blockIdx.x * 64;
  unsigned char *P = hm_smem, *Q = hm_smem + HM_TILE;
  float *red = reinterpret_cast<float *>(hm_smem + HM_RED_OFF);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)hm_smem;

  hm_tile_dma(p.att, m0, p.M, c.wave, c.lane, lds0);             // 8 instructions: the attention output -> P
  hm_tile_dma(p.tok, m0, p.M, c.wave, c.lane, lds0 + HM_TILE);   // 8 more: the residual tokens -> Q
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // P has landed (this wave's part; the barrier covers the others')
  __syncthreads();

  floatx16 acc[2][2];
  float rstd[2];
  // ---- out-projection + residual + LayerNorm1 -> x1 (fp16) in place into Q ----
  hm_gemm(c, p.w_out, p.b_out, P, acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // Q has landed
  __syncthreads();                                               // (and every wave is done with P)
  hm_residual_stats(c, Q, red, acc, rstd);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int col = c.wave * 64 + i * 32 + rg * 8 + c.lh * 4;
      const float4 gv = *reinterpret_cast<const float4 *>(p.g1 + col), bv = *reinterpret_cast<const float4 *>(p.be1 + col);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half4 hv;
        hv[0] = (f16)(acc[i][j][rg * 4 + 0] * rstd[j] * gv.x + bv.x);
        hv[1] = (f16)(acc[i][j][rg * 4 + 1] * rstd[j] * gv.y + bv.y);
        hv[2] = (f16)(acc[i][j][rg * 4 + 2] * rstd[j] * gv.z + bv.z);
        hv[3] = (f16)(acc[i][j][rg * 4 + 3] * rstd[j] * gv.w + bv.w);
        *reinterpret_cast<half4 *>(Q + hm_off(c, j * 32 + c.lr, i, rg)) = hv;     // the residual chunk this lane read: same address
      }
    }
  __syncthreads();
  // ---- linear1 + ReLU -> ff (fp16) into P ----
  hm_gemm(c, p.w1, p.b1, Q, acc);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][rg * 4 + e];
        hv = __builtin_elementwise_max(hv, half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
        *reinterpret_cast<half4 *>(P + hm_off(c, j * 32 + c.lr, i, rg)) = hv;
      }
  __syncthreads();
  // ---- linear2 + residual x1 + LayerNorm2 statistics -> sums of the normalised rows over groups of 16 tokens ----
  hm_gemm(c, p.w2, p.b2, P, acc);
  hm_residual_stats(c, Q, red, acc, rstd);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = hm_row16_sum(acc[i][j][rg * 4 + e] * rstd[j]);
        const int g = (m0 + j * 32 + (c.lr & 16)) >> 4;          // global 16-token group (16 divides 400: never two hypotheses)
        if ((c.lr & 15) == 0 && g * 16 < p.M)
          *reinterpret_cast<float4 *>(p.gsum + (size_t)g * 512 + c.wave * 64 + i * 32 + rg * 8 + c.lh * 4) = make_float4(v[0], v[1], v[2], v[3]);
      }
}

void head_mlp_kernel_lds(std::vector<KernelLds> &v) { v.push_back({(const void *)head_mlp_kernel, HM_LDS_BYTES}); }

int launch_head_mlp(fp_ctx *ctx, const HeadMlpArgs &a, hipStream_t s) {
  FP_REQUIRE(a.att && a.tok && a.w_out && a.w1 && a.w2 && a.b_out && a.b1 && a.b2 && a.g1 && a.be1 && a.gsum, "head_mlp: null argument");
  FP_REQUIRE(a.M >= 0 && a.M % 16 == 0, "head_mlp: M=%d must be a multiple of 16", a.M);
  if (a.M == 0) return FP_OK;
  FP_REQUIRE((double)a.M * 1024.0 < 4294967296.0, "head_mlp: M=%d too large for 32-bit lane offsets", a.M);
  ProfScope ps(ctx, s, "linear", 3.0 * 2.0 * (double)a.M * 512.0 * 512.0);
  hipLaunchKernelGGL(head_mlp_kernel, dim3((a.M + 63) / 64), dim3(HM_THREADS), HM_LDS_BYTES, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

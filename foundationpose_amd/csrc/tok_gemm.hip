// Token GEMM for the transformer heads (gfx950): every nn.Linear that acts on the (N*400, 512) token tensor -
// the q/k/v in-projections, the attention out-projection, linear1 and linear2 of nn.TransformerEncoderLayer
// (learning/models/refine_network.py:56-70, score_network.py:53-54) - with the residual add, the LayerNorm
// and the token mean that follow them fused into the epilogue.
//
//   out[m][c] = sum_k x[m][k] * W[c][k] + b[c]            K = 512, 512 output columns per workgroup
//
// A 1x1 layer has no spatial reuse: in the implicit-GEMM kernel of conv.hip both operands stream through LDS-DMA and the
// L2 -> LDS fill (25 B/clk/CU measured) bounds the K loop at half the MFMA rate, with a workgroup barrier per K-step.
// Here
//   * the ACTIVATION tile (128 tokens x 512 = 128 KB) is brought into LDS ONCE (LDS-DMA, XOR-swizzled 256-byte row
//     segments: conflict-free ds_read_b128) and stays resident for the whole K loop;
//   * the WEIGHTS never touch LDS: each of the 8 waves owns 64 output columns and loads its MFMA fragments straight
//     from global memory / L2 into registers (fragment-ordered packing done at load time: one coalesced 1-KB
//     global_load_dwordx4 per fragment), prefetched TG_D k-steps ahead;
//   * so the K loop has NO barrier and no LDS write: per k-step of 16 a wave issues 2 global loads, 4 ds_read_b128 and
//     8 v_mfma_f32_32x32x16_f16 (64 columns x 128 tokens = 2 x 4 accumulator tiles, 128 VGPRs).
// Epilogues (compile-time):
//   EPI_ROWS   bias (+ReLU) -> fp16 rows (q|k projection, linear1)
//   EPI_VT     bias -> transposed V image [b][4][128][416] in the attention kernel's token order (vt_col); the MFMA
//              operands are swapped for this one so that a lane owns one channel and 4 consecutive tokens per register quad
//   EPI_LN     bias + residual (fp32) -> LayerNorm over the 512 columns (two-pass statistics in fp32 on the accumulators,
//              cross-wave through LDS) -> fp16 rows (out-projection + norm1)
//   EPI_LNSUM  the same LayerNorm, but instead of the rows only their sums over groups of 16 tokens are written (fp32):
//              norm2 feeds nothing but the token mean (refine_network.py:90-91); 16 divides 400, so a group never spans two
//              hypotheses and the reduction order of a hypothesis does not depend on where it sits in the batch.
// The residual stream and the LayerNorm inputs stay fp32 (the reference's autocast keeps them fp32 as well: `x + pe`
// promotes, layer_norm runs in fp32); only GEMM operands are fp16.
#include "common.h"

#define TG_ROWS 128
#define TG_K 512
#define TG_THREADS 512
#define TG_D 4                                  // weight prefetch distance in k-steps of 16
#define TG_ACT_BYTES (TG_ROWS * TG_K * 2)       // 131072: resident activation tile
#define TG_STAGE_LD 72                          // halfs per staged row of 64 columns (128 B + 16 B pad: conflict-free 8-byte writes)
#define TG_LDS_BYTES 163840                     // 160 KB: tile (128 KB) + reduction scratch; the epilogue staging re-uses the tile

enum { EPI_ROWS = TG_EPI_ROWS, EPI_VT = TG_EPI_VT, EPI_LN = TG_EPI_LN, EPI_LNSUM = TG_EPI_LNSUM };

typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tg_glds16(const f16 *sbase, unsigned voff_bytes, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(TG_THREADS, 1) void tok_gemm_kernel(TokGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tg_smem[];
  const int L = xcd_remap(blockIdx.x, gridDim.x);          // workgroups that share an XCD's L2 walk the column blocks of one row tile
  const int rt = L / p.nblk, cb = L - rt * p.nblk;
  const TokGemmBlock &blk = p.blk[cb];
  const int m0 = rt * TG_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  // ---- activation tile -> LDS: [k segment of 128][row][256 B], 16-byte chunk c of a row segment stored at c ^ (row & 15).
  // One DMA instruction = 4 rows of one segment (1 KB, lane-linear destination; the swizzle is applied on the source address);
  // rows past M repeat the last row (their outputs are never stored).
  {
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)tg_smem;
#pragma unroll
    for (int seg = 0; seg < 4; ++seg)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r4 = wave * 4 + u, row = r4 * 4 + (lane >> 4);
        const int m = min(m0 + row, p.M - 1);
        const unsigned voff = (unsigned)(((size_t)m * TG_K + seg * 128 + (((lane & 15) ^ (row & 15)) * 8)) * 2);
        tg_glds16(p.in, voff, lds0 + seg * 32768 + r4 * 1024);
      }
  }

  // ---- weight fragments: packed [wave][k16][i][lane][8 halfs]; accumulators start at the bias ----
  const tg_u32x4 *wp = reinterpret_cast<const tg_u32x4 *>(blk.w) + (size_t)wave * (32 * 2 * 64) + lane;
  floatx16 acc[2][4];
  if constexpr (EPI == EPI_VT) {          // swapped operands: a lane owns ONE channel (i*32 + lr) of the wave's 64
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float b = blk.bias[wave * 64 + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = b;
    }
  } else {                                // a lane owns one token and channels i*32 + rg*8 + lh*4 + (0..3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 bv = *reinterpret_cast<const float4 *>(blk.bias + wave * 64 + i * 32 + rg * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j][rg * 4 + 0] = bv.x;
          acc[i][j][rg * 4 + 1] = bv.y;
          acc[i][j][rg * 4 + 2] = bv.z;
          acc[i][j][rg * 4 + 3] = bv.w;
        }
      }
  }
  tg_u32x4 wr[TG_D][2];
#pragma unroll
  for (int d = 0; d < TG_D; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) wr[d][i] = wp[(d * 2 + i) * 64];

  // ---- token fragments: row j*32 + lr, chunk (2*(k16&7) + lh) ^ (lr & 15) of segment k16 >> 3 ----
  const unsigned char *xb = tg_smem + lr * 256;
  unsigned xo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) xo[s] = (unsigned)(((2 * s + lh) ^ (lr & 15)) * 16);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile (and the first weight fragments) landed
  __syncthreads();

  half8 bf[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[0][j] = *reinterpret_cast<const half8 *>(xb + j * 8192 + xo[0]);
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int cur = k & 1, slot = k % TG_D;
    half8 af[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<half8 *>(&wr[slot][i]);
    __builtin_amdgcn_sched_barrier(0);      // pin the prefetch: hipcc otherwise sinks the loads next to their use (distance 1)
    if (k + TG_D < 32) {
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[slot][i] = wp[((k + TG_D) * 2 + i) * 64];
    }
    if (k + 1 < 32) {
      const int kn = k + 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[cur ^ 1][j] = *reinterpret_cast<const half8 *>(xb + (kn >> 3) * 32768 + j * 8192 + xo[kn & 7]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (EPI == EPI_VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur][j], af[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[cur][j], acc[i][j], 0, 0, 0);
      }
  }

  __syncthreads();                                       // every wave is done with the tile: the epilogues re-use its LDS
  f16 *stage = reinterpret_cast<f16 *>(tg_smem) + (size_t)wave * (TG_ROWS * TG_STAGE_LD);   // 18 KB per wave
  float *red = reinterpret_cast<float *>(tg_smem + 8 * TG_ROWS * TG_STAGE_LD * 2);          // [2][8 waves][128] behind the staging

  if constexpr (EPI == EPI_VT) {
    // acc[i][j][r]: channel c = wave*64 + i*32 + lr, token j*32 + (r&3) + 8*(r>>2) + 4*lh.  Stage as [channel][token in vt
    // order] (row = 128 tokens = 256 B + pad), then 16-byte stores along the token axis of the image.
    constexpr int VLD = 136;                            // halfs per staged channel row (128 + 8 pad)
    f16 *vs = reinterpret_cast<f16 *>(tg_smem) + (size_t)wave * (64 * VLD);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][q * 4 + e];
          const int t = j * 32 + 8 * q + 4 * lh;        // first of 4 consecutive tokens (tile-relative); the tile starts at a multiple of 16
          *reinterpret_cast<half4 *>(&vs[(i * 32 + lr) * VLD + vt_col(t)]) = hv;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // read back: instruction u covers 4 channel rows x 16 chunks of 8 tokens
    f16 *img = (f16 *)blk.out;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = u * 4 + (lane >> 4), ch16 = lane & 15;
      const uint4 v = *reinterpret_cast<const uint4 *>(&vs[c * VLD + ch16 * 8]);
      const int m = m0 + ch16 * 8;                      // the 8 tokens of a chunk lie in one group of 16: one hypothesis
      if (m < p.M) {
        const int b = m / p.tokens, t = m - b * p.tokens;            // t is a multiple of 8: vt_col keeps chunks of 8 together
        const int col = blk.coff + wave * 64 + c, h = col >> 7, d = col & 127;
        *reinterpret_cast<uint4 *>(img + (((size_t)b * 4 + h) * 128 + d) * 416 + (t & ~15) + (ch16 & 1) * 8) = v;
      }
    }
    return;
  }

  if constexpr (EPI == EPI_LN || EPI == EPI_LNSUM) {
    // residual in fp32, straight into the accumulators (8-byte loads in the accumulator layout)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = min(m0 + j * 32 + lr, p.M - 1);
      half4 rq[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rq[i][rg] = *reinterpret_cast<const half4 *>(p.res + (size_t)m * 512 + wave * 64 + i * 32 + rg * 8 + lh * 4);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][rg * 4 + e] += (float)rq[i][rg][e];
    }
    // LayerNorm statistics per token over the 512 columns: lane -> 32 of the wave's 64 columns, partner lane (xor 32) the
    // other 32, then the 8 waves through LDS in a fixed order.  Two passes (mean, then centred second moment).
    float mean[4], rstd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
      s += __shfl_xor(s, 32);
      if (lh == 0) red[wave * 128 + j * 32 + lr] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[w * 128 + j * 32 + lr];
      mean[j] = s * (1.f / 512.f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[i][j][e] -= mean[j];
          s += acc[i][j][e] * acc[i][j][e];
        }
      s += __shfl_xor(s, 32);
      if (lh == 0) red[1024 + wave * 128 + j * 32 + lr] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[1024 + w * 128 + j * 32 + lr];
      rstd[j] = rsqrtf(s * (1.f / 512.f) + 1e-5f);
    }
    if constexpr (EPI == EPI_LNSUM) {
      // sums of the normalised values over groups of 16 tokens (lanes lr 0-15 / 16-31 of a token tile): xor-butterfly,
      // fixed order; gamma / beta are applied after the token mean (mean_head_kernel)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[i][j][rg * 4 + e] * rstd[j];
              x += __shfl_xor(x, 1);
              x += __shfl_xor(x, 2);
              x += __shfl_xor(x, 4);
              x += __shfl_xor(x, 8);
              v[e] = x;
            }
            const int g = (m0 + j * 32 + (lr & 16)) >> 4;          // global 16-token group
            if ((lr & 15) == 0 && g * 16 < p.M)
              *reinterpret_cast<float4 *>(p.gsum + (size_t)g * 512 + wave * 64 + i * 32 + rg * 8 + lh * 4) = make_float4(v[0], v[1], v[2], v[3]);
          }
      return;
    }
    // gamma / beta, then the fp16 row epilogue below
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int c = wave * 64 + i * 32 + rg * 8 + lh * 4;
        const float4 gv = *reinterpret_cast<const float4 *>(p.gamma + c), bv = *reinterpret_cast<const float4 *>(p.beta + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j][rg * 4 + 0] = acc[i][j][rg * 4 + 0] * rstd[j] * gv.x + bv.x;
          acc[i][j][rg * 4 + 1] = acc[i][j][rg * 4 + 1] * rstd[j] * gv.y + bv.y;
          acc[i][j][rg * 4 + 2] = acc[i][j][rg * 4 + 2] * rstd[j] * gv.z + bv.z;
          acc[i][j][rg * 4 + 3] = acc[i][j][rg * 4 + 3] * rstd[j] * gv.w + bv.w;
        }
      }
  }

  if constexpr (EPI != EPI_VT && EPI != EPI_LNSUM) {
    // fp16 rows: the wave's 128 tokens x 64 columns through its private staging (no workgroup barrier), 16-byte stores of
    // whole 128-byte row segments
    const bool relu = blk.relu != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][rg * 4 + e];
          if (relu) hv = __builtin_elementwise_max(hv, half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
          *reinterpret_cast<half4 *>(&stage[(j * 32 + lr) * TG_STAGE_LD + i * 32 + rg * 8 + lh * 4]) = hv;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    f16 *obase = (f16 *)blk.out + blk.coff + wave * 64;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int px = u * 8 + (lane >> 3), c16 = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4 *>(&stage[px * TG_STAGE_LD + c16 * 8]);
      if (m0 + px < p.M) *reinterpret_cast<uint4 *>(obase + (size_t)(m0 + px) * blk.ld + c16 * 8) = v;
    }
  }
}

template <int EPI>
static int tg_launch(const TokGemmArgs &a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    FP_CHECK_HIP(hipFuncSetAttribute((const void *)tok_gemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS_BYTES));
    attr_set = true;
  }
  const int n_rt = (a.M + TG_ROWS - 1) / TG_ROWS;
  hipLaunchKernelGGL((tok_gemm_kernel<EPI>), dim3(n_rt * a.nblk), dim3(TG_THREADS), TG_LDS_BYTES, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int launch_tok_gemm(fp_ctx *ctx, const TokGemmArgs &a, int epi, hipStream_t s) {
  FP_REQUIRE(a.in && a.M >= 0 && a.nblk >= 1 && a.nblk <= TG_MAXBLK, "tok_gemm: bad arguments");
  if (a.M == 0) return FP_OK;
  FP_REQUIRE((double)a.M * TG_K * 2.0 < 4294967296.0, "tok_gemm: M=%d too large for 32-bit lane offsets", a.M);
  for (int b = 0; b < a.nblk; ++b) FP_REQUIRE(a.blk[b].w && a.blk[b].bias, "tok_gemm: block %d has null weights", b);
  ProfScope ps(ctx, s, "linear", 2.0 * (double)a.M * 512.0 * 512.0 * a.nblk);
  switch (epi) {
    case EPI_ROWS:
      for (int b = 0; b < a.nblk; ++b) FP_REQUIRE(a.blk[b].out && a.blk[b].ld % 8 == 0 && a.blk[b].coff % 64 == 0, "tok_gemm: bad output of block %d", b);
      return tg_launch<EPI_ROWS>(a, s);
    case EPI_VT:
      FP_REQUIRE(a.tokens > 0 && a.tokens % 16 == 0 && a.tokens <= 416, "tok_gemm: tokens=%d must be a multiple of 16, <= 416", a.tokens);
      for (int b = 0; b < a.nblk; ++b) FP_REQUIRE(a.blk[b].out, "tok_gemm: block %d has no output", b);
      return tg_launch<EPI_VT>(a, s);
    case EPI_LN:
      FP_REQUIRE(a.nblk == 1 && a.res && a.gamma && a.beta && a.blk[0].out && a.blk[0].ld % 8 == 0, "tok_gemm: LN epilogue needs one block, residual, gamma, beta");
      return tg_launch<EPI_LN>(a, s);
    case EPI_LNSUM:
      FP_REQUIRE(a.nblk == 1 && a.res && a.gsum && a.M % 16 == 0, "tok_gemm: LN-sum epilogue needs one block, residual, gsum, M %% 16 == 0");
      return tg_launch<EPI_LNSUM>(a, s);
  }
  FP_REQUIRE(false, "tok_gemm: unknown epilogue %d", epi);
}

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
//   * the ACTIVATION tile (64 tokens x 512 = 64 KB) is brought into LDS ONCE (LDS-DMA, XOR-swizzled 256-byte row
//     segments: conflict-free ds_read_b128) and stays resident for the whole K loop;
//   * the WEIGHTS never touch LDS: each of the 4 waves owns 128 output columns and loads its MFMA fragments straight
//     from global memory / L2 into registers (fragment-ordered packing done at load time: one coalesced 1-KB
//     global_load_dwordx4 per fragment), prefetched TG_D k-steps ahead;
//   * so the K loop has NO barrier and no LDS write: per k-step of 16 a wave issues 4 global loads, 2 ds_read_b128 and
//     8 v_mfma_f32_32x32x16_f16 (128 columns x 64 tokens = 4 x 2 accumulator tiles, 128 VGPRs);
//   * TWO workgroups share a CU (76 KB of LDS, <= 256 VGPRs each): the three phases of a workgroup's life - tile load
//     (HBM read), K loop (MFMA) and epilogue (LDS transpose + HBM write) - take about the same time (stamps of the first form,
//     one 8-wave workgroup per CU with a 128-token tile: 15k / 15k / 10-48k cycles) and use different units, so one
//     workgroup's K loop runs beside the other's load or epilogue instead of after it.
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

#define TG_ROWS 64
#define TG_K 512
#define TG_THREADS 256
#define TG_SEG_BYTES (TG_ROWS * 256)            // one k segment of 128 of the resident tile: 16 KB
#define TG_STAGE_LD 136                         // halfs per staged row of 128 columns (256 B + 16 B pad: conflict-free 8-byte writes)
#define TG_VT_LD 72                             // halfs per staged channel row of 64 tokens (128 B + 16 B pad)
#define TG_LDS_BYTES 77824                      // 76 KB: two workgroups per CU.  Tile 64 KB; the epilogue staging (68 - 72 KB) re-uses it
#define TG_RED_OFF 73728                        // reduction scratch [2][4 waves][64] floats behind the staging

enum { EPI_ROWS = TG_EPI_ROWS, EPI_VT = TG_EPI_VT, EPI_LN = TG_EPI_LN, EPI_LNSUM = TG_EPI_LNSUM };

typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));

#ifdef HALO_STAMP
// diagnostic build only (make -B EXTRA=-DHALO_STAMP): s_memtime per wave at entry / tile resident / K loop done / exit
__device__ unsigned long long g_tok_stamps[2048 * 4 * 4];
extern "C" __attribute__((visibility("default"))) int fp_dbg_tok_stamps(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tok_stamps), sizeof(g_tok_stamps)) == hipSuccess ? 0 : -1;
}
#define TSTAMP(i) do { if (blockIdx.x < 2048 && lane == 0) g_tok_stamps[((size_t)blockIdx.x * 4 + wave) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void tg_glds16(const f16 *sbase, unsigned voff_bytes, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(lds_addr) : "memory");
}

// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane; fixed order.  quad_perm [1,0,3,2] and
// [2,3,0,1] add within quads, row_half_mirror / row_mirror exchange quads whose four lanes already hold equal sums.
__device__ __forceinline__ float tg_row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
  return x;
}

// 64 rows x 512 fp16 of `src` (rows past M repeat the last one) -> LDS [k segment of 128][row][256 B], 16-byte chunk c of
// a row segment stored at c ^ (row & 15).  One DMA instruction = 4 rows of one segment (1 KB, lane-linear destination; the
// swizzle is applied on the source address).  16 instructions per wave.
__device__ __forceinline__ void tg_tile_dma(const f16 *src, int m0, int M, int wave, int lane, unsigned lds0) {
#pragma unroll
  for (int seg = 0; seg < 4; ++seg)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r4 = wave * 4 + u, row = r4 * 4 + (lane >> 4);
      const int m = min(m0 + row, M - 1);
      const unsigned voff = (unsigned)(((size_t)m * TG_K + seg * 128 + (((lane & 15) ^ (row & 15)) * 8)) * 2);
      tg_glds16(src, voff, lds0 + seg * TG_SEG_BYTES + r4 * 1024);
    }
}

template <int EPI>
__global__ __launch_bounds__(TG_THREADS, 2) void tok_gemm_kernel(TokGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tg_smem[];
  const int L = xcd_remap(blockIdx.x, gridDim.x);          // workgroups that share an XCD's L2 walk the column blocks of one row tile
  const int rt = L / p.nblk, cb = L - rt * p.nblk;
  const TokGemmBlock &blk = p.blk[cb];
  const int m0 = rt * TG_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)tg_smem;
  TSTAMP(0);
  tg_tile_dma(p.in, m0, p.M, wave, lane, lds0);

  // ---- weight fragments: packed [wave][k16][i][lane][8 halfs]; accumulators start at the bias ----
  const tg_u32x4 *wp = reinterpret_cast<const tg_u32x4 *>(blk.w) + (size_t)wave * (32 * 4 * 64) + lane;
  floatx16 acc[4][2];
  if constexpr (EPI == EPI_VT) {          // swapped operands: a lane owns ONE channel (i*32 + lr) of the wave's 128
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float b = blk.bias[wave * 128 + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = b;
    }
  } else {                                // a lane owns one token and channels i*32 + rg*8 + lh*4 + (0..3)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 bv = *reinterpret_cast<const float4 *>(blk.bias + wave * 128 + i * 32 + rg * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][rg * 4 + 0] = bv.x;
          acc[i][j][rg * 4 + 1] = bv.y;
          acc[i][j][rg * 4 + 2] = bv.z;
          acc[i][j][rg * 4 + 3] = bv.w;
        }
      }
  }
  constexpr int TG_D = (EPI == EPI_LN || EPI == EPI_LNSUM) ? 3 : 4;   // weight prefetch distance in k-steps of 16 (3 where the
                                                                        // LayerNorm epilogue needs the registers: no spills)
  tg_u32x4 wr[TG_D][4];
#pragma unroll
  for (int d = 0; d < TG_D; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[d][i] = wp[(d * 4 + i) * 64];

  // ---- token fragments: row j*32 + lr, chunk (2*(k16&7) + lh) ^ (lr & 15) of segment k16 >> 3 ----
  const unsigned char *xb = tg_smem + lr * 256;
  unsigned xo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) xo[s] = (unsigned)(((2 * s + lh) ^ (lr & 15)) * 16);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile (and the first weight fragments) landed
  __syncthreads();
  TSTAMP(1);

  half8 bf[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bf[0][j] = *reinterpret_cast<const half8 *>(xb + j * 8192 + xo[0]);
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int cur = k & 1, slot = k % TG_D;
    half8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<half8 *>(&wr[slot][i]);
    __builtin_amdgcn_sched_barrier(0);      // pin the prefetch: hipcc otherwise sinks the loads next to their use (distance 1)
    if (k + TG_D < 32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wr[slot][i] = wp[((k + TG_D) * 4 + i) * 64];
    }
    if (k + 1 < 32) {
      const int kn = k + 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[cur ^ 1][j] = *reinterpret_cast<const half8 *>(xb + (kn >> 3) * TG_SEG_BYTES + j * 8192 + xo[kn & 7]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (EPI == EPI_VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur][j], af[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[cur][j], acc[i][j], 0, 0, 0);
      }
  }

  TSTAMP(2);
  __syncthreads();                                       // every wave is done with the tile: the epilogues re-use its LDS
  float *red = reinterpret_cast<float *>(tg_smem + TG_RED_OFF);

  if constexpr (EPI == EPI_VT) {
    // acc[i][j][r]: channel c = wave*128 + i*32 + lr, token j*32 + (r&3) + 8*(r>>2) + 4*lh.  Stage as [channel][token in vt
    // order] (row = 64 tokens = 128 B + pad), then 16-byte stores along the token axis of the image.
    f16 *vs = reinterpret_cast<f16 *>(tg_smem) + (size_t)wave * (128 * TG_VT_LD);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][q * 4 + e];
          const int t = j * 32 + 8 * q + 4 * lh;        // first of 4 consecutive tokens (tile-relative); the tile starts at a multiple of 16
          *reinterpret_cast<half4 *>(&vs[(i * 32 + lr) * TG_VT_LD + vt_col(t)]) = hv;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // read back: instruction u covers 8 channel rows x 8 chunks of 8 tokens
    f16 *img = (f16 *)blk.out;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = u * 8 + (lane >> 3), ch8 = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4 *>(&vs[c * TG_VT_LD + ch8 * 8]);
      const int m = m0 + ch8 * 8;                       // the 8 tokens of a chunk lie in one group of 16: one hypothesis
      if (m < p.M) {
        const int b = m / p.tokens, t = m - b * p.tokens;
        const int col = blk.coff + wave * 128 + c, h = col >> 7, d = col & 127;
        f16 *row = img + (((size_t)b * 4 + h) * 128 + d) * 416;
        *reinterpret_cast<uint4 *>(row + (t & ~15) + (ch8 & 1) * 8) = v;
        // the image is 416 tokens wide: the lanes that store a hypothesis' last 16 tokens also zero the columns behind them
        // (the attention kernel reads whole 64-key blocks; was a launch of its own)
        if ((t & ~15) + 16 == p.tokens)
          for (int z = p.tokens; z < 416; z += 16) *reinterpret_cast<uint4 *>(row + z + (ch8 & 1) * 8) = uint4{0u, 0u, 0u, 0u};
      }
    }
    TSTAMP(3);
    return;
  }

  if constexpr (EPI == EPI_LN || EPI == EPI_LNSUM) {
    // residual tile -> the (now free) tile region by LDS-DMA, then added in fp32 in the accumulator layout: global column
    // wave*128 + i*32 + rg*8 + lh*4 is chunk i*4 + rg of k segment `wave`, 8-byte half lh
    tg_tile_dma(p.res, m0, p.M, wave, lane, lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 32 + lr;
      const unsigned char *rb = tg_smem + wave * TG_SEG_BYTES + row * 256 + lh * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const half4 rq = *reinterpret_cast<const half4 *>(rb + (((i * 4 + rg) ^ (row & 15)) * 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][rg * 4 + e] += (float)rq[e];
        }
    }
    // LayerNorm statistics per token over the 512 columns: lane -> 64 of the wave's 128 columns, partner lane (xor 32) the
    // other 64, then the 4 waves through LDS in a fixed order.  Two passes (mean, then centred second moment).
    float mean[2], rstd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
      s += __shfl_xor(s, 32);
      if (lh == 0) red[wave * 64 + j * 32 + lr] = s;
    }
    __syncthreads();                                     // (also: every wave has consumed its residual chunks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[w * 64 + j * 32 + lr];
      mean[j] = s * (1.f / 512.f);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[i][j][e] -= mean[j];
          s += acc[i][j][e] * acc[i][j][e];
        }
      s += __shfl_xor(s, 32);
      if (lh == 0) red[256 + wave * 64 + j * 32 + lr] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[256 + w * 64 + j * 32 + lr];
      rstd[j] = rsqrtf(s * (1.f / 512.f) + 1e-5f);
    }
    if constexpr (EPI == EPI_LNSUM) {
      // sums of the normalised values over groups of 16 tokens (lanes lr 0-15 / 16-31 of a token tile = one DPP row each);
      // gamma / beta are applied after the token mean (mean_head_kernel)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tg_row16_sum(acc[i][j][rg * 4 + e] * rstd[j]);
            const int g = (m0 + j * 32 + (lr & 16)) >> 4;          // global 16-token group
            if ((lr & 15) == 0 && g * 16 < p.M)
              *reinterpret_cast<float4 *>(p.gsum + (size_t)g * 512 + wave * 128 + i * 32 + rg * 8 + lh * 4) = make_float4(v[0], v[1], v[2], v[3]);
          }
      TSTAMP(3);
      return;
    }
    // gamma / beta, then the fp16 row epilogue below
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int c = wave * 128 + i * 32 + rg * 8 + lh * 4;
        const float4 gv = *reinterpret_cast<const float4 *>(p.gamma + c), bv = *reinterpret_cast<const float4 *>(p.beta + c);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][rg * 4 + 0] = acc[i][j][rg * 4 + 0] * rstd[j] * gv.x + bv.x;
          acc[i][j][rg * 4 + 1] = acc[i][j][rg * 4 + 1] * rstd[j] * gv.y + bv.y;
          acc[i][j][rg * 4 + 2] = acc[i][j][rg * 4 + 2] * rstd[j] * gv.z + bv.z;
          acc[i][j][rg * 4 + 3] = acc[i][j][rg * 4 + 3] * rstd[j] * gv.w + bv.w;
        }
      }
  }

  if constexpr (EPI != EPI_VT && EPI != EPI_LNSUM) {
    // fp16 rows: the wave's 64 tokens x 128 columns through its private staging (no workgroup barrier), 16-byte stores of
    // whole 256-byte row segments.  (EPI_LN: the barrier after the first statistics pass ordered the last residual reads of
    // every wave before these writes.)
    f16 *stage = reinterpret_cast<f16 *>(tg_smem) + (size_t)wave * (TG_ROWS * TG_STAGE_LD);   // 17 KB per wave
    const bool relu = blk.relu != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][rg * 4 + e];
          if (relu) hv = __builtin_elementwise_max(hv, half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
          *reinterpret_cast<half4 *>(&stage[(j * 32 + lr) * TG_STAGE_LD + i * 32 + rg * 8 + lh * 4]) = hv;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    f16 *obase = (f16 *)blk.out + blk.coff + wave * 128;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int px = u * 4 + (lane >> 4), c16 = lane & 15;
      const uint4 v = *reinterpret_cast<const uint4 *>(&stage[px * TG_STAGE_LD + c16 * 8]);
      if (m0 + px < p.M) *reinterpret_cast<uint4 *>(obase + (size_t)(m0 + px) * blk.ld + c16 * 8) = v;
    }
    TSTAMP(3);
  }
}

template <int EPI>
static int tg_launch(const TokGemmArgs &a, hipStream_t s) {
  const int n_rt = (a.M + TG_ROWS - 1) / TG_ROWS;
  hipLaunchKernelGGL((tok_gemm_kernel<EPI>), dim3(n_rt * a.nblk), dim3(TG_THREADS), TG_LDS_BYTES, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

void tok_gemm_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)tok_gemm_kernel<EPI_ROWS>, TG_LDS_BYTES});
  v.push_back({(const void *)tok_gemm_kernel<EPI_VT>, TG_LDS_BYTES});
  v.push_back({(const void *)tok_gemm_kernel<EPI_LN>, TG_LDS_BYTES});
  v.push_back({(const void *)tok_gemm_kernel<EPI_LNSUM>, TG_LDS_BYTES});
}

int launch_tok_gemm(fp_ctx *ctx, const TokGemmArgs &a, int epi, hipStream_t s) {
  FP_REQUIRE(a.in && a.M >= 0 && a.nblk >= 1 && a.nblk <= TG_MAXBLK, "tok_gemm: bad arguments");
  if (a.M == 0) return FP_OK;
  FP_REQUIRE((double)a.M * TG_K * 2.0 < 4294967296.0, "tok_gemm: M=%d too large for 32-bit lane offsets", a.M);
  for (int b = 0; b < a.nblk; ++b) FP_REQUIRE(a.blk[b].w && a.blk[b].bias, "tok_gemm: block %d has null weights", b);
  ProfScope ps(ctx, s, "linear", 2.0 * (double)a.M * 512.0 * 512.0 * a.nblk);
  switch (epi) {
    case EPI_ROWS:
      for (int b = 0; b < a.nblk; ++b) FP_REQUIRE(a.blk[b].out && a.blk[b].ld % 8 == 0 && a.blk[b].coff % 128 == 0, "tok_gemm: bad output of block %d", b);
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

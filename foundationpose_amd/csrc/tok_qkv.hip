// In-projections of the transformer heads (gfx950): q, k and v of nn.MultiheadAttention / nn.TransformerEncoderLayer
// (learning/models/refine_network.py:56-70, score_network.py:53-54; packed in_proj_weight (1536, 512) = [Wq; Wk; Wv])
// for up to TG_MAXBLK blocks of 512 output columns that read the SAME tokens - RefineNet: q, k, v of both heads, six blocks.
//
// What bounds a token GEMM on this chip is the weight stream L2 -> registers: tok_gemm.hip's 64-token tile re-streams the 512 KB of a
// 512-column block per 64 tokens, 64 B/clk/CU at the MFMA rate where the path delivers ~30-35 (MI355X_MICROARCH.md: 66-73 GB/s per CU
// from the XCD's L2), and it loads the token tile once per block.  Here
//   * a workgroup (8 waves, one per CU) keeps a tile of 128 tokens x 512 (128 KB) resident in LDS - tok_gemm.hip's image: k segments
//     of 128, 256-byte row segments, 16-byte chunk c of a row at c ^ (row & 15): conflict-free ds_read_b128 - and walks the column
//     blocks over it: the tile is loaded once for all blocks, and a weight fragment (one coalesced 1-KB load per wave, fragment order
//     of pack_tok_weights) now feeds FOUR MFMAs: 32 B/clk/CU at the MFMA rate;
//   * a wave owns 64 output columns x 128 tokens (2 x 4 accumulator tiles, v_mfma_f32_32x32x16_f16); the K loop has no barrier and no
//     LDS write: per k-step of 16 two weight fragments (prefetched 3 steps ahead), four token fragments, eight MFMAs;
//   * work = (tile, block) units in tile-major order, cut into equal contiguous ranges over the workgroups (at most one per CU): a
//     launch is balanced to one unit whatever M is - 252 hypotheses are 788 tiles x 6 blocks = 18.5 units per CU, 32 hypotheses 100
//     x 6 = 2.3 - and a workgroup reloads the tile only when its range crosses into the next one;
//   * epilogues, per wave and 32-token slice through a private 4-KB staging area (the two waves of a SIMD drift apart: one's epilogue
//     runs beside the other's K loop): fp16 rows as whole 128-byte segments (q | k), or the transposed V image [b][4][128][416] in the
//     attention kernel's token order (vt_col; MFMA operands swapped so that a lane owns a channel and 4 consecutive tokens).
// Arithmetic per output element is that of tok_gemm.hip (bias first, k ascending in steps of 16, one rounding to fp16): bit-identical
// (tests/test_gpu_kernels.py::test_token_qkv_equals_the_64_token_kernel).
#include "common.h"

#define TQ_ROWS 128
#define TQ_THREADS 512
#define TQ_SEG_BYTES (TQ_ROWS * 256)              // one k segment of 128 of the resident tile: 32 KB
#define TQ_TILE_BYTES (4 * TQ_SEG_BYTES)          // 128 KB
#define TQ_STAGE_BYTES 4096                       // per wave: 32 tokens x 64 columns (rows) / 64 channels x 32 tokens (V image)
#define TQ_LDS_BYTES (TQ_TILE_BYTES + 8 * TQ_STAGE_BYTES)      // 160 KB

typedef unsigned int tq_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tq_glds16(const f16 *sbase, unsigned voff_bytes, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(lds_addr) : "memory");
}

// 128 rows x 512 fp16 of `src` (rows past M repeat the last one) -> LDS [k segment of 128][row][256 B], chunk c of a row segment at
// c ^ (row & 15).  One DMA instruction = 4 rows of one segment (1 KB, lane-linear destination; the swizzle is applied on the source
// address).  16 instructions per wave.
__device__ __forceinline__ void tq_tile_dma(const f16 *src, int m0, int M, int wave, int lane, unsigned lds0) {
  asm volatile("" : "+v"(lane));          // the 16 lane offsets are recomputed per tile: hoisted out of the unit loop they would live across the K loops
#pragma unroll
  for (int seg = 0; seg < 4; ++seg)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r4 = wave * 4 + u, row = r4 * 4 + (lane >> 4);
      const int m = min(m0 + row, M - 1);
      const unsigned voff = (unsigned)(((size_t)m * 512 + seg * 128 + (((lane & 15) ^ (row & 15)) * 8)) * 2);
      tq_glds16(src, voff, lds0 + seg * TQ_SEG_BYTES + r4 * 1024);
    }
}

// acc[i][j] = bias + tile (tokens j*32 .. +31) x W^T (columns wave*64 + i*32 .. +31).  VT: operands swapped (a lane owns a channel).
template <bool VT>
__device__ __forceinline__ void tq_kloop(const TokGemmBlock &blk, const unsigned char *tile, int wave, int lane, floatx16 (&acc)[2][4]) {
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if constexpr (VT) {
      const float b = blk.bias[wave * 64 + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = b;
    } else {
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
  }
  // packed [wave4][k16 32][i4 4][lane 64][8 halfs] (pack_tok_weights): this wave's fragments are (wave >> 1, k16, (wave & 1) * 2 + i)
  const tq_u32x4 *wp = reinterpret_cast<const tq_u32x4 *>(blk.w) + (size_t)(wave >> 1) * (32 * 4 * 64) + ((wave & 1) * 2) * 64 + lane;
  constexpr int D = 3;                    // weight prefetch distance in k-steps (4 spills)
  tq_u32x4 wr[D][2];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) wr[d][i] = wp[(d * 4 + i) * 64];
  // token fragments: row j*32 + lr, chunk (2*(k16&7) + lh) ^ (lr & 15) of segment k16 >> 3 - the even part of the XOR is applied per
  // k-step (one v_xor), the odd part and the row sit in the base.  The four fragments of a k-step rotate through ONE register set: the
  // fragment of (k+1, j) is requested right behind the MFMAs of (k, j) - six MFMAs (192 cycles) cover its latency.
  const unsigned char *xb = tile + lr * 256 + ((lh ^ (lr & 1)) * 16);
  const unsigned xe = (unsigned)(lr & 14);
  half8 bf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const half8 *>(xb + j * 8192 + (xe << 4));
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int slot = k % D;
    half8 af[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<half8 *>(&wr[slot][i]);
    __builtin_amdgcn_sched_barrier(0);      // pin the prefetch (hipcc otherwise sinks the loads next to their use)
    if (k + D < 32) {
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[slot][i] = wp[((k + D) * 4 + i) * 64];
    }
    const int kn = k + 1;
    const unsigned char *xn = xb + (kn >> 3) * TQ_SEG_BYTES + (((unsigned)(2 * (kn & 7)) ^ xe) << 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kn < 32) bf[j] = *reinterpret_cast<const half8 *>(xn + j * 8192);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__global__ __launch_bounds__(TQ_THREADS, 1) void tok_qkv_kernel(TokGemmArgs p, int n_units, int units_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tq_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)tq_smem;
  unsigned char *stage = tq_smem + TQ_TILE_BYTES + wave * TQ_STAGE_BYTES;
  const int u0 = blockIdx.x * units_per_wg, u1 = min(u0 + units_per_wg, n_units);
  int cur_tile = -1;
  for (int u = u0; u < u1; ++u) {
    const int tile = u / p.nblk, b = u - tile * p.nblk;
    const TokGemmBlock &blk = p.blk[b];
    const int m0 = tile * TQ_ROWS;
    if (tile != cur_tile) {
      __syncthreads();                                     // every wave is done with the previous tile
      tq_tile_dma(p.in, m0, p.M, wave, lane, lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur_tile = tile;
    }
    floatx16 acc[2][4];
    if (blk.vt) {
      tq_kloop<true>(blk, tq_smem, wave, lane, acc);
      // acc[i][j][r]: channel c = wave*64 + i*32 + lr, token j*32 + (r&3) + 8*(r>>2) + 4*lh.  Per 32-token slice: stage [channel][token
      // in vt order] (64-byte rows, 16-byte units XORed with (c >> 1) & 3), then 16-byte stores along the token axis of the image
      f16 *img = (f16 *)blk.out;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][q * 4 + e];
            const int c = i * 32 + lr, t = 8 * q + 4 * lh;           // first of 4 consecutive tokens inside the slice
            const int pos = vt_col(t);                                // a multiple of 4
            *reinterpret_cast<half4 *>(stage + c * 64 + ((((pos >> 3) ^ ((c >> 1) & 3)) * 16) + (pos & 4) * 2)) = hv;
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = v * 16 + (lane >> 2), unit = lane & 3;
          const uint4 val = *reinterpret_cast<const uint4 *>(stage + c * 64 + ((unit ^ ((c >> 1) & 3)) * 16));
          const int m = m0 + j * 32 + unit * 8;                     // the 8 tokens of a unit lie in one group of 16: one hypothesis
          if (m < p.M) {
            const int bb = m / p.tokens, t = m - bb * p.tokens;
            const int col = blk.coff + wave * 64 + c, h = col >> 7, d = col & 127;
            f16 *row = img + (((size_t)bb * 4 + h) * 128 + d) * 416;
            *reinterpret_cast<uint4 *>(row + (t & ~15) + (unit & 1) * 8) = val;
            // the image is 416 tokens wide: the lanes that store a hypothesis' last 16 tokens also zero the columns behind them
            if ((t & ~15) + 16 == p.tokens)
              for (int z = p.tokens; z < 416; z += 16) *reinterpret_cast<uint4 *>(row + z + (unit & 1) * 8) = uint4{0u, 0u, 0u, 0u};
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
    } else {
      tq_kloop<false>(blk, tq_smem, wave, lane, acc);
      // fp16 rows: per 32-token slice the wave's 64 columns through its staging (128-byte rows, 16-byte units XORed with row & 7),
      // then whole 128-byte row segments
      const bool relu = blk.relu != 0;
      f16 *obase = (f16 *)blk.out + blk.coff + wave * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][rg * 4 + e];
            if (relu) hv = __builtin_elementwise_max(hv, half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
            *reinterpret_cast<half4 *>(stage + lr * 128 + (((i * 4 + rg) ^ (lr & 7)) * 16) + lh * 8) = hv;
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = v * 8 + (lane >> 3), unit = lane & 7;
          const uint4 val = *reinterpret_cast<const uint4 *>(stage + row * 128 + ((unit ^ (row & 7)) * 16));
          const int m = m0 + j * 32 + row;
          if (m < p.M) *reinterpret_cast<uint4 *>(obase + (size_t)m * blk.ld + unit * 8) = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
    }
  }
}

// The same in-projections for ONE or TWO hypotheses (a tracking frame, src/estimater.py:250-268).  There the kernel above is 24 units on
// 24 CUs, each with 67 MFLOP of its own: 14.6 us, bound by the matrix pipes of a tenth of the chip.  Here a workgroup is (32 tokens,
// block, 128 of its 512 columns) - 13 x 6 x 4 = 312 per hypothesis, two per CU - and its 8 waves are 4 quarters of K x 2 halves of the
// columns: 16 weight fragments per wave (coalesced 1-KB loads of pack_tok_weights' image, all requested up front), the token tile -
// 32 KB, contiguous - copied to LDS once (row pitch 512 + 8 halfs), 16 MFMAs, the four K quarters added through LDS in order, bias,
// and the rows or the transposed V image written as the kernel above writes them.  K in four quarters: a summation order of its own (the
// few-image size class, like conv_small.hip), not the bit pattern of tok_gemm.hip.
#define TS_PITCH 520                               // halfs per token row in LDS
#define TS_PLD 132                                 // floats per token row of a partial tile (128 columns + 4)
#define TS_LDS_BYTES (4 * 32 * TS_PLD * 4)         // 4 K quarters x 32 tokens x 128 columns fp32 (67.6 KB); the token tile (33 KB) lies in the same bytes first
__global__ __launch_bounds__(512, 2) void tok_qkv_small_kernel(TokGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ts_smem[];
  f16 *tile = reinterpret_cast<f16 *>(ts_smem);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
  const int kq = w & 3, chh = w >> 2;                        // K quarter, column half (64 columns: fragments i = 2 chh, 2 chh + 1)
  const int g = blockIdx.x & 3, b = (blockIdx.x >> 2) % p.nblk, tl = (blockIdx.x >> 2) / p.nblk;
  const TokGemmBlock &blk = p.blk[b];
  const int m0 = tl * 32;
  // weights: fragment (g, k16, i) of the packed image is 64 lanes x 16 bytes
  tq_u32x4 wf[8][2];
  {
    const tq_u32x4 *wp = reinterpret_cast<const tq_u32x4 *>(blk.w) + ((size_t)g * 32 + kq * 8) * 4 * 64 + (chh * 2) * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[k][i] = wp[(k * 4 + i) * 64];
  }
  // the token tile: rows m0 .. m0 + 31 (rows past M repeat the last one), 2048 pieces of 16 bytes
  {
    tq_u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 512, row = i >> 6, c8 = i & 63;
      v[u] = *reinterpret_cast<const tq_u32x4 *>(p.in + (size_t)min(m0 + row, p.M - 1) * 512 + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 512, row = i >> 6, c8 = i & 63;
      *reinterpret_cast<tq_u32x4 *>(tile + row * TS_PITCH + c8 * 8) = v[u];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (not a __syncthreads(): the weight loads stay in flight)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  floatx16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const half8 tf = *reinterpret_cast<const half8 *>(tile + lr * TS_PITCH + (kq * 8 + k) * 16 + lh * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8 *>(&wf[k][i]), tf, acc[i], 0, 0, 0);
  }
  __syncthreads();                                              // the tile is read: its bytes take the partial sums [K quarter][token][column]
  float *part = reinterpret_cast<float *>(ts_smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      *reinterpret_cast<float4 *>(part + ((size_t)kq * 32 + lr) * TS_PLD + chh * 64 + i * 32 + rg * 8 + lh * 4) =
          make_float4(acc[i][rg * 4 + 0], acc[i][rg * 4 + 1], acc[i][rg * 4 + 2], acc[i][rg * 4 + 3]);
  __syncthreads();
  const int cbase = blk.coff + g * 128;
  if (!blk.vt) {
    // rows: thread = (token tid / 16, 8 columns): the four quarters in order, bias, ReLU, one 16-byte store
    const int tk = tid >> 4, c8 = (tid & 15) * 8, m = m0 + tk;
    if (m >= p.M) return;
    float v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v8[e] = part[(size_t)tk * TS_PLD + c8 + e];
#pragma unroll
    for (int q = 1; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] += part[((size_t)q * 32 + tk) * TS_PLD + c8 + e];
    half8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v8[e] + blk.bias[g * 128 + c8 + e];
      if (blk.relu) x = fmaxf(x, 0.f);
      hv[e] = (f16)x;
    }
    *reinterpret_cast<half8 *>((f16 *)blk.out + (size_t)m * blk.ld + cbase + c8) = hv;
    return;
  }
  // transposed V image: thread = (channel tid & 127, group of 16 tokens (tid >> 7) & 1), threads 0 .. 255; a group lies in one hypothesis
  if (tid >= 256) return;
  const int ch = tid & 127, grp = tid >> 7, mg = m0 + grp * 16;
  if (mg >= p.M) return;
  const int bb = mg / p.tokens, t0 = mg - bb * p.tokens;          // t0 is a multiple of 16
  const int col = cbase + ch, hh = col >> 7, d = col & 127;
  const float bias = blk.bias[g * 128 + ch];
  half8 lo, hi;                                                   // image order of a group of 16: tokens 0-3, 8-11 | 4-7, 12-15
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int row = grp * 16 + t;
    float x = part[(size_t)row * TS_PLD + ch];
#pragma unroll
    for (int q = 1; q < 4; ++q) x += part[((size_t)q * 32 + row) * TS_PLD + ch];
    const f16 hx = (f16)(x + bias);
    const int pos = vt_col(t);
    if (pos < 8) lo[pos] = hx;
    else hi[pos - 8] = hx;
  }
  f16 *rowp = (f16 *)blk.out + (((size_t)bb * 4 + hh) * 128 + d) * 416;
  *reinterpret_cast<half8 *>(rowp + t0) = lo;
  *reinterpret_cast<half8 *>(rowp + t0 + 8) = hi;
  if (t0 + 16 == p.tokens)                                        // the image is 416 tokens wide: zeros behind a hypothesis' last tokens
    for (int z = p.tokens; z < 416; z += 8) *reinterpret_cast<uint4 *>(rowp + z) = uint4{0u, 0u, 0u, 0u};
}

void tok_qkv_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)tok_qkv_kernel, TQ_LDS_BYTES});
  v.push_back({(const void *)tok_qkv_small_kernel, TS_LDS_BYTES});
}

int tok_qkv_small_max() {
  static const int small_max = getenv("FP_QKV_SMALL") ? atoi(getenv("FP_QKV_SMALL")) : 2;       // hypotheses of the pass up to which the few-image form runs (0: off)
  return small_max;
}

int launch_tok_qkv(fp_ctx *ctx, const TokGemmArgs &a, hipStream_t s, int hyp) {
  FP_REQUIRE(a.in && a.M >= 0 && a.nblk >= 1 && a.nblk <= TG_MAXBLK, "tok_qkv: bad arguments");
  if (a.M == 0) return FP_OK;
  FP_REQUIRE((double)a.M * 512 * 2.0 < 4294967296.0, "tok_qkv: M=%d too large for 32-bit lane offsets", a.M);
  for (int b = 0; b < a.nblk; ++b) {
    const TokGemmBlock &k = a.blk[b];
    FP_REQUIRE(k.w && k.bias && k.out, "tok_qkv: block %d has a null pointer", b);
    if (k.vt) FP_REQUIRE(a.tokens > 0 && a.tokens % 16 == 0 && a.tokens <= 416 && k.coff % 64 == 0, "tok_qkv: tokens=%d must be a multiple of 16, <= 416", a.tokens);
    else FP_REQUIRE(k.ld % 8 == 0 && k.coff % 64 == 0, "tok_qkv: bad output of block %d", b);
  }
  ProfScope ps(ctx, s, "linear", 2.0 * (double)a.M * 512.0 * 512.0 * a.nblk);
  if (hyp > 0 && hyp <= tok_qkv_small_max()) {
    hipLaunchKernelGGL(tok_qkv_small_kernel, dim3(((a.M + 31) / 32) * a.nblk * 4), dim3(512), TS_LDS_BYTES, s, a);
    FP_CHECK_HIP(hipGetLastError());
    return FP_OK;
  }
  const int n_tiles = (a.M + TQ_ROWS - 1) / TQ_ROWS, n_units = n_tiles * a.nblk;
  int n_wg = n_units < ctx->num_cu ? n_units : ctx->num_cu;
  const int upw = (n_units + n_wg - 1) / n_wg;
  n_wg = (n_units + upw - 1) / upw;
  hipLaunchKernelGGL(tok_qkv_kernel, dim3(n_wg), dim3(TQ_THREADS), TQ_LDS_BYTES, s, a, n_units, upw);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

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
#include <cstdlib>

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


// ------------------------------------------------------------------------------------------------------------------------
// 128-TOKEN FORM (round 4; the default).  What bounds the 64-token kernel above is the weight stream L2 -> registers: 1.5 MB of
// fragments per 64 tokens, 64 B/clk/CU at the MFMA rate against the ~30-35 the path delivers - a K loop takes ~14 k cycles where
// the MFMAs need 8.2 k.  With 128 tokens per workgroup a fragment feeds FOUR MFMAs (32 B/clk/CU) and the K loops run at the MFMA
// rate (stamps: 16.3 k cycles for the 2 x 256 MFMAs of the two waves of a SIMD) - but two 128-KB tiles do not fit 160 KB of LDS.
// They do not have to:
//   * ONE resident 128-KB tile Q (tok_gemm.hip's image).  The residual tokens land in it at the start; LayerNorm1 writes x1 over
//     them in place (the 64-token form does the same): A of linear1; then ff: A of linear2;
//   * the attention output, the A operand of the out-projection, is read exactly once per k: it streams through a 2-slot ring of
//     64-k segments (2 x 16 KB; 128-byte rows, chunk c at c ^ (row & 7)) by LDS-DMA, one barrier per segment: the barrier that says
//     "segment s has landed everywhere" also says "everybody is done with segment s - 1", whose slot the DMA of s + 1 then takes;
//     the wait is a counted vmcnt (the 8 weight loads issued since may stay in flight).  (Tried instead: the attention output
//     resident and the residual read from global memory straight into the accumulators - 8-byte pieces of 32 rows per instruction:
//     13 k cycles slower per tile than the ring's 7 barriers.)
//   * ff never needs a second tile: ReLU(linear1) waits in registers (packed fp16, 64 VGPRs) until every wave has left the K loop
//     that reads x1, then goes into the tile over x1.  x1 is still the residual of LayerNorm2 - so linear2's accumulators START at
//     b2 + x1 (each wave reads its own 64 columns of x1 back in the accumulator layout before it overwrites them);
//   * a wave owns 64 columns x 128 tokens (2 x 4 accumulator tiles); token fragments rotate through one register set; the first
//     weight fragments of a K loop are requested in front of the barrier that precedes it;
//   * LayerNorm statistics in one pass (sum and sum of squares, fp32), one barrier; the 16-token sums behind LayerNorm2 go through
//     LDS (the tile is dead by then) and come back one column per lane - 512 DPP adds per wave in the 64-token form.
// 160 KB of LDS (the LayerNorm reduction scratch lives in the idle ring), one workgroup per CU.  Per output element, against the 64-token
// form: linear2 adds its residual first instead of last, the LayerNorm variance is E[x^2] - mean^2 and the 16-token sums add tokens in
// ascending order - fp32 everywhere.
// ------------------------------------------------------------------------------------------------------------------------
#define H2_ROWS 128
#ifdef HM_STAMP
// diagnostic build only (make -B EXTRA=-DHM_STAMP; scripts/hm_stamps.py): s_memtime per wave at the phase boundaries of head_mlp128_kernel
__device__ unsigned long long g_hm_stamps[1024 * 8 * 8];
extern "C" __attribute__((visibility("default"))) int fp_dbg_hm_stamps(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hm_stamps), sizeof(g_hm_stamps)) == hipSuccess ? 0 : -1;
}
#define HSTAMP(i) do { if (blockIdx.x < 1024 && c.lane == 0) g_hm_stamps[((size_t)blockIdx.x * 8 + c.wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif
#define H2_SEG (H2_ROWS * 256)                    // one k segment of 128 of the resident tile: 32 KB
#define H2_TILE (4 * H2_SEG)                      // 128 KB
#define H2_SLOT (H2_ROWS * 128)                   // ring slot: 128 rows x 64 k: 16 KB
#define H2_RING_OFF H2_TILE
#define H2_RED_OFF H2_RING_OFF                    // reduction scratch [8 waves][128 tokens] x (sum, sum of squares) floats = 8 KB: in the idle ring
#define H2_LDS_BYTES (H2_TILE + 2 * H2_SLOT)      // 160 KB

__device__ __forceinline__ void h2_tile_dma(const f16 *src, int m0, int M, int wave, int lane, unsigned lds0) {
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int seg = 0; seg < 4; ++seg)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r4 = wave * 4 + u, row = r4 * 4 + (lane >> 4);
      const int m = min(m0 + row, M - 1);
      const unsigned voff = (unsigned)(((size_t)m * 512 + seg * 128 + (((lane & 15) ^ (row & 15)) * 8)) * 2);
      hm_glds16(src, voff, lds0 + seg * H2_SEG + r4 * 1024);
    }
}

// 64-k segment s8 (0 .. 7) of 128 rows of `src` -> ring slot: [row][128 B], chunk c (0 .. 7) at c ^ (row & 7); 2 DMA instructions per wave
__device__ __forceinline__ void h2_ring_dma(const f16 *src, int m0, int M, int s8, int wave, int lane, unsigned slot_addr) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r8 = wave * 2 + u, row = r8 * 8 + (lane >> 3);
    const int m = min(m0 + row, M - 1);
    const unsigned voff = (unsigned)(((size_t)m * 512 + s8 * 64 + (((lane & 7) ^ (row & 7)) * 8)) * 2);
    hm_glds16(src, voff, slot_addr + r8 * 1024);
  }
}

struct H2Ctx {
  int lane, wave, lr, lh;
};

// acc[i][j] = C0 + A (128 tokens x 512: the resident tile, or RING: the two ring slots) @ W^T for the wave's 64 columns; C0 = the bias (BIAS: it is the C operand of
// the first k-step's MFMAs - no 128 register copies) or what acc holds on entry.  `before_loop` runs once the first weight fragments
// have been requested: the barrier (and wait) that makes the tile valid, so the fragments' L2 latency passes behind it.
template <bool RING, bool BIAS, typename Pre>
__device__ __forceinline__ void h2_gemm(const H2Ctx &c, const f16 *w, const float *bias, const unsigned char *smem, floatx16 (&acc)[2][4], Pre before_loop,
                                        unsigned lds0 = 0, const f16 *ring_src = nullptr, int m0 = 0, int M = 0) {
  // packed [wave4][k16 32][i4 4][lane 64][8 halfs] (pack_tok_weights): this wave's fragments are (wave >> 1, k16, (wave & 1) * 2 + i)
  const hm_u32x4 *wp = reinterpret_cast<const hm_u32x4 *>(w) + (size_t)(c.wave >> 1) * (32 * 4 * 64) + ((c.wave & 1) * 2) * 64 + c.lane;
  constexpr int D = 3;
  hm_u32x4 wr[D][2];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) wr[d][i] = wp[(d * 4 + i) * 64];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (RING) {
    // RING: A streams through the two ring slots; segment 0 was requested by the caller in front of this call (and in front of the 6 weight
    // loads above, which may stay in flight)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    h2_ring_dma(ring_src, m0, M, 1, c.wave, c.lane, lds0 + H2_RING_OFF + H2_SLOT);
  }
  floatx16 bvec[2];
  if constexpr (BIAS) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 bv = *reinterpret_cast<const float4 *>(bias + c.wave * 64 + i * 32 + rg * 8 + c.lh * 4);
        bvec[i][rg * 4 + 0] = bv.x, bvec[i][rg * 4 + 1] = bv.y, bvec[i][rg * 4 + 2] = bv.z, bvec[i][rg * 4 + 3] = bv.w;
      }
  }
  __builtin_amdgcn_sched_barrier(0);
  before_loop();
  // token fragment (k16, j): resident tile: row j*32 + lr, chunk (2*(k16&7) + lh) ^ (lr & 15) of segment k16 >> 3 (tok_gemm.hip);
  // ring: slot (k16 >> 2) & 1, row j*32 + lr, chunk (2*(k16&3) + lh) ^ (lr & 7).  The even part of the XOR is applied per k-step, the odd
  // part and the row sit in the base
  const unsigned char *xb = RING ? smem + H2_RING_OFF + c.lr * 128 + ((c.lh ^ (c.lr & 1)) * 16) : smem + c.lr * 256 + ((c.lh ^ (c.lr & 1)) * 16);
  const unsigned xe = (unsigned)(c.lr & (RING ? 6 : 14));
  auto frag_addr = [&](int k, int j) __attribute__((always_inline)) -> const unsigned char * {
    if constexpr (RING) return xb + ((k >> 2) & 1) * H2_SLOT + j * 4096 + (((unsigned)(2 * (k & 3)) ^ xe) << 4);
    else return xb + (k >> 3) * H2_SEG + j * 8192 + (((unsigned)(2 * (k & 7)) ^ xe) << 4);
  };
  half8 bf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const half8 *>(frag_addr(0, j));
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
    __builtin_amdgcn_sched_barrier(0);
    const int kn = k + 1;
    const bool seg_edge = RING && (kn & 3) == 0 && kn < 32;        // the next k-step opens ring segment kn >> 2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], (BIAS && k == 0) ? bvec[i] : acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kn < 32 && !seg_edge) bf[j] = *reinterpret_cast<const half8 *>(frag_addr(kn, j));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (seg_edge) {
      // segment kn >> 2 was requested at the top of the previous segment; since then this wave has issued 8 weight loads (2 per k-step)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __syncthreads();                                          // landed everywhere, and everybody is done with the other slot
      if ((kn >> 2) + 1 < 8) h2_ring_dma(ring_src, m0, M, (kn >> 2) + 1, c.wave, c.lane, lds0 + H2_RING_OFF + (((kn >> 2) + 1) & 1) * H2_SLOT);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const half8 *>(frag_addr(kn, j));
    }
  }
}

// byte offset inside the resident tile of the 8-byte half chunk that holds channels wave*64 + i*32 + rg*8 + lh*4 .. +3 of token `row`
__device__ __forceinline__ unsigned h2_off(const H2Ctx &c, int row, int i, int rg) {
  return (unsigned)((c.wave >> 1) * H2_SEG + row * 256 + ((((c.wave & 1) * 8 + i * 4 + rg) ^ (row & 15)) * 16) + c.lh * 8);
}
// a copy of the lane coordinates the compiler cannot see through: addresses derived from it are recomputed where they are used instead
// of being shared between the phases of the kernel (and living in registers across the K loops in between)
__device__ __forceinline__ H2Ctx h2_fresh(const H2Ctx &c0) {
  H2Ctx c = c0;
  asm volatile("" : "+v"(c.lr), "+v"(c.lh), "+v"(c.lane));
  return c;
}

// LayerNorm statistics of every token over the 512 columns, ONE pass in fp32 (sum and sum of squares together: one barrier round instead
// of two; the rows are O(1) residual-stream values with a variance of the same order, so E[x^2] - mean^2 loses nothing that matters in
// fp32): lane -> partner lane -> the 8 waves through LDS in a fixed order.  Returns mean and rstd per token tile.
__device__ __forceinline__ void h2_stats(const H2Ctx &c0, float *red, const floatx16 (&acc)[2][4], float (&mean)[4], float (&rstd)[4]) {
  H2Ctx c = c0;
  asm volatile("" : "+v"(c.lr));          // the scratch addresses are recomputed per call: shared between the two calls they would live across two K loops
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s += acc[i][j][e];
        q += acc[i][j][e] * acc[i][j][e];
      }
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (c.lh == 0) *reinterpret_cast<float2 *>(&red[(c.wave * 128 + j * 32 + c.lr) * 2]) = make_float2(s, q);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float2 v = *reinterpret_cast<const float2 *>(&red[(w * 128 + j * 32 + c.lr) * 2]);
      s += v.x;
      q += v.y;
    }
    mean[j] = s * (1.f / 512.f);
    rstd[j] = rsqrtf(fmaxf(q * (1.f / 512.f) - mean[j] * mean[j], 0.f) + 1e-5f);
  }
}

__global__ __launch_bounds__(HM_THREADS, 1) void head_mlp128_kernel(HeadMlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char h2_smem[];
  H2Ctx c;
  const int tid = threadIdx.x;
  c.lane = tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.lr = c.lane & 31;
  c.lh = c.lane >> 5;
  const int m0 = blockIdx.x * H2_ROWS;
  unsigned char *Q = h2_smem;
  float *red = reinterpret_cast<float *>(h2_smem + H2_RED_OFF);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)h2_smem;

  HSTAMP(0);
  h2_tile_dma(p.tok, m0, p.M, c.wave, c.lane, lds0);                                   // 16 instructions: the residual tokens -> Q
  h2_ring_dma(p.att, m0, p.M, 0, c.wave, c.lane, lds0 + H2_RING_OFF);                  // 2: segment 0 of the attention output -> slot 0
  floatx16 acc[2][4];
  float mean[4], rstd[4];
  // ---- out-projection (A = attention output through the ring) ----
  h2_gemm<true, true>(c, p.w_out, p.b_out, h2_smem, acc, []() {}, lds0, p.att, m0, p.M);
  HSTAMP(1);
  // ---- + residual (Q) + LayerNorm1 -> x1 (fp16) in place into Q ----
  __syncthreads();                                               // every wave is done with the ring: it becomes the reduction scratch
  {
    const H2Ctx cr = h2_fresh(c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 32 + cr.lr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const half4 rq = *reinterpret_cast<const half4 *>(Q + h2_off(cr, row, i, rg));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][rg * 4 + e] += (float)rq[e];
        }
    }
  }
  h2_stats(c, red, acc, mean, rstd);
  const H2Ctx c1 = h2_fresh(c);
  float4 gv[2][4], bv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int col = c.wave * 64 + i * 32 + rg * 8 + c1.lh * 4;
      gv[i][rg] = *reinterpret_cast<const float4 *>(p.g1 + col);
      bv[i][rg] = *reinterpret_cast<const float4 *>(p.be1 + col);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float nm = -mean[j] * rstd[j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        half4 hv;
        hv[0] = (f16)__builtin_fmaf(__builtin_fmaf(acc[i][j][rg * 4 + 0], rstd[j], nm), gv[i][rg].x, bv[i][rg].x);
        hv[1] = (f16)__builtin_fmaf(__builtin_fmaf(acc[i][j][rg * 4 + 1], rstd[j], nm), gv[i][rg].y, bv[i][rg].y);
        hv[2] = (f16)__builtin_fmaf(__builtin_fmaf(acc[i][j][rg * 4 + 2], rstd[j], nm), gv[i][rg].z, bv[i][rg].z);
        hv[3] = (f16)__builtin_fmaf(__builtin_fmaf(acc[i][j][rg * 4 + 3], rstd[j], nm), gv[i][rg].w, bv[i][rg].w);
        *reinterpret_cast<half4 *>(Q + h2_off(c1, j * 32 + c1.lr, i, rg)) = hv;
      }
  }
  HSTAMP(2);
  // ---- linear1 + ReLU: ff waits in registers (packed fp16) until every wave has left the K loop that reads x1 ----
  h2_gemm<false, true>(c, p.w1, p.b1, h2_smem, acc, [&]() __attribute__((always_inline)) { __syncthreads(); });      // (x1 is complete)
  HSTAMP(3);
  half4 ff[2][4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][rg * 4 + e];
        ff[i][j][rg] = __builtin_elementwise_max(hv, half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
      }
  const H2Ctx c2 = h2_fresh(c);
  float4 b2v[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) b2v[i][rg] = *reinterpret_cast<const float4 *>(p.b2 + c.wave * 64 + i * 32 + rg * 8 + c2.lh * 4);
  __syncthreads();
  // ---- linear2: accumulators start at b2 + x1 (this wave's own 64 columns of x1, read back before ff overwrites them) ----
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned char *q = Q + h2_off(c2, j * 32 + c2.lr, i, rg);
        const half4 xq = *reinterpret_cast<const half4 *>(q);
        acc[i][j][rg * 4 + 0] = b2v[i][rg].x + (float)xq[0];
        acc[i][j][rg * 4 + 1] = b2v[i][rg].y + (float)xq[1];
        acc[i][j][rg * 4 + 2] = b2v[i][rg].z + (float)xq[2];
        acc[i][j][rg * 4 + 3] = b2v[i][rg].w + (float)xq[3];
        *reinterpret_cast<half4 *>(q) = ff[i][j][rg];
      }
  HSTAMP(4);
  h2_gemm<false, false>(c, p.w2, nullptr, h2_smem, acc, [&]() __attribute__((always_inline)) { __syncthreads(); });   // (ff is complete)
  HSTAMP(5);
  // ---- LayerNorm2 statistics -> sums of the normalised rows over groups of 16 tokens.  The sums cross lanes (a lane owns one token):
  // instead of 512 DPP adds per wave the normalised values go through LDS - the tile is dead once every wave has left the last K loop
  // (the barrier inside h2_stats) - as fp32 [token][this wave's 64 columns], 64 tokens at a time (16 KB per wave, 16-byte units XORed
  // with token & 15), and come back one COLUMN per lane: 16 reads + adds per group, tokens in ascending order, and a group's 64 sums
  // leave as one 256-byte store.
  h2_stats(c, red, acc, mean, rstd);
  {
    const H2Ctx c3 = h2_fresh(c);
    unsigned char *tb = Q + c.wave * 16384;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * h + jj, row = jj * 32 + c3.lr;
        const float nm = -mean[j] * rstd[j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            *reinterpret_cast<float4 *>(tb + row * 256 + (((i * 8 + rg * 2 + c3.lh) ^ (row & 15)) * 16)) =
                make_float4(__builtin_fmaf(acc[i][j][rg * 4 + 0], rstd[j], nm), __builtin_fmaf(acc[i][j][rg * 4 + 1], rstd[j], nm),
                            __builtin_fmaf(acc[i][j][rg * 4 + 2], rstd[j], nm), __builtin_fmaf(acc[i][j][rg * 4 + 3], rstd[j], nm));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) sum += *reinterpret_cast<const float *>(tb + (gq * 16 + t) * 256 + ((((c3.lane >> 2) ^ t)) * 16) + (c3.lane & 3) * 4);
        const int g = (m0 + h * 64 + gq * 16) >> 4;            // global 16-token group (16 divides 400: never two hypotheses)
        if (g * 16 < p.M) p.gsum[(size_t)g * 512 + c.wave * 64 + c3.lane] = sum;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  HSTAMP(6);
}

void head_mlp_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)head_mlp_kernel, HM_LDS_BYTES});
  v.push_back({(const void *)head_mlp128_kernel, H2_LDS_BYTES});
}

int launch_head_mlp(fp_ctx *ctx, const HeadMlpArgs &a, hipStream_t s) {
  FP_REQUIRE(a.att && a.tok && a.w_out && a.w1 && a.w2 && a.b_out && a.b1 && a.b2 && a.g1 && a.be1 && a.gsum, "head_mlp: null argument");
  FP_REQUIRE(a.M >= 0 && a.M % 16 == 0, "head_mlp: M=%d must be a multiple of 16", a.M);
  if (a.M == 0) return FP_OK;
  FP_REQUIRE((double)a.M * 1024.0 < 4294967296.0, "head_mlp: M=%d too large for 32-bit lane offsets", a.M);
  ProfScope ps(ctx, s, "linear", 3.0 * 2.0 * (double)a.M * 512.0 * 512.0);
  static const bool form64 = getenv("FP_HEADMLP64") != nullptr;       // A/B knob: the 64-token form (two resident tiles)
  // 1 .. 4 hypotheses (tracking): the 64-token form.  Every workgroup streams all 1.5 MB of weights whatever its tile holds, so a launch
  // of a handful of workgroups lasts one workgroup's life: 26.8 us for seven 64-token tiles against 38.5 for four 128-token ones at one
  // hypothesis.  (Its own size class, like split-K in the 3x3 kernel: the two forms differ in the last bits, see the 128-token form's header.)
  if (form64 || a.M <= 1600) hipLaunchKernelGGL(head_mlp_kernel, dim3((a.M + 63) / 64), dim3(HM_THREADS), HM_LDS_BYTES, s, a);
  else hipLaunchKernelGGL(head_mlp128_kernel, dim3((a.M + H2_ROWS - 1) / H2_ROWS), dim3(HM_THREADS), H2_LDS_BYTES, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

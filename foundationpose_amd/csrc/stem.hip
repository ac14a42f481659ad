// 7x7 / stride-2 / pad-3 stem convolution (encodeA.0 / encoderA.0: ConvBNReLU(C_in=6 -> 64), network_modules.py:37-50 via
// refine_network.py:37 / score_network.py:36) for gfx950 with the input REGION of an output block resident in LDS.
//
// The implicit-GEMM kernel of conv.hip gathers every tap of every output pixel separately (one 16-byte pixel per lane and tap
// through LDS-DMA: each input pixel is staged 12 times, 51 FLOP per staged byte) and is bound by that fill: 336 TF/s useful.
// Here a 16 x 16 block of output pixels needs a 37 x 37 region of the (r,g,b,x,y,z,0,0) fp16 input = 22 KB, staged ONCE:
//   * region rows in LDS as two column-parity planes (20 + 20 sixteen-byte slots): tap (ky,kx) of output column ox is slot
//     ox + (kx >> 1) of plane kx & 1 - the 16 columns of a lane group read 16 consecutive slots, the two output rows of a
//     32-pixel MFMA tile sit 1280 B (= 5 x 256) apart: every ds_read_b128 lane group covers all 64 banks once;
//   * one 16-byte input pixel IS one tap's k-slice (8 channels); a k-step of 16 = two taps, chosen per lane half: 25 steps;
//   * the weights (64 couts x 50 tap slices, 50 KB) are staged once per workgroup, in MFMA-fragment order;
//   * 8 waves = two output blocks per workgroup (waves 0-3 / 4-7), each wave 2 tiles of 32 pixels x 64 couts (4 accumulator
//     tiles): 100 MFMAs per wave and block; persistent workgroups walk the block pairs so that the weights are loaded once.
// Epilogue: accumulators start at the folded BN bias, ReLU, fp16, staged per wave, 16-byte NHWC stores (2 KB per output row
// of a block).  The accumulation order of an output element (taps in order, 8 channels per tap) does not depend on the block.
#include "common.h"

#define ST_THREADS 512
#define ST_BLK 16                         // output block edge
#define ST_REG 37                         // input region edge: 2*16 + 5
#define ST_ROW_SLOTS 40                   // 16-byte slots per region row: plane 0 (even columns) 20 + plane 1 (odd columns) 20
#define ST_REGION_BYTES (ST_REG * ST_ROW_SLOTS * 16)      // 23680
#define ST_REGION_INSTR ((ST_REG * ST_ROW_SLOTS + 63) / 64)   // 24 DMA instructions (the last one partly past the region)
#define ST_REGION_ALLOC (ST_REGION_INSTR * 1024)              // 24576
#define ST_W_BYTES (2 * 25 * 1024)        // weight fragments [co tile 2][k-step 25][lane 64][8 halfs]
#define ST_STAGE_LD 72                    // halfs per staged output pixel (64 + 8 pad)
#define ST_LDS_BYTES (ST_W_BYTES + 4 * ST_REGION_ALLOC)       // 149504: weights + two sets of two regions (the next pair of blocks lands
                                                              // while this one is multiplied); the output staging re-uses the current set

__device__ __forceinline__ void st_glds16(const f16 *g, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

__global__ __launch_bounds__(ST_THREADS, 1) void stem7x7_kernel(ConvArgs p, const f16 *__restrict__ zero_page, int n_pairs, int blocks_x,
                                                                int blocks_per_img, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int half = wave >> 2, wq = wave & 3;                 // block of the pair, wave within the block
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)st_smem;
  unsigned char *wlds = st_smem;

  // ---- weights -> LDS in fragment order: fragment f = i*25 + s, lane (lr, lh) = W[i*32 + lr][(2s + lh)*8 .. +8] ----
  for (int f = wave; f < 50; f += 8) {
    const int i = f / 25, s = f - i * 25;
    st_glds16(p.w + (size_t)(i * 32 + lr) * p.Kpad + (2 * s + lh) * 8, lds0 + f * 1024);
  }
  float4 bias[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) bias[i][rg] = *reinterpret_cast<const float4 *>(p.bias + i * 32 + rg * 8 + lh * 4);

  // ---- B fragments: lane = output pixel (row lr >> 4, column lr & 15) of a 2-row tile; tile jj of a block covers rows 2jj, 2jj+1
  // of it.  Region slot of tap (ky,kx): (4 jj + 2 (lr>>4) + ky) * 40 + (kx & 1) * 20 + (lr & 15) + (kx >> 1).
  const unsigned pix_off = (unsigned)(((2 * (lr >> 4)) * ST_ROW_SLOTS + (lr & 15)) * 16);
  unsigned tap_off[25];                                      // this lane half's tap of every k-step (tap 49 = zero weights: any slot)
#pragma unroll
  for (int s = 0; s < 25; ++s) {
    const int t = min(2 * s + lh, 48), ky = t / 7, kx = t - ky * 7;
    tap_off[s] = (unsigned)((ky * ST_ROW_SLOTS + (kx & 1) * 20 + (kx >> 1)) * 16);
  }

  // region of block `half` of pair `pr` -> LDS set `set`: slot q of region row r = input pixel (2 oy0 - 3 + r, 2 ox0 - 3 + 2 (q % 20) + q / 20);
  // outside the image: zeros.  Exactly ST_REGION_INSTR / 4 = 6 DMA instructions per wave (the counted waits below rely on it).
  auto load_region = [&](int pr, int set) __attribute__((always_inline)) {
    const int blk = min(pr * 2 + half, n_blocks - 1);
    const int img = blk / blocks_per_img, bi = blk - img * blocks_per_img;
    const int oy0 = (bi / blocks_x) * ST_BLK, ox0 = (bi % blocks_x) * ST_BLK;
    const f16 *src_img = p.in + (size_t)img * p.H * p.W * 8;
#pragma unroll
    for (int v = 0; v < ST_REGION_INSTR / 4; ++v) {
      const int u = wq + 4 * v;
      const int slot = u * 64 + lane, r = slot / ST_ROW_SLOTS, q = slot - r * ST_ROW_SLOTS;
      const int iy = 2 * oy0 - 3 + r, ix = 2 * ox0 - 3 + 2 * (q % 20) + q / 20;
      const bool ok = r < ST_REG && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      st_glds16(ok ? src_img + ((size_t)iy * p.W + ix) * 8 : zero_page, lds0 + ST_W_BYTES + (set * 2 + half) * ST_REGION_ALLOC + u * 1024);
    }
  };
  static_assert(ST_REGION_INSTR % 4 == 0, "six region DMAs per wave");

  int it = 0;
  if ((int)blockIdx.x < n_pairs) load_region(blockIdx.x, 0);
  for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x, ++it) {
    const int set = it & 1;
    const int blk = min(pair * 2 + half, n_blocks - 1);      // an odd block count: the last pair computes its block twice (second copy not stored)
    const bool store = pair * 2 + half < n_blocks;
    const int img = blk / blocks_per_img, bi = blk - img * blocks_per_img;
    const int oy0 = (bi / blocks_x) * ST_BLK, ox0 = (bi % blocks_x) * ST_BLK;
    const bool more = pair + (int)gridDim.x < n_pairs;
    // the other set is free (its staging was read back before the barrier that ended the previous iteration): the NEXT pair's
    // regions stream into it while this pair is multiplied.  vmcnt counts loads, LDS-DMA and stores together in issue order: this
    // wave's DMAs of THIS pair are older than its 8 output stores of the previous iteration and the 6 DMAs just issued.
    if (more) {
      load_region(pair + gridDim.x, set ^ 1);
      if (it == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");      // no stores yet: only the 6 new DMAs may be in flight
      else asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][rg * 4 + 0] = bias[i][rg].x;
          acc[i][j][rg * 4 + 1] = bias[i][rg].y;
          acc[i][j][rg * 4 + 2] = bias[i][rg].z;
          acc[i][j][rg * 4 + 3] = bias[i][rg].w;
        }
    const unsigned char *region = st_smem + ST_W_BYTES + (set * 2 + half) * ST_REGION_ALLOC;
    const unsigned char *rb = region + pix_off + (wq * 2) * (4 * ST_ROW_SLOTS * 16);       // tile jj = wq*2 + j
#pragma unroll
    for (int s = 0; s < 25; ++s) {
      half8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8 *>(wlds + (i * 25 + s) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const half8 *>(rb + j * (4 * ST_ROW_SLOTS * 16) + tap_off[s]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave is done with this set's regions: staging may overwrite them
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: ReLU, fp16, per-wave staging of one 32-pixel tile at a time (in this set's regions), 16-byte NHWC stores ----
    f16 *stage = reinterpret_cast<f16 *>(st_smem + ST_W_BYTES + set * 2 * ST_REGION_ALLOC) + (size_t)wave * (32 * ST_STAGE_LD);
    const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)fmaxf(acc[i][j][rg * 4 + e], lo);
          *reinterpret_cast<half4 *>(&stage[lr * ST_STAGE_LD + i * 32 + rg * 8 + lh * 4]) = hv;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own staging writes have landed (same-wave LDS ops are in order)
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4 *>(&stage[(u * 8 + (lane >> 3)) * ST_STAGE_LD + (lane & 7) * 8]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int px = u * 8 + (lane >> 3), c16 = lane & 7;
        const int oy = oy0 + (wq * 2 + j) * 2 + (px >> 4), ox = ox0 + (px & 15);
        // 4 stores per tile = 8 per iteration and wave: the counted wait at the top relies on it (`store` is false only in the
        // last pair, after which nothing is waited for by count)
        f16 *dst = (f16 *)p.out + (((size_t)img * p.Ho + oy) * p.Wo + ox) * 64 + c16 * 8;
        if (store) *reinterpret_cast<uint4 *>(dst) = v[u];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // staging read back before the next tile overwrites it
    }
    __builtin_amdgcn_s_barrier();                            // every wave has read its staging back: the next iteration's DMAs may overwrite this set
  }
}

bool stem_supported(const ConvArgs &a) {
  return a.KH == 7 && a.KW == 7 && a.stride == 2 && a.pad == 3 && a.Cin == 8 && a.Cout == 64 && a.Kpad >= 400 && a.out_mode == 0 && !a.res &&
         !a.post_add && a.out_ld == 64 && a.split_m >= a.M && a.Ho % ST_BLK == 0 && a.Wo % ST_BLK == 0 && a.H == 2 * a.Ho && a.W == 2 * a.Wo;
}

void stem_kernel_lds(std::vector<KernelLds> &v) { v.push_back({(const void *)stem7x7_kernel, ST_LDS_BYTES}); }

int launch_stem(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  const int blocks_x = a.Wo / ST_BLK, per_img = blocks_x * (a.Ho / ST_BLK), n_blocks = a.Nimg * per_img, n_pairs = (n_blocks + 1) / 2;
  const int grid = n_pairs < ctx->num_cu ? n_pairs : ctx->num_cu;
  hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(ST_THREADS), ST_LDS_BYTES, s, a, (const f16 *)ctx->zero_page, n_pairs, blocks_x, per_img, n_blocks);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

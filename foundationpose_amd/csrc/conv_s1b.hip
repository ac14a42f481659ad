// 3x3 / stride-1 / pad-1 convolutions on 40x40 maps - 128 -> 128 (the four ResnetBasicBlock convolutions of encodeA / encoderA) and
// 256 -> 256 (those of encodeAB's first stage; network_modules.py:73-111 via refine_network.py:39-47) - on the structure of conv_s2.hip
// instead of conv_halo.hip.
// Bit-identical to conv3x3_halo_dma_kernel<40>: an output element accumulates in the same order - 32-channel groups, inside a group
// kernel row by kernel row, inside a row tap by tap, inside a tap the two 16-channel MFMA steps - from the same bias, and the epilogue
// rounds once after the same fp32 residual add (tests/test_gpu_kernels.py::test_band_kernel_equals_halo_kernel).  Full-batch launches
// (504 images, kernel trace, first form with 16-channel chunks): 215 / 242 us without / with residual against 223 / 247 us.
//
// With four K chunks of 32 channels the 8-wave halo kernel spends 39 % of a workgroup's life in prologue and epilogue, one workgroup
// per CU.  Here
//   * a workgroup (4 waves) owns 8 output rows x 40 = 320 pixels of ONE image (5 tiles per image: no tile straddles two) x a block of
//     128 output channels (the blocks of a pixel tile are neighbours in the grid: they share the band in L2): wave (c, p) owns 64 couts (2 accumulator tiles) x the 160 pixels of rows 4p .. 4p+3 (5 pixel tiles of 32);
//   * per 32-channel group the band goes to LDS once by LDS-DMA as two 16-channel planes (10 input rows x 42 columns of 32-byte
//     pixels each, row pitch 56 slots: 56 = 40 (mod 16) keeps the lane -> bank map of a 32-pixel tile unchanged across its row breaks;
//     the two 16-byte halves of slot q swapped when bit 3 of q is set), double buffered (72 KB), and feeds all 9 taps of both planes;
//   * weights never touch LDS: packed at load time in MFMA-fragment order (s2_pack_weights, order 1: group, tap, 16-channel half,
//     co tile), each wave streams the fragments of its 64 couts L2 -> registers three steps ahead: 2 KB per 10 MFMAs;
//   * two workgroups per CU (<= 256 VGPRs each).
// Epilogue: accumulators start at the folded BN bias; residual rows staged through LDS and added in fp32, ReLU, one rounding to fp16,
// 16-byte NHWC stores (with the channel-concat addressing of the last encodeA layer: out_ld / split_m / coff_hi).
// Default for the 128 -> 128 layers (FP_C128_BAND=0 sends them back to the halo kernel).  The 256 -> 256 layers take it with FP_C128_BAND=2
// only: bit-identical too, but in an alternating A/B of the bench step 35.54 / 35.66 / 35.60 ms against 35.50 / 35.54 / 35.57 - with
// eight K groups the halo kernel's prologue / epilogue share is small and its weights are shared through LDS (here each cout block
// re-reads the band and each wave streams its own weights).
#include "common.h"

#define B1_THREADS 256
#define B1_W 40
#define B1_ROWS 8
#define B1_P 56
#define B1_BROWS (B1_ROWS + 2)
#define B1_PLANE_BYTES (B1_BROWS * B1_P * 32)             // one 16-channel plane of the band: 560 slots of 32 B = 17 920 B (a multiple of 256)
#define B1_BAND_INSTR 36                                  // 1-KB DMA instructions per 32-channel group (two planes = 35 KB), 9 per wave
#define B1_BAND_BYTES (B1_BAND_INSTR * 1024)
#define B1_DPW (B1_BAND_INSTR / 4)
#define B1_NPT 5
#define B1_STAGE_LD 72                                    // halfs per staged pixel: 64 couts + 8 pad
#define B1_LDS_BYTES (2 * B1_BAND_BYTES)

__device__ __forceinline__ void b1_glds16(const void *g, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

template <bool RES>
__global__ __launch_bounds__(B1_THREADS, 2) void conv3x3_s1_band_kernel(ConvArgs p, const f16 *__restrict__ wpk, const f16 *__restrict__ zero_page, int n_tiles,
                                                                        int n_cob) {
  constexpr int W = B1_W, H = B1_W, P = B1_P, NPT = B1_NPT, DPW = B1_DPW, CT = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char b1_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int ch = wave & 1, ph = wave >> 1;                 // cout half, pixel half
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)b1_smem;
  const int tile = xcd_remap(blockIdx.x, n_tiles);
  const int cob = tile % n_cob, pt = tile / n_cob;          // 128-cout block, pixel tile
  const int img = pt / (H / B1_ROWS), oy0 = (pt - img * (H / B1_ROWS)) * B1_ROWS;
  const int ngrp = p.Cin >> 5;                             // 32-channel groups

  // ---- band DMA: instruction u = wave + 4 v, lane l -> slot 32 u + l/2 of the group's 1120 (plane ks = slot / 560, slot q of the plane),
  // physical half l&1 = logical half ^ bit 3 of q ----
  unsigned src_off[DPW];                                   // byte offset of (pixel, plane, logical half) in the input, group 0; ~0u: zeros
#pragma unroll
  for (int v = 0; v < DPW; ++v) {
    const int qq = (wave + 4 * v) * 32 + (lane >> 1);
    const int ks = qq >= B1_BROWS * P ? 1 : 0, q = qq - ks * (B1_BROWS * P), hl = (lane & 1) ^ ((q >> 3) & 1);
    const int b = q / P, c = q - b * P;
    const int iy = oy0 - 1 + b, ix = c - 1;
    const bool ok = b < B1_BROWS && c < W + 2 && iy >= 0 && iy < H && ix >= 0 && ix < W;
    src_off[v] = ok ? (unsigned)((((img * H + iy) * W + ix) * p.Cin + ks * 16 + hl * 8) * 2) : 0xffffffffu;
  }
  auto band_dma = [&](int grp, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < DPW; ++v) {
      const char *src = src_off[v] != 0xffffffffu ? (const char *)p.in + src_off[v] + grp * 64 : (const char *)zero_page;
      b1_glds16(src, lds0 + buf * B1_BAND_BYTES + (wave + 4 * v) * 1024);
    }
  };
  // ---- B fragments: lane = output pixel 32 j + lr of the wave's 160 (local row jr of its 4, column ox), k half lh; tap (ky, kx) of
  // that pixel is slot qb[j] + ky P + kx (band row 4 ph + jr + ky holds input row oy0 + 4 ph + jr + ky - 1, slot c holds column c - 1)
  int qb[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pl = 32 * j + lr, jr = pl / W, ox = pl - jr * W;
    qb[j] = (((4 * ph + jr) * P + ox) << 5) | (lh << 4);
  }
  // ---- A fragments: this wave's stream of 1-KB fragments, (group, tap, 16-channel half, co tile) in order (s2_pack_weights order 1) ----
  const half8 *wp = reinterpret_cast<const half8 *>(wpk) + ((size_t)(cob * 2 + ch) * ngrp * 18 * CT) * 64 + lane;
  half8 aq[3][CT];
  int f = 0;
  auto a_fetch = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) aq[slot][ct] = wp[(size_t)(f * CT + ct) * 64];
    ++f;
  };
  band_dma(0, 0);
#pragma unroll
  for (int d = 0; d < 3; ++d) a_fetch(d);

  floatx16 acc[CT][NPT];
  {
    const int co0 = cob * 128 + ch * 64;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.bias + co0 + ct * 32 + rg * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
          acc[ct][j][rg * 4 + 0] = bv.x;
          acc[ct][j][rg * 4 + 1] = bv.y;
          acc[ct][j][rg * 4 + 2] = bv.z;
          acc[ct][j][rg * 4 + 3] = bv.w;
        }
      }
  }
  for (int grp = 0; grp < ngrp; ++grp) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp + 1 < ngrp) band_dma(grp + 1, (grp + 1) & 1);
    const unsigned char *band = b1_smem + (grp & 1) * B1_BAND_BYTES;
    // step st = tap * 2 + ks (tap = ky * 3 + kx): the halo kernel's order inside a 32-channel group
    auto b_read = [&](int st, int j) __attribute__((always_inline)) -> half8 {
      const int t = st >> 1, ks = st & 1, ky = t / 3, kx = t - ky * 3;
      int x = qb[j];
      asm volatile("" : "+v"(x));
      x += (ky * P + kx) << 5;
      return *reinterpret_cast<const half8 *>(band + ks * B1_PLANE_BYTES + (x ^ ((x >> 4) & 16)));
    };
    half8 b[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) b[j] = b_read(0, j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[st % 3][ct], b[j], acc[ct][j], 0, 0, 0);
        if (st < 17) b[j] = b_read(st + 1, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      a_fetch(st % 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: one 32-pixel tile at a time through this wave's staging rows (32 px x 64 couts) ----
  f16 *stage = reinterpret_cast<f16 *>(b1_smem) + (size_t)wave * (32 * B1_STAGE_LD);
  const float lo = p.relu ? 0.f : -__builtin_inff();
  const int m_wave = (img * H + oy0 + 4 * ph) * W;        // first output pixel of this wave (flattened over images)
  const int co0 = cob * 128 + ch * 64;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    if constexpr (RES) {
      // residual rows of the tile: 16-byte pieces, 8 lanes per pixel row of 64 couts -> staging, then added in fp32 in the accumulator layout
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int px = u * 8 + (lane >> 3), c8 = lane & 7;
        const int m = m_wave + 32 * j + px;
        const uint4 rv = *reinterpret_cast<const uint4 *>(p.res + (size_t)m * p.Cout + co0 + c8 * 8);
        *reinterpret_cast<uint4 *>(&stage[px * B1_STAGE_LD + c8 * 8]) = rv;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        f16 *sp = &stage[lr * B1_STAGE_LD + ct * 32 + rg * 8 + lh * 4];
        float v[4] = {acc[ct][j][rg * 4 + 0], acc[ct][j][rg * 4 + 1], acc[ct][j][rg * 4 + 2], acc[ct][j][rg * 4 + 3]};
        if constexpr (RES) {
          const half4 rq = *reinterpret_cast<const half4 *>(sp);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rq[e];
        }
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (f16)fmaxf(v[e], lo);
        *reinterpret_cast<half4 *>(sp) = hv;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint4 v4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const uint4 *>(&stage[(u * 8 + (lane >> 3)) * B1_STAGE_LD + (lane & 7) * 8]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m_wave + 32 * j + u * 8 + (lane >> 3);
      const bool hi = m >= p.split_m;
      const long long orow = hi ? (long long)(m - p.split_m) : (long long)m;
      *reinterpret_cast<uint4 *>((f16 *)p.out + orow * p.out_ld + (hi ? p.coff_hi : 0) + co0 + (lane & 7) * 8) = v4[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

bool s1b_supported(const ConvArgs &a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.out_mode == 0 && !a.post_add && a.H == B1_W && a.W == B1_W && a.Cin % 32 == 0 &&
         a.Cout % 128 == 0 && a.Kpad == 9 * a.Cin && a.out_ld % 8 == 0 && a.coff_hi % 8 == 0 && a.wpk != nullptr && !(a.splitk && a.ksplit > 1) &&
         (long long)a.Nimg * B1_W * B1_W * a.Cin * 2 < (1ll << 31);        // 32-bit byte offsets into the input (above: the general 3x3 kernel, 4 GB)
}

void conv_s1b_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)conv3x3_s1_band_kernel<true>, B1_LDS_BYTES});
  v.push_back({(const void *)conv3x3_s1_band_kernel<false>, B1_LDS_BYTES});
}

int launch_conv_s1b(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  FP_REQUIRE(s1b_supported(a), "launch_conv_s1b: unsupported layer");
  const int n_cob = a.Cout / 128, n_tiles = a.Nimg * (B1_W / B1_ROWS) * n_cob;
  if (a.res) {
    hipLaunchKernelGGL(conv3x3_s1_band_kernel<true>, dim3(n_tiles), dim3(B1_THREADS), B1_LDS_BYTES, s, a, a.wpk, (const f16 *)ctx->zero_page, n_tiles, n_cob);
  } else {
    hipLaunchKernelGGL(conv3x3_s1_band_kernel<false>, dim3(n_tiles), dim3(B1_THREADS), B1_LDS_BYTES, s, a, a.wpk, (const f16 *)ctx->zero_page, n_tiles, n_cob);
  }
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// 3x3 / stride-2 / pad-1 convolution onto 20x20 maps (encodeAB.2: 256 -> 512 at 40x40 -> 20x20; ConvBNReLU,
// network_modules.py:37-50 via refine_network.py:46 / score_network.py:45) for gfx950 with the input BAND of an output tile resident
// in LDS.
//
// The implicit-GEMM kernel of conv.hip gathers every tap of every output pixel separately (each input pixel is staged 2.25 times
// per 128-cout block through LDS-DMA, 85 FLOP per staged byte) and is bound by that fill.  Here
//   * a workgroup (4 waves) owns 160 consecutive output pixels (8 rows of 20, flattened over images: a tile may straddle two
//     images) x 256 (or 128) output channels; wave w owns 64 (32) couts x ALL 160 pixels = CT x 5 accumulator tiles of
//     v_mfma_f32_32x32x16_f16;
//   * per 16-channel input chunk the band of input rows the tile needs (<= 18 rows x 41 columns of 32-byte pixels = 24 KB) goes to
//     LDS ONCE by LDS-DMA, double buffered, and feeds all 9 taps.  A band row is stored as two column-parity planes (even input
//     columns, then odd ones), so the stride-2 taps of consecutive output pixels are consecutive LDS slots; the two 16-byte halves
//     of pixel slot q are swapped when bit 3 of q is set, and the row pitch P obeys 2P = W_out (mod 16): every 16-lane group of a
//     ds_read_b128 covers all 64 banks once, also across the row breaks of a 32-pixel tile.  The swizzle is applied on the DMA's
//     source address (an LDS-DMA destination is lane-linear);
//   * the waves split the output channels and nothing else, so no weight is shared between waves: weights never touch LDS.  They are
//     packed at load time in MFMA-fragment order (s2_pack_weights) and each wave streams its own fragments L2 -> registers with
//     coalesced 1-KB loads, three taps ahead;
//   * the B fragment of (tap t + 1, pixel tile j) is read right behind the MFMAs of (tap t, tile j): five MFMA pairs hide its latency;
//   * two workgroups share a CU (48 KB LDS, <= 256 VGPRs each): one's chunk loop runs beside the other's prologue / epilogue.
// Epilogue: accumulators start at the folded BN bias, ReLU, fp16, staged per wave through the (free) band buffers, 16-byte NHWC stores.
// The accumulation order of an output element (chunks in order, taps in order, 16 channels per step) does not depend on the tile.
// The 64 -> 128 layer onto 40x40 maps stays on the implicit-GEMM kernel: the same form (4 rows of 40, P = 84) was measured 226 against
// 208 us in a bench step - a 16-channel chunk pass touches every 128-byte line of the band and uses 32 bytes of it, and with Cin = 64 the
// bands in flight on an XCD (64 workgroups x 104 KB of lines) overflow its 4-MB L2 before the other three chunks come by.
#include "common.h"

#define S2_THREADS 256
#define S2_STAGE_PAD 8

__device__ __forceinline__ void s2_glds16(const void *g, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

template <int WO, int ROWS, int CT>
struct S2Cfg {
  static constexpr int P = 42;                                         // band row pitch in 32-byte pixel slots: WO even + (WO + 1) odd columns, 2P = WO (mod 16)
  static_assert(WO == 20, "pitch chosen for 20-wide outputs");
  static constexpr int BROWS = 2 * ROWS + 2;                           // + 1 for the zero row between two images inside a tile
  static constexpr int NPT = ROWS * WO / 32;                           // pixel tiles of 32
  static constexpr int BAND_INSTR = ((BROWS * P + 31) / 32 + 3) / 4 * 4;   // 1-KB DMA instructions per chunk (a multiple of the 4 waves)
  static constexpr int BAND_BYTES = BAND_INSTR * 1024;
  static constexpr int DPW = BAND_INSTR / 4;
  static constexpr int STAGE_LD = CT * 32 + S2_STAGE_PAD;              // halfs per staged output pixel
  static constexpr int LDS_BYTES = 2 * BAND_BYTES;
  static_assert(ROWS * WO == NPT * 32 && (2 * P - WO) % 16 == 0 && P >= 2 * WO + 1, "tile / pitch");
  static_assert(4 * 32 * STAGE_LD * 2 <= BAND_BYTES, "epilogue staging fits one band buffer");
};

// Fragment order: [cout block][wave 4][chunk Cin/16][tap 9][co tile CT][lane 64][8 halfs]; lane (lr, lh) of a fragment holds
// W[co = ((cb*4 + wave)*CT + ct)*32 + lr][k = tap*Cin + chunk*16 + lh*8 .. +8] of the [Cout][Kpad] image (K index = tap*Cin + ci).
// order 1 (conv_s1b.hip): [wave group][32-channel group][tap 9][16-channel half 2][co tile CT][lane][8 halfs]
__global__ void s2_pack_kernel(const f16 *__restrict__ w, int Cin, int Kpad, int CT, size_t total, f16 *__restrict__ out, int order) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7, lane = (idx >> 3) & 63;
  size_t rest = idx >> 9;
  const int nch = Cin / 16;
  const int ct = rest % CT;
  rest /= CT;
  int tap, chunk;
  if (order == 1) {
    const int ks = rest % 2;
    rest /= 2;
    tap = rest % 9;
    rest /= 9;
    chunk = (int)(rest % (nch / 2)) * 2 + ks;
    rest /= nch / 2;
  } else {
    tap = rest % 9;
    rest /= 9;
    chunk = rest % nch;
    rest /= nch;                                                        // = cb*4 + wave
  }
  const int co = ((int)rest * CT + ct) * 32 + (lane & 31);
  const int k = tap * Cin + chunk * 16 + (lane >> 5) * 8 + e;
  out[idx] = w[(size_t)co * Kpad + k];
}

template <int WO, int ROWS, int CT>
__global__ __launch_bounds__(S2_THREADS, 2) void conv3x3_s2_kernel(ConvArgs p, const f16 *__restrict__ wpk, const f16 *__restrict__ zero_page,
                                                                    int n_tiles, int n_cob) {
  using C = S2Cfg<WO, ROWS, CT>;
  constexpr int HO = WO, HI = 2 * WO, WI = 2 * WO, P = C::P, NPT = C::NPT, DPW = C::DPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)s2_smem;

  const int tile = xcd_remap(blockIdx.x, n_tiles);
  const int cb = tile % n_cob, pt = tile / n_cob;                       // the cout blocks of a pixel tile are neighbours: they share the band in L2
  const int g0 = pt * ROWS;                                             // first flattened output row (image * HO + oy)
  const int img0 = g0 / HO, oy0 = g0 - img0 * HO;
  const int jb = HO - oy0;                                              // first local row that belongs to the next image (>= ROWS: none)
  const int nch = p.Cin >> 4;

  // ---- band DMA: instruction u = wave + 4 v, lane l -> pixel slot q = 32 u + l/2, physical half l&1 = logical half ^ bit 3 of q ----
  // band row b: b <= 2 jb -> input row 2 oy0 + b - 1 of image img0; b > 2 jb -> input row b - 2 jb - 2 of image img0 + 1 (-1: zero row)
  unsigned src_off[DPW];                                                // byte offset of (pixel, logical half) in the input, chunk 0; ~0u: zeros
#pragma unroll
  for (int v = 0; v < DPW; ++v) {
    const int q = (wave + 4 * v) * 32 + (lane >> 1), hl = (lane & 1) ^ ((q >> 3) & 1);
    const int b = q / P, c = q - b * P;
    const int ix = c < WO ? 2 * c : 2 * (c - WO) - 1;
    const bool second = b > 2 * jb;
    const int img = img0 + (second ? 1 : 0), iy = second ? b - 2 * jb - 2 : 2 * oy0 + b - 1;
    const bool ok = b < C::BROWS && iy >= 0 && iy < HI && ix >= 0 && ix < WI && img < p.Nimg;
    src_off[v] = ok ? (unsigned)((((img * HI + iy) * WI + ix) * p.Cin + hl * 8) * 2) : 0xffffffffu;
  }
  auto band_dma = [&](int ch, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < DPW; ++v) {
      const char *src = src_off[v] != 0xffffffffu ? (const char *)p.in + src_off[v] + ch * 32 : (const char *)zero_page;
      s2_glds16(src, lds0 + buf * C::BAND_BYTES + (wave + 4 * v) * 1024);
    }
  };

  // ---- B fragments: lane = output pixel 32 j + lr of the tile (local row jr, column ox), k half lh.  Tap (ky, kx) of that pixel is pixel
  // slot qb[j] + ky P + {WO, 0, WO + 1}[kx] (odd plane slot WO + s = input column 2 s - 1, even plane slot s = column 2 s) ----
  int qb[NPT];                                                          // byte address of (slot qb, half lh) before the swizzle
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pl = 32 * j + lr, jr = pl / WO, ox = pl - jr * WO;
    qb[j] = (((2 * jr + (jr >= jb ? 1 : 0)) * P + ox) << 5) | (lh << 4);
  }

  // ---- A fragments: this wave's stream of 1-KB fragments, (chunk, tap, co tile) in order ----
  const half8 *wp = reinterpret_cast<const half8 *>(wpk) + ((size_t)(cb * 4 + wave) * nch * 9 * CT) * 64 + lane;
  half8 aq[3][CT];                                                      // ring: taps t, t+1, t+2
  int f = 0;                                                            // next fragment group (tap) to fetch
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
    const int co0 = (cb * 4 + wave) * CT * 32;
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

  for (int ch = 0; ch < nch; ++ch) {
    // this wave's part of chunk ch has landed (everything this wave has in flight: the three prefetched weight taps are needed next
    // anyway); after the barrier every wave's part has, and every wave is done reading the other buffer (chunk ch - 1)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (ch + 1 < nch) band_dma(ch + 1, (ch + 1) & 1);
    const unsigned char *band = s2_smem + (ch & 1) * C::BAND_BYTES;
    // B fragment (tap t, pixel tile j): read while the MFMAs of (tap t - 1, tiles j + 1 ..) run - its register is free once MFMA
    // (t - 1, j) has issued - so a read has five MFMA pairs to land.  Only tap 0 of a chunk waits for its reads (the band is new).
    auto b_read = [&](int t, int j) __attribute__((always_inline)) -> half8 {
      const int ky = t / 3, kx = t - ky * 3;
      int x = qb[j];
      asm volatile("" : "+v"(x));                                       // keep the 45 tap addresses out of registers: three VALU ops per read instead
      x += (ky * P + (kx == 1 ? 0 : WO + (kx == 2 ? 1 : 0))) << 5;
      return *reinterpret_cast<const half8 *>(band + (x ^ ((x >> 4) & 16)));      // swap the halves when bit 3 of the slot (bit 8 of the address) is set
    };
    half8 b[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) b[j] = b_read(0, j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[t % 3][ct], b[j], acc[ct][j], 0, 0, 0);
        if (t < 8) b[j] = b_read(t + 1, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      a_fetch(t % 3);                                                   // tap t + 3 (possibly of the next chunk; past the end: the padding of the packed image)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the trailing weight prefetches
  __builtin_amdgcn_s_barrier();                                         // every wave is done with the bands: staging may overwrite them
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: ReLU, fp16, one 32-pixel tile at a time through this wave's staging rows, 16-byte NHWC stores ----
  f16 *stage = reinterpret_cast<f16 *>(s2_smem) + (size_t)wave * (32 * C::STAGE_LD);
  const float lo = p.relu ? 0.f : -__builtin_inff();
  const size_t m0 = (size_t)g0 * WO;
  const int co0 = (cb * 4 + wave) * CT * 32;
  constexpr int LPP = CT * 4;                                           // lanes per pixel row (16 bytes each)
  constexpr int PPI = 64 / LPP;                                         // pixels per store instruction
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (f16)fmaxf(acc[ct][j][rg * 4 + e], lo);
        *reinterpret_cast<half4 *>(&stage[lr * C::STAGE_LD + ct * 32 + rg * 8 + lh * 4]) = hv;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint4 v[32 / PPI];
#pragma unroll
    for (int u = 0; u < 32 / PPI; ++u) v[u] = *reinterpret_cast<const uint4 *>(&stage[(u * PPI + lane / LPP) * C::STAGE_LD + (lane % LPP) * 8]);
#pragma unroll
    for (int u = 0; u < 32 / PPI; ++u) {
      const size_t m = m0 + 32 * j + u * PPI + lane / LPP;
      if (m < (size_t)p.M) *reinterpret_cast<uint4 *>((f16 *)p.out + m * p.Cout + co0 + (lane % LPP) * 8) = v[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// co tiles of 32 per wave: 2 -> 256-cout blocks, 1 -> 128-cout blocks; 0: this Cout is not handled here.  A function of Cout alone,
// so that the weights can be packed when the network is loaded (the spatial size is not known then).
int s2_ct_for(int Cout) { return Cout % 256 == 0 ? 2 : Cout % 128 == 0 ? 1 : 0; }

bool s2_supported(const ConvArgs &a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.out_mode == 0 && !a.res && !a.post_add && a.out_ld == a.Cout && a.split_m >= a.M &&
         a.H == a.W && a.Ho == a.Wo && a.H == 2 * a.Ho && a.Wo == 20 && a.Cin >= 16 && a.Cin % 16 == 0 && a.Kpad >= 9 * a.Cin &&
         s2_ct_for(a.Cout) != 0;
}

size_t s2_packed_halfs(int Cout, int Cin) { return (size_t)Cout * 9 * Cin + 3 * 2 * 512; }    // + three taps of prefetch past the end

int s2_pack_weights(const f16 *d_w, int Cout, int Cin, int Kpad, f16 *d_out, hipStream_t s, int ct_force, int order) {
  const int ct = ct_force ? ct_force : s2_ct_for(Cout);
  FP_REQUIRE(ct != 0 && Cin % 16 == 0 && Kpad >= 9 * Cin && (order == 0 || Cin % 32 == 0), "s2_pack_weights: unsupported layer");
  const size_t total = (size_t)Cout * 9 * Cin;
  FP_CHECK_HIP(hipMemsetAsync(d_out + total, 0, (s2_packed_halfs(Cout, Cin) - total) * sizeof(f16), s));
  hipLaunchKernelGGL(s2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_w, Cin, Kpad, ct, total, d_out, order);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

template <int WO, int ROWS, int CT>
static int s2_launch(fp_ctx *ctx, const ConvArgs &a, const f16 *wpk, hipStream_t s) {
  using C = S2Cfg<WO, ROWS, CT>;
  const int rows = a.Nimg * WO, n_pt = (rows + ROWS - 1) / ROWS, n_cob = a.Cout / (128 * CT), n_tiles = n_pt * n_cob;
  hipLaunchKernelGGL((conv3x3_s2_kernel<WO, ROWS, CT>), dim3(n_tiles), dim3(S2_THREADS), C::LDS_BYTES, s, a, wpk, (const f16 *)ctx->zero_page, n_tiles,
                     n_cob);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

void conv_s2_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)conv3x3_s2_kernel<20, 8, 2>, S2Cfg<20, 8, 2>::LDS_BYTES});
  v.push_back({(const void *)conv3x3_s2_kernel<20, 8, 1>, S2Cfg<20, 8, 1>::LDS_BYTES});
}

int launch_conv_s2(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  FP_REQUIRE(a.wpk && s2_supported(a), "launch_conv_s2: unsupported layer or no packed weights");
  return s2_ct_for(a.Cout) == 2 ? s2_launch<20, 8, 2>(ctx, a, a.wpk, s) : s2_launch<20, 8, 1>(ctx, a, a.wpk, s);
}

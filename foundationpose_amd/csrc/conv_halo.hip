// 3x3 / stride-1 / pad-1 convolution for gfx950 with the input HALO BAND resident in LDS.
//
// 93 % of the network FLOPs are 3x3 stride-1 convolutions on 40x40 (C=128|256) and 20x20 (C=512) maps
// (learning/models/network_modules.py:73-111 via refine_network.py:37-50).  An im2col-style implicit GEMM re-fetches every
// activation 9 times (once per tap) from L2 into LDS.  Here a workgroup owns a run of consecutive output pixels (flattened
// over images) x 128 output channels and, per 32-channel input chunk,
//   * brings the input rows it needs ONCE into LDS (the "halo band") and feeds all 9 taps from it: a tap shift is just a
//     different LDS address per lane (+-1 pixel, +-1 row; taps outside the image select a zero chunk),
//   * streams the 3 taps of one kernel row of weights at a time (128 co x 32 ci x 3 = 24 KB, double buffered) by LDS-DMA.
// conv3x3_halo_dma_kernel: 8 waves, 512 px, band by LDS-DMA (double buffered), v_mfma_f32_32x32x16_f16.  The diagnostic forms
// (the earlier register-staged kernel that carries the in-kernel cycle stamps, the persistent-workgroup experiment) live in
// diag/conv_halo_diag.inc and are compiled only by `make -B EXTRA=-DHALO_STAMP` (never shipped); identical results.
// Epilogue: accumulators start at the bias; residual (staged through LDS) + ReLU (+ positional embedding) in fp32, one
// rounding to fp16, LDS transpose, 16-byte row-contiguous NHWC stores.
#include "common.h"
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16-byte register value (plain vector loads / stores in IR)


#define HL_BM 128
#define HL_CK 32
#define HL_SLD 136  // halfs per staged output row (128 + 8 pad) -> 272 B
#define HL_PS 40    // halfs per halo pixel (32 channels + 8 pad = 80 B): conflict-free ds_read_b128 AND every tap
                    // shift / k-step is a compile-time immediate offset from ONE base register per pixel tile

#ifdef HALO_STAMP
// diagnostic build only (make STAMP=1): per-wave cycle sums of [wait + barrier] and [group body], see scripts/halo_stamps.py
__device__ unsigned long long g_halo_stamps[4096 * 8 * 8];
extern "C" __attribute__((visibility("default"))) int fp_dbg_halo_stamps(unsigned long long *host, int clear) {
  if (host) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_halo_stamps), sizeof(g_halo_stamps));
  if (clear) {
    static unsigned long long z[4096 * 8 * 8];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_halo_stamps), z, sizeof(z));
  }
  return 0;
}
#define STAMP(x) x
#else
#define STAMP(x)
#endif

template <int V>
struct IC {
  static constexpr int value = V;
};

// LDS-DMA issued from inline asm.  Through __builtin_amdgcn_global_load_lds the compiler marks a "flat access that may
// touch LDS" as pending until the next full drain, and while that mark is up EVERY wait it inserts for an LDS fragment read
// is s_waitcnt lgkmcnt(0) (and every barrier drains vmcnt(0)) - no LDS read can stay in flight under the MFMAs.  Hidden in
// asm, the DMA is outside its bookkeeping: fragment reads get counted lgkmcnt(n); the DMA's completion is waited for by
// the explicit s_waitcnt vmcnt(n) in front of the barriers below.  m0 = wave-uniform LDS byte address, lane i lands at +16 i.
__device__ __forceinline__ void glds16(const f16 *sbase, unsigned voff_bytes, f16 *l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)l);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(la) : "memory");
}

#ifdef HALO_STAMP      // diagnostic builds only: the register-staged form that carries the in-kernel cycle stamps
#define HALO_DIAG_SECTION 1
#include "diag/conv_halo_diag.inc"
#undef HALO_DIAG_SECTION
#endif

// ------------------------------------------------------------------------------------------------------------------------
// DEFAULT FORM.  8-wave tile (512 px x 128 co, one workgroup per CU) with the halo band staged by LDS-DMA as well, double
// buffered:
//   * halo pixels are 64 bytes, UNPADDED (so a DMA instruction's 64 lanes x 16 B land linearly: 16 pixels x 4 channel groups);
//     bank conflicts are avoided by an XOR swizzle instead: pixel P keeps channel group c at position c ^ ((P >> 2) & 3),
//     applied on the SOURCE address of the DMA (the destination of an LDS-DMA is always lane-linear).  A ds_read_b128 lane
//     group ({0-3,12-15,20-27}, ...) then reads 16 pixels whose four members of each residue class mod 4 carry four
//     different swizzle values, for any tile alignment and tap shift: conflict free;
//   * no staging registers, no ds_write of the band, no barrier pair at the top of a chunk: the band of chunk cc+1 streams
//     into the other buffer while chunk cc is multiplied, one DMA instruction per pixel-tile visit;
//   * v_mfma_f32_32x32x16_f16, a wave holds 64 co x 128 px (2 x 4 accumulator tiles); a kernel row is 24 "tile visits" of 2
//     MFMAs, pixel fragments rotate through 4 registers, weight fragments of the next step go to a spare set.
// (A 16x16x32 MFMA form of this tile was measured too: ~7 % more cycles at a ~8 % higher clock, the same time.)
// ------------------------------------------------------------------------------------------------------------------------
template <int W, int TM>
struct HaloCfgD {
  static constexpr int MAXSLOT = (W - 1 + TM - 1) / W + 1 + 2;
  static constexpr int NB = MAXSLOT * W + 2;                       // band pixels in use (pixel p at index p+1)
  static constexpr int HQ = ((NB + 15) / 16 + 7) / 8;              // halo DMA instructions per wave per chunk (8 waves)
  static constexpr int BPX = HQ * 8 * 16;                          // band pixels allocated
  static constexpr int HBUF_HALFS = BPX * 32;
  static constexpr int ZERO_OFF = 2 * HBUF_HALFS;                  // half offset of the zero chunk
  static constexpr int WBUF_OFF = ZERO_OFF + 8;
  static constexpr int WBUF_HALFS = 3 * HL_BM * HL_CK;
  static constexpr int LDS_HALFS_MAIN = WBUF_OFF + 2 * WBUF_HALFS;
  static constexpr int LDS_HALFS_EPI = TM * HL_SLD;
  static constexpr int LDS_BYTES = 2 * (LDS_HALFS_MAIN > LDS_HALFS_EPI ? LDS_HALFS_MAIN : LDS_HALFS_EPI);
};

// Epilogue of a tile (both schedules): residual (staged through LDS) + ReLU (+ positional embedding) in fp32, one rounding to fp16,
// LDS transpose, 16-byte row-contiguous NHWC stores.
template <int NT, bool RES, bool POST>
__device__ __forceinline__ void halo_epilogue(const ConvArgs &p, const int m0, const int c0, f16 *lds, floatx16 (&acc)[2][NT]) {
  constexpr int NPW = 4, TM = 32 * NT * NPW, NTH = 128 * NPW, PXW = 32 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NPW, wn = wave % NPW;
  const int lr = lane & 31, lh = lane >> 5;
  // ---------------- epilogue (as in halo_tile) ----------------
  f16 *stage = lds;   // [TM px][HL_SLD]
  constexpr int NRES = TM * 16 / NTH;
  u32x4 rv[NRES];
  if constexpr (RES) {
#pragma unroll
    for (int u = 0; u < NRES; ++u) {
      const int idx = tid + NTH * u, px = idx >> 4, c16 = idx & 15;
      const int m = min(m0 + px, p.M - 1);
      rv[u] = *reinterpret_cast<const u32x4 *>(p.res + (size_t)m * p.Cout + c0 + c16 * 8);
    }
  }
  const float lo = p.relu ? 0.f : -__builtin_inff();
  __syncthreads();          // every wave is done with the band / weight images: the staging tile may overwrite them
  if constexpr (RES) {
#pragma unroll
    for (int u = 0; u < NRES; ++u) {
      const int idx = tid + NTH * u, px = idx >> 4, c16 = idx & 15;
      *reinterpret_cast<u32x4 *>(&stage[px * HL_SLD + c16 * 8]) = rv[u];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int pxl = wn * PXW + j * 32 + lr;
    float4 pvs[2][4];
    if constexpr (POST) {
      const int prow = min(m0 + pxl, p.M - 1) % p.post_period;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          pvs[i][rg] = *reinterpret_cast<const float4 *>(p.post_add + (size_t)prow * p.Cout + c0 + wm * 64 + i * 32 + rg * 8 + lh * 4);
    }
    half4 rq[2][4];
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rq[i][rg] = *reinterpret_cast<const half4 *>(&stage[pxl * HL_SLD + wm * 64 + i * 32 + rg * 8 + lh * 4]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4] = {acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
        if constexpr (RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rq[i][rg][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
        if constexpr (POST) {
          const float4 pv = pvs[i][rg];
          v[0] += pv.x;
          v[1] += pv.y;
          v[2] += pv.z;
          v[3] += pv.w;
        }
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (f16)v[e];
        *reinterpret_cast<half4 *>(&stage[pxl * HL_SLD + wm * 64 + i * 32 + rg * 8 + lh * 4]) = hv;
      }
    }
  }
  __syncthreads();
  constexpr int NOUT = TM * 16 / NTH, OB = NOUT < 8 ? NOUT : 8;
#pragma unroll
  for (int i0 = 0; i0 < NOUT; i0 += OB) {
    u32x4 ov[OB];
#pragma unroll
    for (int u = 0; u < OB; ++u) {
      if (i0 + u >= NOUT) break;                 // (NOUT = 12 for the 384-pixel tile: the second batch is half full)
      const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
      ov[u] = *reinterpret_cast<const u32x4 *>(&stage[px * HL_SLD + c16 * 8]);
    }
#pragma unroll
    for (int u = 0; u < OB; ++u) {
      if (i0 + u >= NOUT) break;
      const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
      const int m = m0 + px;
      if (m < p.M) {
        const bool hi = m >= p.split_m;
        const long long orow = hi ? (long long)(m - p.split_m) : (long long)m;
        const int coff = hi ? p.coff_hi : 0;
        *reinterpret_cast<u32x4 *>((f16 *)p.out + orow * p.out_ld + coff + c0 + c16 * 8) = ov[u];
      }
    }
  }
}

template <int W, int NT, bool RES, bool POST, bool SPLIT = false>
__device__ __forceinline__ void halo_tile_dma(const ConvArgs &p, const int m0, const int c0, f16 *lds, const int ks = 0) {
  constexpr int NPW = 4, TM = 32 * NT * NPW;
  using C = HaloCfgD<W, TM>;
  constexpr int H = W;
  // (512 threads)
  constexpr int PXW = 32 * NT;          // pixels per wave
  constexpr int HQ = C::HQ;
  constexpr int WQ = 24 / (2 * NPW);    // weight DMA instructions per wave per group
  STAMP(const unsigned long long t_entry = __builtin_amdgcn_s_memtime(); const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();)
  f16 *wbuf = lds + C::WBUF_OFF;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NPW, wn = wave % NPW;
  const int lr = lane & 31, lh = lane >> 5;
  // static priority for the second-dispatched half of the workgroup (waves 4-7 are the arbitration loser of each SIMD pair at
  // equal priority): measured A/B in one session, 39.55 / 39.33 -> 38.99 / 38.92 ms per step, 1139 / 1145 -> 1153 / 1157 TF/s
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int GR0 = m0 / W - 1;                       // global input row (n*H + iy) held by band row 0
  const int total_rows = p.Nimg * H;
  // input-channel chunks [cbeg, nchunk) of this workgroup: all of them, or (SPLIT) share ks of p.ksplit equal shares
  const int call = p.Cin / HL_CK;
  const int cbeg = SPLIT ? ks * (call / p.ksplit) : 0;
  const int nchunk = SPLIT ? cbeg + call / p.ksplit : call;

  // ---- B fragments: band index of the top-left tap of this lane's pixel in tile 0 (tile j, taps: see load_b) ----
  const int pb = m0 + wn * PXW + lr - GR0 * W - W;
  unsigned vmp[(NT + 2) / 3];          // 9 validity bits (ky*3+kx) per 32-pixel tile, three tiles per register
#pragma unroll
  for (int t = 0; t < (NT + 2) / 3; ++t) vmp[t] = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int m = min(m0 + wn * PXW + j * 32 + lr, p.M - 1);
    const int gr = m / W, ox = m - gr * W, oy = gr % H;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) vmp[j / 3] |= 1u << ((j % 3) * 9 + ky * 3 + kx);
      }
  }
  // ---- A fragments: row co = wm*64 + i*32 + lr, channel group ks*2 + lh at position ^ ((co>>2)&3) (same for co and co+32)
  int wa[2];
  {
    const int co = wm * 64 + lr;
    wa[0] = co * 32 + ((lh ^ ((co >> 2) & 3)) * 8);
    wa[1] = co * 32 + (((2 + lh) ^ ((co >> 2) & 3)) * 8);
  }

  // ---- halo band by LDS-DMA: instruction h of wave w lands band pixels (h*8 + w)*16 .. +15, lane i = (pixel i/4, position i%4)
  const int hpix_max = total_rows * W - 1;
  auto halo_dma = [&](int cc, int hb, auto hc) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    int l4 = lane >> 2;
    asm volatile("" : "+v"(l4));                    // recompute per use: six hoisted offsets would cost six registers
    const int P = (h * 8 + wave) * 16 + l4;
    const int c = (lane & 3) ^ ((P >> 2) & 3);
    const int gp = min(max(GR0 * W + P - 1, 0), hpix_max);      // clamped: what lands for rows outside the tensor is never read
    const unsigned off = (unsigned)(gp * p.Cin + cc * HL_CK + c * 8) * 2u;
    glds16(p.in, off, lds + hb * C::HBUF_HALFS + (h * 8 + wave) * 512);
  };
  // ---- weights (as in halo_tile): row (g%8)*16 + lane/4 -> (row>>2)&3 = (lane>>4)&3
  const unsigned woff = (unsigned)(((wave * 16 + (lane >> 2)) * p.Kpad + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2);
  auto wstage = [&](int cc, int ky, int buf, auto q0c, auto q1c) __attribute__((always_inline)) {
    constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value;
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
      const int g0 = q * 2 * NPW;
      const f16 *sb = p.w + (size_t)(c0 + (g0 & 7) * 16) * p.Kpad + (ky * 3 + (g0 >> 3)) * p.Cin + cc * HL_CK;
      glds16(sb, woff, wbuf + buf * C::WBUF_HALFS + (q * 2 * NPW + wave) * 512);
    }
  };

  // accumulators start at the bias (fp32): a lane owns channels wm*64 + i*32 + rg*8 + lh*4 + (0..3) of its pixels
  floatx16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float4 bv = *reinterpret_cast<const float4 *>(p.bias + c0 + wm * 64 + i * 32 + rg * 8 + lh * 4);
      if (SPLIT && ks != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);       // the bias rides in split 0
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j][rg * 4 + 0] = bv.x;
        acc[i][j][rg * 4 + 1] = bv.y;
        acc[i][j][rg * 4 + 2] = bv.z;
        acc[i][j][rg * 4 + 3] = bv.w;
      }
    }

  if (tid < 1) *reinterpret_cast<u32x4 *>(&lds[C::ZERO_OFF]) = u32x4{0, 0, 0, 0};
  {
    auto all = [&](auto hc) __attribute__((always_inline)) {
      if constexpr (decltype(hc)::value < HQ) halo_dma(cbeg, cbeg & 1, hc);
    };
    all(IC<0>{}), all(IC<1>{}), all(IC<2>{}), all(IC<3>{}), all(IC<4>{}), all(IC<5>{}), all(IC<6>{}), all(IC<7>{});
    static_assert(HQ <= 8, "extend the list");
  }
  wstage(cbeg, 0, 0, IC<0>{}, IC<WQ>{});
  int g = 0;
  STAMP(unsigned long long t_wait = 0; unsigned long long t_body = 0; unsigned long long t_prev = __builtin_amdgcn_s_memtime();)
  for (int cc = cbeg; cc < nchunk; ++cc) {
    const f16 *halo = lds + (cc & 1) * C::HBUF_HALFS;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky, ++g) {
      const int buf = g & 1;
      STAMP(unsigned long long ta = __builtin_amdgcn_s_memtime();)
      // weights(g) - and at ky=0 the band of this chunk - must have landed.  The band DMAs of the NEXT chunk, issued in
      // group ky=0, are younger than the weights needed at ky=1: a counted vmcnt leaves them in flight there.
      if (ky == 1 && cc + 1 < nchunk) {
        static_assert(HQ >= 2 && HQ <= 6, "add the vmcnt immediate for this tile");
        if constexpr (HQ == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else if constexpr (HQ == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else if constexpr (HQ == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if constexpr (HQ == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      STAMP(unsigned long long tb = __builtin_amdgcn_s_memtime(); t_wait += tb - ta;)
      __builtin_amdgcn_sched_barrier(0);
      int ncc = cc, nky = ky + 1;
      if (nky == 3) {
        nky = 0;
        ncc = cc + 1;
      }
      const bool more_w = ncc < nchunk, more_h = (ky == 0) && (cc + 1 < nchunk);
      const f16 *wb = wbuf + buf * C::WBUF_HALFS;
      unsigned vm[(NT + 2) / 3];           // keep the border selects inside the loop (see halo_tile)
#pragma unroll
      for (int t = 0; t < (NT + 2) / 3; ++t) {
        vm[t] = vmp[t];
        asm volatile("" : "+v"(vm[t]));
      }
      // 6 steps (kx, k-step) x NT pixel tiles = 6 NT tile visits, 2 MFMAs each; pixel fragments rotate through BD registers
      // (the fragment of visit t+BD is requested right after the MFMAs of visit t), weight fragments of step st+1 go to the
      // spare set at the start of step st.  A tap's swizzle term is the same for all tiles of a step (tiles are 32 px apart).
      constexpr int NV = 6 * NT, BD = NT >= 4 ? 4 : (NT == 3 ? 3 : 2);
      half8 af[2][2], bf[BD];
      int tapb[3][2];                      // half offset of (tap pixel, channel group ks*2 + lh) of tile 0
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int Pt = pb + ky * W + kx, sw = (Pt >> 2) & 3;
        tapb[kx][0] = Pt * 32 + ((lh ^ sw) * 8);
        tapb[kx][1] = Pt * 32 + (((2 + lh) ^ sw) * 8);
      }
      auto load_a = [&](int st, int set) __attribute__((always_inline)) {
        const int kx = st >> 1, ks = st & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[set][i] = *reinterpret_cast<const half8 *>(&wb[wa[ks] + i * (32 * 32) + kx * (HL_BM * HL_CK)]);
      };
      auto load_b = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, st = t / NT, j = t % NT, kx = st >> 1, ks = st & 1;
        constexpr int imm = j * 32 * 32;
        const bool ok = (vm[j / 3] >> ((j % 3) * 9 + ky * 3 + kx)) & 1u;
        const int base = ok ? tapb[kx][ks] : (C::ZERO_OFF - (cc & 1) * C::HBUF_HALFS - imm);
        bf[t % BD] = *reinterpret_cast<const half8 *>(&halo[base + imm]);
      };
      load_a(0, 0);
      load_b(IC<0>{});
      load_b(IC<1>{});
      if constexpr (BD > 2) load_b(IC<2>{});
      if constexpr (BD > 3) load_b(IC<3>{});
      __builtin_amdgcn_sched_barrier(0);
      auto visit = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, st = t / NT, j = t % NT, cur = st & 1;
        if constexpr (j == 0 && st + 1 < 6) {
          load_a(st + 1, cur ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][0], bf[t % BD], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][1], bf[t % BD], acc[1][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + BD < NV) load_b(IC<t + BD>{});
        // weight DMAs of the next group in the first visits, the next chunk's band DMAs right after them
        // (program order "weights, then band": the counted vmcnt at ky=1 relies on it)
        static_assert(WQ + HQ <= NV, "DMA issue slots");
        if constexpr (t < WQ) {
          if (more_w) wstage(ncc, nky, buf ^ 1, IC<t>{}, IC<t + 1>{});
        } else if constexpr (t - WQ < HQ) {
          if (more_h) halo_dma(cc + 1, (cc + 1) & 1, IC<t - WQ>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#define V4(a) visit(IC<(a) < NV ? (a) : NV - 1>{}); if constexpr ((a) + 1 < NV) visit(IC<(a) + 1 < NV ? (a) + 1 : NV - 1>{}); if constexpr ((a) + 2 < NV) visit(IC<(a) + 2 < NV ? (a) + 2 : NV - 1>{}); if constexpr ((a) + 3 < NV) visit(IC<(a) + 3 < NV ? (a) + 3 : NV - 1>{});
      V4(0) if constexpr (NV > 4) { V4(4) } if constexpr (NV > 8) { V4(8) } if constexpr (NV > 12) { V4(12) } if constexpr (NV > 16) { V4(16) } if constexpr (NV > 20) { V4(20) }
#undef V4
      static_assert(NV <= 24, "visit list covers NT = 1 .. 4");
      STAMP(t_body += __builtin_amdgcn_s_memtime() - tb;)
    }
  }
  STAMP(const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();)

  if constexpr (SPLIT) {
    // fp32 partial sums of this share -> p.splitk[ks][m][Cout] (16-byte stores in the accumulator layout); splitk_finish adds
    // the shares in a fixed order and applies residual / ReLU / positional embedding
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int m = m0 + wn * PXW + j * 32 + lr;
      if (m >= p.M) continue;
      float *o = p.splitk + ((size_t)ks * p.M + m) * p.Cout + c0 + wm * 64 + lh * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *reinterpret_cast<float4 *>(o + i * 32 + rg * 8) = make_float4(acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]);
    }
    return;
  }
  halo_epilogue<NT, RES, POST>(p, m0, c0, lds, acc);
  STAMP(if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long *o = g_halo_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = t_wait; o[1] = t_body; o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32; o[3] = t_loop_end - t_prev; o[4] = t_prev - t_entry;
    o[5] = __builtin_amdgcn_s_memtime() - t_loop_end; o[6] = r_entry; o[7] = __builtin_amdgcn_s_memrealtime();
  })
}

#ifdef HALO_STAMP      // diagnostic builds only: the persistent-workgroup form (FP_HALO_PERSIST=1)
#define HALO_DIAG_SECTION 2
#include "diag/conv_halo_diag.inc"
#undef HALO_DIAG_SECTION
#endif

// Grid = n_main workgroups of 512 px x 128 co (whole rounds), then the rest of the pixels as tiles of nt_tail x 128 px (halo_plan):
// what is less than a round of 512-pixel tiles is cut so that it fills the chip once.  A pixel's arithmetic does not depend on its tile.
template <int W, bool RES, bool POST>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_dma_kernel(ConvArgs p, int n_main, int tile0, int nt_tail) {
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  constexpr int TMM = 512;
  const int n_ct = p.Cout / HL_BM;
  // tile0: logical tiles before it were done by the persistent launch (0 when this launch is the whole layer)
  if ((int)blockIdx.x < n_main) {
    const int L = tile0 + xcd_remap(blockIdx.x, n_main);
    halo_tile_dma<W, 4, RES, POST>(p, (L / n_ct) * TMM, (L % n_ct) * HL_BM, lds);
  } else {
    const int t = xcd_remap(blockIdx.x - n_main, gridDim.x - n_main);
    const int m0 = ((tile0 + n_main) / n_ct) * TMM + (t / n_ct) * (nt_tail * 128), c0 = (t % n_ct) * HL_BM;
    if (m0 >= p.M) return;
    if (nt_tail == 1) halo_tile_dma<W, 1, RES, POST>(p, m0, c0, lds);
    else if (nt_tail == 2) halo_tile_dma<W, 2, RES, POST>(p, m0, c0, lds);
    else if (nt_tail == 3) halo_tile_dma<W, 3, RES, POST>(p, m0, c0, lds);
    else halo_tile_dma<W, 4, RES, POST>(p, m0, c0, lds);
  }
}


// Split-K form for launches of a few workgroups (3 .. 4 hypotheses; one and two run conv_small.hip since round 5, FP_SMALL=0: this form): every tile is a 128-pixel quarter tile, p.ksplit workgroups
// per tile each take an equal share of the input-channel chunks.  At one hypothesis a 512-channel layer is 16 workgroups that
// each walk 48 weight groups behind one DMA round trip apiece (40 us); four shares of 12 groups + the finishing pass take a third,
// eight shares of 6 (1 and 2 hypotheses: the grid still fits the chip) another 2 % of a one-hypothesis step.
template <int W>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_splitk_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  const int n_ct = p.Cout / HL_BM;
  const int ks = blockIdx.x % p.ksplit, t = blockIdx.x / p.ksplit;
  const int m0 = (t / n_ct) * 128;
  if (m0 >= p.M) return;
  halo_tile_dma<W, 1, false, false, true>(p, m0, (t % n_ct) * HL_BM, lds, ks);
}

// out = [+ pos.emb.] relu?( sum_ks partial[ks] [+ residual] ), one thread per pixel x 4 channels, shares added in order
__global__ __launch_bounds__(256) void splitk_finish_kernel(ConvArgs p) {
  const int q = blockIdx.x * 256 + threadIdx.x, c4 = p.Cout / 4;
  if (q >= p.M * c4) return;
  const int m = q / c4, c = (q - m * c4) * 4;
  float4 v = *reinterpret_cast<const float4 *>(p.splitk + (size_t)m * p.Cout + c);
  for (int ks = 1; ks < p.ksplit; ++ks) {
    const float4 w = *reinterpret_cast<const float4 *>(p.splitk + ((size_t)ks * p.M + m) * p.Cout + c);
    v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
  }
  if (p.res) {
    const half4 r = *reinterpret_cast<const half4 *>(p.res + (size_t)m * p.Cout + c);
    v.x += (float)r[0], v.y += (float)r[1], v.z += (float)r[2], v.w += (float)r[3];
  }
  const float lo = p.relu ? 0.f : -__builtin_inff();
  v.x = fmaxf(v.x, lo), v.y = fmaxf(v.y, lo), v.z = fmaxf(v.z, lo), v.w = fmaxf(v.w, lo);
  if (p.post_add) {
    const float4 pv = *reinterpret_cast<const float4 *>(p.post_add + (size_t)(m % p.post_period) * p.Cout + c);
    v.x += pv.x, v.y += pv.y, v.z += pv.z, v.w += pv.w;
  }
  half4 hv;
  hv[0] = (f16)v.x, hv[1] = (f16)v.y, hv[2] = (f16)v.z, hv[3] = (f16)v.w;
  const bool hi = m >= p.split_m;
  const long long orow = hi ? (long long)(m - p.split_m) : (long long)m;
  *reinterpret_cast<half4 *>((f16 *)p.out + orow * p.out_ld + (hi ? p.coff_hi : 0) + c) = hv;
}


bool conv_halo_supported(const ConvArgs &a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.H == a.W && (a.W == 40 || a.W == 20) && a.Cin % HL_CK == 0 &&
         a.Cout % HL_BM == 0 && a.out_mode == 0 && a.Kpad == 9 * a.Cin && a.out_ld % 8 == 0 && a.coff_hi % 8 == 0;
}

int g_halo_tail = 1;   // FP_HALO_TAIL=0 disables the tail split (A/B timing only; results are identical)
int g_halo_form = 0;   // diagnostic builds only, FP_HALO_FORM: 0 = band by LDS-DMA, 8 waves (default); 1 = band through registers, FP_HALO_NPW x 2 waves
                       // (A/B timing; identical results: same MFMA shape and accumulation order)
int g_halo_npw = 4;    // diagnostic builds: FP_HALO_NPW=2 selects the 4-wave / 2-workgroups-per-CU form (A/B timing only; results are identical)

// main/tail split: whole rounds of `slots` main tiles stay; the remainder is cut in four when that shortens the last round
// (a quarter tile costs ~0.35 of a main tile: less operand reuse), i.e. when the remainder fills < ~70 % of a round.
void halo_split(int n_tiles, int slots, int *n_main, int *n_tail4) {
  const int rem = n_tiles % slots;
  *n_main = n_tiles;
  *n_tail4 = 0;
  if (!g_halo_tail || rem == 0) return;
  if (n_tiles < slots) {
    // less than one round (tracking; a rank's shard of an 8-way job: 32 hypotheses = 100 tiles at C = 512): the launch lasts as
    // long as ONE workgroup, so quarter tiles (4 x the workgroups, ~0.35 x the time each) win as long as they fit two rounds
    if (4 * n_tiles <= 2 * slots) {
      *n_main = 0;
      *n_tail4 = 4 * n_tiles;
    }
    return;
  }
  const double cost_whole = 1.0, cost_quarter = 0.35 * ((4 * rem + slots - 1) / slots);
  if (cost_quarter < cost_whole) {
    *n_main = n_tiles - rem;
    *n_tail4 = 4 * rem;
  }
}

// Tiling of a launch (M pixels, n_ct cout tiles of 128, `slots` CUs, one workgroup each): whole rounds of 512-pixel tiles, and what is
// left - less than a round, or the whole of a small launch (a rank's shard of an 8-way job, a tracking frame) - as tiles of nt x 128
// pixels, nt chosen for the shortest finish.  Relative cost of a tile by its nt (measured on one-round launches at C = 512: a round
// of 128-pixel tiles lasts 0.43 of a round of 512-pixel ones - less operand reuse, the same prologue and epilogue).
static const double kHaloTileCost[5] = {0.0, 0.43, 0.62, 0.81, 1.0};
void halo_plan(int M, int n_ct, int slots, int *n_main, int *nt_tail, int *n_tail) {
  const int qm = (M + 127) / 128;                       // 128-pixel quarters along the pixel axis
  const int per_round = (slots / n_ct) * 4;             // quarters a round of 512-pixel tiles covers
  int full = per_round > 0 ? qm / per_round : 0;
  int rem = qm - full * per_round;
  *n_main = full * (slots / n_ct) * n_ct;
  *nt_tail = 4;
  *n_tail = 0;
  if (rem == 0) return;
  if (!g_halo_tail) {                                   // FP_HALO_TAIL=0: 512-pixel tiles only
    *n_tail = ((rem + 3) / 4) * n_ct;
    return;
  }
  static const int force_nt = getenv("FP_HALO_NT") ? atoi(getenv("FP_HALO_NT")) : 0;      // experiments: tiles of this many x 128 pixels for whatever is not a whole round
  if (force_nt >= 1 && force_nt <= 4) {
    *nt_tail = force_nt;
    *n_tail = ((rem + force_nt - 1) / force_nt) * n_ct;
    return;
  }
  double best = 1e30;
  for (int nt = 4; nt >= 1; --nt) {
    const int wgs = ((rem + nt - 1) / nt) * n_ct;
    const double cost = ((wgs + slots - 1) / slots) * kHaloTileCost[nt];
    if (cost < best - 1e-9) {
      best = cost;
      *nt_tail = nt;
      *n_tail = wgs;
    }
  }
}

#ifdef HALO_STAMP      // diagnostic builds only: entry points and launchers of the forms above
#define HALO_DIAG_SECTION 3
#include "diag/conv_halo_diag.inc"
#undef HALO_DIAG_SECTION
#endif

template <int W>
static void halo_lds_all(std::vector<KernelLds> &v) {
  using C = HaloCfgD<W, 512>;
  v.push_back({(const void *)conv3x3_halo_dma_kernel<W, false, false>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_halo_dma_kernel<W, true, false>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_halo_dma_kernel<W, false, true>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_halo_dma_kernel<W, true, true>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_halo_splitk_kernel<W>, HaloCfgD<W, 128>::LDS_BYTES});
}
void conv_halo_kernel_lds(std::vector<KernelLds> &v) { halo_lds_all<40>(v), halo_lds_all<20>(v); }

template <int W, bool RES, bool POST>
static int launch_halo_dma(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  using C = HaloCfgD<W, 512>;
  static_assert(HaloCfgD<W, 128>::LDS_BYTES <= C::LDS_BYTES, "tail tiles fit the main tile's LDS");
  const int slots = ctx->num_cu;
  int n_main, nt_tail, n_tail;
  halo_plan(a.M, a.Cout / HL_BM, slots, &n_main, &nt_tail, &n_tail);
  int tile0 = 0;
#ifdef HALO_STAMP
  FP_TRY((diag_launch_persist<W, RES, POST>(a, slots, C::LDS_BYTES, &n_main, &tile0, s)));
#endif
  if (n_main + n_tail > 0)
    hipLaunchKernelGGL((conv3x3_halo_dma_kernel<W, RES, POST>), dim3(n_main + n_tail), dim3(512), C::LDS_BYTES, s, a, n_main, tile0, nt_tail);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

template <int W>
static int launch_halo_dma_flags(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  if (a.post_add) return a.res ? launch_halo_dma<W, true, true>(ctx, a, s) : launch_halo_dma<W, false, true>(ctx, a, s);
  return a.res ? launch_halo_dma<W, true, false>(ctx, a, s) : launch_halo_dma<W, false, false>(ctx, a, s);
}

int launch_splitk_finish(const ConvArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((a.M * (a.Cout / 4) + 255) / 256), dim3(256), 0, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

template <int W>
static int launch_halo_splitk(const ConvArgs &a, hipStream_t s) {
  using C = HaloCfgD<W, 128>;
  const int n_q = ((a.M + 127) / 128) * (a.Cout / HL_BM);
  hipLaunchKernelGGL((conv3x3_halo_splitk_kernel<W>), dim3(n_q * a.ksplit), dim3(512), C::LDS_BYTES, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return launch_splitk_finish(a, s);
}

// Split factor for a launch of this shape: 0 unless the quarter tiles fill at most a quarter of the CUs (1 .. 4 hypotheses);
// then, from 8 chunks (256 channels) on, as many shares (at most 8; FP_KSPLIT_MAX) as keep >= 2 chunks per share and the grid within the chip.  Decided by the caller
// that owns the scratch (its size: shares x M x Cout floats <= 4 x hypotheses x 1600 x 256 for every layer of the networks).
int conv_halo_ksplit(const ConvArgs &a, int num_cu) {
  if (!conv_halo_supported(a)) return 0;
  const int n_q = ((a.M + 127) / 128) * (a.Cout / HL_BM), nchunk = a.Cin / HL_CK;
  if (n_q * 4 > num_cu) return 0;
  static const int k_max = getenv("FP_KSPLIT_MAX") ? atoi(getenv("FP_KSPLIT_MAX")) : 8;
  // from 256 input channels on: with the four chunks of a 128-channel layer two shares + the finishing launch take longer than the one
  // launch (one-hypothesis step 2.17 -> 2.12 ms, four hypotheses 2.74 -> 2.56 without them; FP_KSPLIT_MIN_CHUNKS=0: split those too)
  static const int min_chunks = getenv("FP_KSPLIT_MIN_CHUNKS") ? atoi(getenv("FP_KSPLIT_MIN_CHUNKS")) : 8;
  if (nchunk < min_chunks) return 0;
  int k = k_max;
  while (k > 1 && (nchunk % k != 0 || nchunk / k < 2 || n_q * k > num_cu)) k >>= 1;
  return k > 1 ? k : 0;
}

int launch_conv_halo(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  // lane offsets into the input tensor are 32-bit byte offsets from its base
  FP_REQUIRE((double)a.M * a.Cin * 2.0 < 4294967296.0, "conv3x3: input tensor of %.1f GB exceeds the 4 GB the kernel addresses", (double)a.M * a.Cin * 2e-9);
  if (a.splitk && a.ksplit > 1) {
    FP_REQUIRE((a.Cin / HL_CK) % a.ksplit == 0 && a.out_ld % 4 == 0 && a.coff_hi % 4 == 0, "conv3x3 split-K: %d shares do not divide %d chunks", a.ksplit, a.Cin / HL_CK);
    return a.W == 40 ? launch_halo_splitk<40>(a, s) : launch_halo_splitk<20>(a, s);
  }
#ifdef HALO_STAMP
  if (g_halo_form != 0) {
    if (g_halo_npw == 2) return a.W == 40 ? launch_halo_flags<40, 2>(a, s) : launch_halo_flags<20, 2>(a, s);
    return a.W == 40 ? launch_halo_flags<40, 4>(a, s) : launch_halo_flags<20, 4>(a, s);
  }
#endif
  return a.W == 40 ? launch_halo_dma_flags<40>(ctx, a, s) : launch_halo_dma_flags<20>(ctx, a, s);
}

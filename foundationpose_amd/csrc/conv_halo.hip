// 3x3 / stride-1 / pad-1 convolution for gfx950 with the input HALO BAND resident in LDS.
//
// 93 % of the network FLOPs are 3x3 stride-1 convolutions on 40x40 (C=128|256) and 20x20 (C=512) maps
// (learning/models/network_modules.py:73-111 via refine_network.py:37-50).  An im2col-style implicit
// GEMM re-fetches every activation 9 times (once per tap) from L2 into LDS; at a 128x128 tile that is
// 64 FLOP per LDS-fill byte, i.e. >30 TB/s of L2->LDS traffic at the MFMA peak.  Here a workgroup owns
// 256 consecutive output pixels (flattened over images: 6.4 rows of a 40-wide map, 12.8 rows of a
// 20-wide one) x 128 output channels and, per 32-channel input chunk,
//   * loads the input rows it needs ONCE into LDS ("halo band": <= 10 x 40 or 16 x 20 pixels x 32 ch,
//     24-26 KB, global -> registers (prefetched one chunk ahead under the MFMAs) -> ds_write_b128),
//   * streams the 3 taps of one kernel row of weights at a time (128 co x 32 ci x 3 = 24 KB, double
//     buffered) with LDS-DMA (global_load_lds_dwordx4: no VGPRs, in flight across the MFMA block),
//   * and feeds v_mfma_f32_32x32x16_f16 for all 9 taps from that one band: the tap shift is just a
//     different LDS address per lane (+-1 pixel, +-1 row; image borders select a zero chunk).
// => 190 FLOP per LDS-fill byte, 48 MFMAs per wave between barriers, 2 workgroups per CU (81.3 KB LDS,
// 4 waves each, each wave 64 co x 128 px = 2x4 accumulator tiles).  Halo pixels are padded to 80 B and the
// weight image is XOR-swizzled, so both ds_read_b128 fragment reads are bank-conflict free.
// Epilogue: bias (+ residual staged through LDS) + ReLU (+ positional embedding) in fp32, one rounding
// to fp16, LDS transpose, 16-byte row-contiguous NHWC stores.
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16-byte register value (plain vector loads / stores in IR)

#define HL_TM 256   // pixels per workgroup of the main tiles (tail tiles: 64, see conv3x3_halo_kernel)
#define HL_BM 128
#define HL_CK 32
#define HL_SLD 136  // halfs per staged output row (128 + 8 pad) -> 272 B
#define HL_PS 40    // halfs per halo pixel (32 channels + 8 pad = 80 B): conflict-free ds_read_b128 AND every tap
                    // shift / k-step is a compile-time immediate offset from ONE base register per pixel tile

template <int V>
struct IC {
  static constexpr int value = V;
};

template <int W, int TM = HL_TM>
struct HaloCfg {
  static constexpr int MAXSLOT = (W - 1 + TM - 1) / W + 1 + 2;     // input rows a TM-pixel run can touch (+1 above, +1 below)
  static constexpr int HALO_CHUNKS = MAXSLOT * W * 4;              // 16-byte chunks (4 per pixel at 32 channels)
  static constexpr int HALO_HALFS = (MAXSLOT * W + 2) * HL_PS + 8; // pixel p lives at index p+1; + one zero chunk
  static constexpr int ZERO_OFF = (MAXSLOT * W + 2) * HL_PS;       // half offset of the zero chunk
  static constexpr int halo_loads(int nth) { return (HALO_CHUNKS + nth - 1) / nth; }
  static constexpr int WBUF_HALFS = 3 * HL_BM * HL_CK;             // one kernel row of taps
  static constexpr int LDS_HALFS_MAIN = HALO_HALFS + 2 * WBUF_HALFS;
  static constexpr int LDS_HALFS_EPI = TM * HL_SLD;
  static constexpr int LDS_BYTES = 2 * (LDS_HALFS_MAIN > LDS_HALFS_EPI ? LDS_HALFS_MAIN : LDS_HALFS_EPI);
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

// One workgroup tile: 4 waves as 2 (cout halves) x 2 (pixel halves); each wave 64 co x 32*NT px (2 x NT accumulator tiles).
// NT = 4 -> 256-pixel tile (the main tiles), NT = 1 -> 64-pixel tile (tail tiles).  The accumulation order of every output
// element (chunk, ky, kx, k-step) does not depend on NT, so the tile shape never changes a result bit.
template <int W, int NT>
__device__ __forceinline__ void halo_tile(const ConvArgs &p, const int m0, const int c0, f16 *lds) {
  constexpr int TM = 64 * NT;
  using C = HaloCfg<W, TM>;
  constexpr int H = W;
  constexpr int NWN = 2;
  constexpr int NTH = 256;              // threads
  constexpr int PXW = 32 * NT;          // pixels per wave
  constexpr int HLOADS = C::halo_loads(NTH);
  constexpr int WQ = 24 / (2 * NWN);    // weight DMA instructions per wave per group
  f16 *halo = lds;
  f16 *wbuf = lds + C::HALO_HALFS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int lr = lane & 31, lh = lane >> 5;
  const int GR0 = m0 / W - 1;                       // global input row (n*H + iy) held by slot 0
  const int mlast = min(m0 + TM - 1, p.M - 1);
  const int NS = mlast / W + 1 - GR0 + 1;           // slots in use
  const int total_rows = p.Nimg * H;
  const int nchunk = p.Cin / HL_CK;

  // ---- per-lane B-fragment bases: pixel (slot(ky), ox) for the 4 pixel tiles of this wave ----
  int pb[NT];         // LDS half-offset of the top-left tap; tap (ky,kx), k-step ks add the immediate ((ky*W+kx)*40 + ks*16)
  unsigned vmask[NT]; // bit (ky*3+kx): tap inside the image
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int m = m0 + wn * PXW + j * 32 + lr;
    m = min(m, p.M - 1);
    const int gr = m / W, ox = m - gr * W, oy = gr % H;
    pb[j] = ((gr - GR0) * W + ox - W) * HL_PS + lh * 8;   // top-left tap (ky=0,kx=0) of this pixel, k-half lh
    unsigned vm = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) vm |= 1u << (ky * 3 + kx);
      }
    vmask[j] = vm;
  }

  // A-fragment bases (weights image is XOR-swizzled: chunk ^ ((co>>2)&3), identical for co and co+32)
  int wa[2];
  {
    const int co = wm * 64 + lr;
    wa[0] = co * 32 + ((lh ^ ((co >> 2) & 3)) * 8);
    wa[1] = co * 32 + (((2 + lh) ^ ((co >> 2) & 3)) * 8);
  }

  // ---- halo staging: global -> registers (prefetch) -> LDS ----
  // Load i of thread tid fetches 16-byte chunk (tid + 256 i): band pixel tid/4 + 64 i, channel group tid%4.  Rows outside
  // the tensor (above the first / below the last image, or past the slots in use) are CLAMPED to a valid pixel instead of
  // skipped: what lands there is never read (those taps select the zero chunk through vmask), and every wave then
  // issues exactly HLOADS loads - the counted vmcnt at ky=1 depends on that.  Offsets are 32-bit from the tensor base.
  u32x4 hreg[HLOADS];
  const int hpix0 = GR0 * W + (tid >> 2), hpix_max = total_rows * W - 1;
  auto halo_load = [&](int cc, auto i0c, auto i1c) __attribute__((always_inline)) {     // loads I0 .. I1-1 of chunk cc (compile-time range: hreg stays in registers)
    constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value < HLOADS ? decltype(i1c)::value : HLOADS;
    int px0 = hpix0;
    asm volatile("" : "+v"(px0));     // recompute the 32-bit offsets per chunk (hoisted, hipcc keeps 7 register pairs alive)
    auto one = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i >= I0 && i < I1) {
        const int px = min(max(px0 + 64 * i, 0), hpix_max);
        hreg[i] = *reinterpret_cast<const u32x4 *>(p.in + (unsigned)(px * p.Cin + cc * HL_CK + (tid & 3) * 8));
      }
    };
    one(IC<0>{}), one(IC<1>{}), one(IC<2>{}), one(IC<3>{}), one(IC<4>{}), one(IC<5>{}), one(IC<6>{});
    static_assert(HLOADS <= 7, "extend the list");
  };
  auto halo_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
      const int idx = tid + NTH * i;
      const int pix = idx >> 2, ch = idx & 3;
      if (idx < C::HALO_CHUNKS) *reinterpret_cast<u32x4 *>(&halo[(pix + 1) * HL_PS + ch * 8]) = hreg[i];
    }
  };
  // ---- weights: one kernel row (3 taps) per group, LDS-DMA, lane-linear image with the swizzle on the SOURCE ----
  // instruction q of wave w covers tap kx = q/2, couts ((q&1)*4 + w)*16 + lane/4, 16-byte channel group lane%4 (swizzled):
  // the lane part of the source address is the same for every q, group and chunk
  const unsigned woff = (unsigned)(((wave * 16 + (lane >> 2)) * p.Kpad + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2);
  auto wstage = [&](int cc, int ky, int buf, auto q0c, auto q1c) __attribute__((always_inline)) {     // DMA instructions Q0 .. Q1-1 of group (cc, ky)
    constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value;
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
      const f16 *sb = p.w + (size_t)(c0 + (q & 1) * 64) * p.Kpad + (ky * 3 + (q >> 1)) * p.Cin + cc * HL_CK;
      glds16(sb, woff, wbuf + buf * C::WBUF_HALFS + (q * 2 * NWN + wave) * 512);
    }
  };

  floatx16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (tid < 1) *reinterpret_cast<u32x4 *>(&halo[C::ZERO_OFF]) = u32x4{0, 0, 0, 0};
  halo_load(0, IC<0>{}, IC<HLOADS>{});
  wstage(0, 0, 0, IC<0>{}, IC<WQ>{});
  int g = 0;
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();          // every wave is done reading the previous chunk's halo
    halo_store();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky, ++g) {
      const int buf = g & 1;
      // weights(g) must have landed and the halo stores must be visible.  The halo prefetch loads issued in
      // group ky=0 are YOUNGER than the weights needed at ky=1, so a counted vmcnt leaves them in flight there
      // (a __syncthreads() would drain them one group after issue).
      if (ky == 1 && cc + 1 < nchunk) {
        static_assert(HLOADS == 7 || HLOADS == 5 || HLOADS == 4 || HLOADS == 3, "add the vmcnt immediate for this tile");
        if constexpr (HLOADS == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
        else if constexpr (HLOADS == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else if constexpr (HLOADS == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // The next group's weights (into the other buffer) and, in group ky=0, the next chunk's halo (into registers) are
      // requested from INSIDE the MFMA steps below - two DMAs after the first MFMA pair of steps 0..2, the halo loads in
      // steps 3..5 - so their issue cycles (~60 per DMA) sit under running MFMAs instead of in front of the group.
      // Program order stays "DMAs, then halo loads": the counted vmcnt at ky=1 relies on it.
      int ncc = cc, nky = ky + 1;
      if (nky == 3) {
        nky = 0;
        ncc = cc + 1;
      }
      const bool more_w = ncc < nchunk, more_h = (ky == 0) && (cc + 1 < nchunk);
      const f16 *wb = wbuf + buf * C::WBUF_HALFS;
      // keep the per-tap border selects INSIDE the loop: hoisted, their 72 results would not fit the register file
      unsigned vm[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        vm[j] = vmask[j];
        asm volatile("" : "+v"(vm[j]));
      }
      // software pipeline over the 6 (kx, k-step) steps of this kernel row, in an order that needs only one spare
      // weight-fragment pair: MFMAs go pixel-tile-major, so a pixel fragment is dead after two MFMAs and its register is
      // reloaded for the NEXT step right away - every fragment is requested >= 6 MFMAs (192 cycles) before its first use.
      // sched_barrier(0) pins that order; the waits the compiler inserts are then counted lgkmcnt(n), not lgkmcnt(0).
      half8 af[2][2], bf[NT];
      auto load_a = [&](int st, int set) __attribute__((always_inline)) {
        const int kx = st >> 1, ks = st & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[set][i] = *reinterpret_cast<const half8 *>(&wb[wa[ks] + i * (32 * 32) + kx * (HL_BM * HL_CK)]);
      };
      auto load_b = [&](int st, int j) __attribute__((always_inline)) {
        const int kx = st >> 1, ks = st & 1;
        const int imm = (ky * W + kx) * HL_PS + ks * 16;
        const bool ok = (vm[j] >> (ky * 3 + kx)) & 1u;
        const int base = ok ? pb[j] : (C::ZERO_OFF - imm);
        bf[j] = *reinterpret_cast<const half8 *>(&halo[base + imm]);
      };
      load_a(0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) load_b(0, j);
      __builtin_amdgcn_sched_barrier(0);
      auto step = [&](auto stc) __attribute__((always_inline)) {
        constexpr int st = decltype(stc)::value, cur = st & 1;
        if (st + 1 < 6) load_a(st + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][0], bf[j], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][1], bf[j], acc[1][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < 6) load_b(st + 1, j);
          if (j == 0) {
            if constexpr (st < 3) {
              if (more_w) wstage(ncc, nky, buf ^ 1, IC<2 * st>{}, IC<2 * st + 2>{});
            } else {
              constexpr int HP = (HLOADS + 2) / 3;
              if (more_h) halo_load(cc + 1, IC<(st - 3) * HP>{}, IC<(st - 3) * HP + HP>{});
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      step(IC<0>{});
      step(IC<1>{});
      step(IC<2>{});
      step(IC<3>{});
      step(IC<4>{});
      step(IC<5>{});
    }
  }

  // ---------------- epilogue ----------------
  __syncthreads();
  f16 *stage = lds;   // [256 px][HL_SLD]
  if (p.res) {
    // all loads of a batch are issued before the first ds_write (a load->store loop serialises their latencies)
    constexpr int NRES = TM * 16 / NTH, RB = NRES < 8 ? NRES : 8;
#pragma unroll
    for (int i0 = 0; i0 < NRES; i0 += RB) {
      u32x4 rv[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
        const int m = min(m0 + px, p.M - 1);      // unconditional (clamped) load: a guarded one makes hipcc wait per element
        rv[u] = *reinterpret_cast<const u32x4 *>(p.res + (size_t)m * p.Cout + c0 + c16 * 8);
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
        *reinterpret_cast<u32x4 *>(&stage[px * HL_SLD + c16 * 8]) = rv[u];
      }
    }
    __syncthreads();
  }
  // bias (and, where used, positional-embedding) values are fetched in batches BEFORE they are consumed: loaded
  // one by one inside the loop hipcc waits vmcnt(0) after every load (32 serial L2 round trips per lane)
  float4 bvs[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) bvs[i][rg] = *reinterpret_cast<const float4 *>(p.bias + c0 + wm * 64 + i * 32 + rg * 8 + lh * 4);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int pxl = wn * PXW + j * 32 + lr;
    const int m = m0 + pxl;
    float4 pvs[2][4];
    if (p.post_add) {
      const int prow = min(m, p.M - 1) % p.post_period;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          pvs[i][rg] = *reinterpret_cast<const float4 *>(p.post_add + (size_t)prow * p.Cout + c0 + wm * 64 + i * 32 + rg * 8 + lh * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int col = wm * 64 + i * 32 + rg * 8 + lh * 4;   // channel within the 128-wide tile
        const float4 bv = bvs[i][rg];
        float v[4] = {acc[i][j][rg * 4 + 0] + bv.x, acc[i][j][rg * 4 + 1] + bv.y, acc[i][j][rg * 4 + 2] + bv.z,
                      acc[i][j][rg * 4 + 3] + bv.w};
        f16 *sp = &stage[pxl * HL_SLD + col];
        if (p.res) {
          half4 rv = *reinterpret_cast<const half4 *>(sp);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.post_add) {
          const float4 pv = pvs[i][rg];
          v[0] += pv.x;
          v[1] += pv.y;
          v[2] += pv.z;
          v[3] += pv.w;
        }
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (f16)v[e];
        *reinterpret_cast<half4 *>(sp) = hv;
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
      const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
      ov[u] = *reinterpret_cast<const u32x4 *>(&stage[px * HL_SLD + c16 * 8]);
    }
#pragma unroll
    for (int u = 0; u < OB; ++u) {
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

// Grid = n_main workgroups of 256 px x 128 co, then n_tail4 workgroups of 64 px x 128 co covering the LAST main-size tiles
// cut in four.  With two workgroups resident per CU a launch has 512 slots; 3150 equal tiles (N=252: every layer of the
// network) would leave 84 % of the chip idle for the whole seventh round - cutting only the remainder into quarters lets
// that round end after a quarter of the time.  (Cutting every tile would cost the big tile's operand reuse.)
template <int W>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(ConvArgs p, int n_main) {
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  const int n_ct = p.Cout / HL_BM;
  if ((int)blockIdx.x < n_main) {
    const int L = xcd_remap(blockIdx.x, n_main);   // consecutive L = cout tiles of one pixel tile, then the next pixel tile
    halo_tile<W, 4>(p, (L / n_ct) * HL_TM, (L % n_ct) * HL_BM, lds);
  } else {
    const int t = xcd_remap(blockIdx.x - n_main, gridDim.x - n_main);
    const int L = n_main + (t >> 2);
    const int m0 = (L / n_ct) * HL_TM + (t & 3) * 64;
    if (m0 >= p.M) return;
    halo_tile<W, 1>(p, m0, (L % n_ct) * HL_BM, lds);
  }
}

bool conv_halo_supported(const ConvArgs &a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.H == a.W && (a.W == 40 || a.W == 20) && a.Cin % HL_CK == 0 &&
         a.Cout % HL_BM == 0 && a.out_mode == 0 && a.Kpad == 9 * a.Cin && a.out_ld % 8 == 0 && a.coff_hi % 8 == 0;
}

int g_halo_tail = 1;   // FP_HALO_TAIL=0 disables the tail split (A/B timing only; results are identical)

// main/tail split: whole rounds of `slots` main tiles stay; the remainder is cut in four when that shortens the last round
// (a quarter tile costs ~0.35 of a main tile: less operand reuse), i.e. when the remainder fills < ~70 % of a round.
static void halo_split(int n_tiles, int slots, int *n_main, int *n_tail4) {
  const int rem = n_tiles % slots;
  *n_main = n_tiles;
  *n_tail4 = 0;
  if (!g_halo_tail || rem == 0 || n_tiles < slots) return;
  const double cost_whole = 1.0, cost_quarter = 0.35 * ((4 * rem + slots - 1) / slots);
  if (cost_quarter < cost_whole) {
    *n_main = n_tiles - rem;
    *n_tail4 = 4 * rem;
  }
}

template <int W>
static int launch_halo_w(const ConvArgs &a, hipStream_t s) {
  using C = HaloCfg<W>;
  static bool attr_set = false;
  static int slots = 512;
  if (!attr_set) {
    FP_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_halo_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    int dev = 0, cus = 256;
    FP_CHECK_HIP(hipGetDevice(&dev));
    FP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    slots = 2 * cus;
    attr_set = true;
  }
  const int n_tiles = ((a.M + HL_TM - 1) / HL_TM) * (a.Cout / HL_BM);
  int n_main, n_tail4;
  halo_split(n_tiles, slots, &n_main, &n_tail4);
  hipLaunchKernelGGL((conv3x3_halo_kernel<W>), dim3(n_main + n_tail4), dim3(256), C::LDS_BYTES, s, a, n_main);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int launch_conv_halo(const ConvArgs &a, hipStream_t s) {
  if (a.W == 40) return launch_halo_w<40>(a, s);
  return launch_halo_w<20>(a, s);
}

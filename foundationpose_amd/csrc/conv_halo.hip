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

#define HL_TM 256
#define HL_BM 128
#define HL_CK 32
#define HL_SLD 136  // halfs per staged output row (128 + 8 pad) -> 272 B
#define HL_PS 40    // halfs per halo pixel (32 channels + 8 pad = 80 B): conflict-free ds_read_b128 AND every tap
                    // shift / k-step is a compile-time immediate offset from ONE base register per pixel tile

template <int W>
struct HaloCfg {
  static constexpr int MAXSLOT = (W - 1 + HL_TM - 1) / W + 1 + 2;  // input rows a 256-pixel run can touch (+1 above, +1 below)
  static constexpr int HALO_CHUNKS = MAXSLOT * W * 4;              // 16-byte chunks (4 per pixel at 32 channels)
  static constexpr int HALO_HALFS = (MAXSLOT * W + 2) * HL_PS + 8; // pixel p lives at index p+1; + one zero chunk
  static constexpr int ZERO_OFF = (MAXSLOT * W + 2) * HL_PS;       // half offset of the zero chunk
  static constexpr int halo_loads(int nth) { return (HALO_CHUNKS + nth - 1) / nth; }
  static constexpr int WBUF_HALFS = 3 * HL_BM * HL_CK;             // one kernel row of taps
  static constexpr int LDS_HALFS_MAIN = HALO_HALFS + 2 * WBUF_HALFS;
  static constexpr int LDS_HALFS_EPI = HL_TM * HL_SLD;
  static constexpr int LDS_BYTES = 2 * (LDS_HALFS_MAIN > LDS_HALFS_EPI ? LDS_HALFS_MAIN : LDS_HALFS_EPI);
};

__device__ __forceinline__ void glds16(const f16 *g, f16 *l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// NWN = wave columns: 2 -> 4 waves, each 64 co x 128 px (2x4 tiles, <=256 VGPRs, 2 waves/SIMD at 2 WG/CU);
//                     4 -> 8 waves, each 64 co x 64 px (2x2 tiles, <=128 VGPRs, 4 waves/SIMD at 2 WG/CU)
template <int W, int NWN>
__global__ __launch_bounds__(128 * NWN, NWN) void conv3x3_halo_kernel(ConvArgs p) {
  using C = HaloCfg<W>;
  constexpr int H = W;
  constexpr int NTH = 128 * NWN;        // threads
  constexpr int NT = 8 / NWN;           // 32-pixel tiles per wave
  constexpr int PXW = 32 * NT;          // pixels per wave
  constexpr int HLOADS = C::halo_loads(NTH);
  constexpr int WQ = 24 / (2 * NWN);    // weight DMA instructions per wave per group
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  f16 *halo = lds;
  f16 *wbuf = lds + C::HALO_HALFS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int lr = lane & 31, lh = lane >> 5;
  const int n_ct = p.Cout / HL_BM;
  const int L = xcd_remap(blockIdx.x, gridDim.x);          // 1-D grid; consecutive L = cout tiles of one pixel tile, then the next pixel tile
  const int m0 = (L / n_ct) * HL_TM, c0 = (L % n_ct) * HL_BM;
  const int GR0 = m0 / W - 1;                       // global input row (n*H + iy) held by slot 0
  const int mlast = min(m0 + HL_TM - 1, p.M - 1);
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
  uint4 hreg[HLOADS];
  auto halo_load = [&](int cc) {
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
      const int idx = tid + NTH * i;
      const int pix = idx >> 2, ch = idx & 3;        // pix = slot*W + px
      const int slot = pix / W;
      const int gr = GR0 + slot;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (slot < NS && gr >= 0 && gr < total_rows)
        v = *reinterpret_cast<const uint4 *>(p.in + ((long long)GR0 * W + pix) * p.Cin + cc * HL_CK + ch * 8);
      hreg[i] = v;
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
      const int idx = tid + NTH * i;
      const int pix = idx >> 2, ch = idx & 3;
      if (idx < C::HALO_CHUNKS) *reinterpret_cast<uint4 *>(&halo[(pix + 1) * HL_PS + ch * 8]) = hreg[i];
    }
  };
  // ---- weights: one kernel row (3 taps) per group, LDS-DMA, lane-linear image with the swizzle on the SOURCE ----
  auto wstage = [&](int cc, int ky, int buf) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int L = (q * 2 * NWN + wave) * 64 + lane;  // linear 16-byte position in the 3x128x4 image
      const int kx = L >> 9, rem = L & 511, co = rem >> 2, chp = rem & 3;
      const int ch = chp ^ ((co >> 2) & 3);
      const f16 *src = p.w + (size_t)(c0 + co) * p.Kpad + (ky * 3 + kx) * p.Cin + cc * HL_CK + ch * 8;
      glds16(src, wbuf + buf * C::WBUF_HALFS + (q * 2 * NWN + wave) * 512);
    }
  };

  floatx16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (tid < 1) *reinterpret_cast<uint4 *>(&halo[C::ZERO_OFF]) = make_uint4(0, 0, 0, 0);
  halo_load(0);
  wstage(0, 0, 0);
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
        if constexpr (HLOADS == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
        else if constexpr (HLOADS == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else if constexpr (HLOADS == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if constexpr (HLOADS == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      {                       // prefetch the next group's weights into the other buffer
        int ncc = cc, nky = ky + 1;
        if (nky == 3) {
          nky = 0;
          ncc = cc + 1;
        }
        if (ncc < nchunk) wstage(ncc, nky, buf ^ 1);
      }
      // next chunk's halo -> registers, issued AFTER this group's barrier so that the barrier's vmcnt(0)
      // drain (hipcc drains every VMEM op before s_barrier while an LDS-DMA is pending) does not expose it
      if (ky == 0 && cc + 1 < nchunk) halo_load(cc + 1);
      const f16 *wb = wbuf + buf * C::WBUF_HALFS;
      // keep the per-tap border selects INSIDE the loop: hoisted, their 72 results would not fit the register file
      unsigned vm[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        vm[j] = vmask[j];
        asm volatile("" : "+v"(vm[j]));
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          half8 af[2], bf[NT];
#pragma unroll
          for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8 *>(&wb[wa[ks] + i * (32 * 32) + kx * (HL_BM * HL_CK)]);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            constexpr int dummy = 0;
            const int imm = (ky * W + kx) * HL_PS + ks * 16;
            const bool ok = (vm[j] >> (ky * 3 + kx)) & 1u;
            const int base = ok ? pb[j] : (C::ZERO_OFF - imm);
            bf[j] = *reinterpret_cast<const half8 *>(&halo[base + imm + dummy]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // ---------------- epilogue ----------------
  __syncthreads();
  f16 *stage = lds;   // [256 px][HL_SLD]
  if (p.res) {
    // all loads of a batch are issued before the first ds_write (a load->store loop serialises their latencies)
    constexpr int NRES = 4096 / NTH, RB = 8;
#pragma unroll
    for (int i0 = 0; i0 < NRES; i0 += RB) {
      uint4 rv[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
        const int m = min(m0 + px, p.M - 1);      // unconditional (clamped) load: a guarded one makes hipcc wait per element
        rv[u] = *reinterpret_cast<const uint4 *>(p.res + (size_t)m * p.Cout + c0 + c16 * 8);
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
        *reinterpret_cast<uint4 *>(&stage[px * HL_SLD + c16 * 8]) = rv[u];
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
#pragma unroll
  for (int i0 = 0; i0 < 4096 / NTH; i0 += 8) {
    uint4 ov[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
      ov[u] = *reinterpret_cast<const uint4 *>(&stage[px * HL_SLD + c16 * 8]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + NTH * (i0 + u), px = idx >> 4, c16 = idx & 15;
      const int m = m0 + px;
      if (m < p.M) {
        const bool hi = m >= p.split_m;
        const long long orow = hi ? (long long)(m - p.split_m) : (long long)m;
        const int coff = hi ? p.coff_hi : 0;
        *reinterpret_cast<uint4 *>((f16 *)p.out + orow * p.out_ld + coff + c0 + c16 * 8) = ov[u];
      }
    }
  }
}

bool conv_halo_supported(const ConvArgs &a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.H == a.W && (a.W == 40 || a.W == 20) && a.Cin % HL_CK == 0 &&
         a.Cout % HL_BM == 0 && a.out_mode == 0 && a.Kpad == 9 * a.Cin && a.out_ld % 8 == 0 && a.coff_hi % 8 == 0;
}

template <int W, int NWN>
static int launch_halo_w(const ConvArgs &a, hipStream_t s) {
  using C = HaloCfg<W>;
  static bool attr_set = false;
  if (!attr_set) {
    FP_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_halo_kernel<W, NWN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid(((a.M + HL_TM - 1) / HL_TM) * (a.Cout / HL_BM));
  hipLaunchKernelGGL((conv3x3_halo_kernel<W, NWN>), grid, dim3(128 * NWN), C::LDS_BYTES, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int g_halo_nwn = 2;   // tuning knob (FP_HALO_NWN env, read once in api.hip)

int launch_conv_halo(const ConvArgs &a, hipStream_t s) {
  if (g_halo_nwn == 2) {
    if (a.W == 40) return launch_halo_w<40, 2>(a, s);
    return launch_halo_w<20, 2>(a, s);
  }
  if (a.W == 40) return launch_halo_w<40, 4>(a, s);
  return launch_halo_w<20, 4>(a, s);
}

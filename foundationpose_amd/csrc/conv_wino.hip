// 3x3 / stride-1 / pad-1 convolution as Winograd F(2,3) ALONG ROWS on the matrix cores (gfx950).
//
// The ResnetBasicBlock convolutions (learning/models/network_modules.py:73-111 via refine_network.py:37-50, score_network.py:35-49) are
// 93 % of the network FLOPs, and the direct kernel (conv_halo.hip) holds the board at its power cap at ~0.47 of the dense fp16 peak: the
// matrix work itself has to shrink.  F(2,3) along x: per PAIR of output pixels (x0, x0+1), kernel row ky and input channel
//     d_j = in[y + ky - 1][x0 - 1 + j], j = 0..3       v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)              (B^T d)
//     u   = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)  of the row's three taps                        (G g, packed offline)
//     m_i = sum over (ky, ci) of u_i v_i                 y(x0) = m0 + m1 + m2,  y(x0+1) = m1 - m2 - m3       (A^T m)
// 4 multiplies for 2 outputs instead of 6: 2/3 of the MFMAs, at twice the accumulators per output.  Numerics: u is rounded to fp16 ONCE
// from the fp32 BN-folded weights, v is ONE fp16 add of fp16 activations (v_pk_add_f16), m accumulates in fp32 - measured on the
// reference's modules (tests/tools/winograd_precision.py, mode W1): rms error of RefineNet's outputs 1.4 x the direct fp16 form's, 0.7 -
// 1.1 x the reference's own fp16 autocast.
//
// Workgroup tile = 512 pixels (256 pairs) x 64 couts = 65536 fp32 accumulators, the register file's worth; two forms (template NW):
//   NW = 8 (default): 8 waves, two per SIMD, 256 registers each.  Wave w owns the 32-pair tile w x two 32-cout tiles x the four m_i
//     (8 accumulator tiles); per k-step (ky, 16 channels): 4 raw + 8 weight ds_read_b128, 16 v_pk_add_f16, 8 MFMAs.  1.5 LDS reads per
//     MFMA is a lot - but a Winograd tile needs 245 bytes of L2 -> LDS traffic per MFMA (the direct kernel: 95), the LDS-DMA that
//     carries it costs the issuing wave ~65 cycles per KB, and only a second wave on the SIMD turns those cycles into MFMA time.
//   NW = 4: 4 waves, ONE per SIMD, 512 registers each (64 pairs per wave: 1.0 LDS read per MFMA).  Measured first (profiles/
//     r05_experiments.md): with nothing to cover its DMA issue a group of 32 MFMAs (1024 cycles) took 1940; kept for A/B.
// LDS: the input band of a 32-channel chunk as TWO COLUMN-PARITY PLANES (even x / odd x; a pair's d1, d3 are neighbours in the even
// plane, d0, d2 in the odd one, so lanes read consecutive 64-byte entries: conflict-free with the same XOR swizzle as conv_halo.hip),
// in PADDED coordinates: every plane row carries one zero entry (x = -1 / x = W) and every image one zero row above it, which the
// LDS-DMA fills from the context's zero page - so the inner loop has no border selects at all.  Double buffered (2 x 46 KB); the
// transformed weights of one (chunk, ky) are a contiguous, pre-swizzled 16-KB block in global memory, ring of 3.
#include "common.h"
#include <cstdlib>

#define WN_CK 32
#define WN_BN 64
#define WN_TM 512
#define WN_RING 3
#define WN_WHALFS (4 * WN_BN * WN_CK)   // halfs per (chunk, ky) weight block: [i 4][co 64][32 ci] = 16 KB

#ifdef HALO_STAMP
// diagnostic build only (make -B EXTRA=-DHALO_STAMP): per-wave cycle sums, see scripts/wino_stamps.py
__device__ unsigned long long g_wino_stamps[4096 * 8 * 8];
extern "C" __attribute__((visibility("default"))) int fp_dbg_wino_stamps(unsigned long long *host, int clear) {
  if (host) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamps), sizeof(g_wino_stamps));
  if (clear) {
    static unsigned long long z[4096 * 8 * 8];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamps), z, sizeof(z));
  }
  return 0;
}
#define WSTAMP(...) __VA_ARGS__
#ifndef WN_EXP
#define WN_EXP 0        // timing experiments of the diagnostic build (WRONG results): 1 no v = B^T d, 2 no raw reads, 3 no weight-fragment reads, 4 no DMA behind B_g,
                        // 5 no band DMA, 6 no weight DMA, 7 band DMA from CONTIGUOUS addresses (1 KB per instruction instead of 16 pieces of 64 B)
#endif
#else
#define WN_EXP 0
#define WSTAMP(...)
#endif

template <int V>
struct WIC {
  static constexpr int value = V;
};

template <int W>
struct WinoCfg {
  static constexpr int H = W;
  static constexpr int RL = W / 2 + 1;                  // entries per plane row (W/2 pixels of one parity + the zero pad)
  static constexpr int NRP = W == 40 ? 17 : 31;         // padded rows a 512-pixel tile can touch (rows + 2 halo + zero rows between images)
  static constexpr int NI = (NRP * RL + 15) / 16;       // DMA instructions per plane and chunk
  static constexpr int PE = NI * 16;                    // entries allocated per plane
  static constexpr int BAND_HALFS = 2 * PE * 32;
  static constexpr int WOFF = 2 * BAND_HALFS;
  static constexpr int LDS_HALFS_MAIN = WOFF + WN_RING * WN_WHALFS;
  static constexpr int LDS_BYTES = 2 * LDS_HALFS_MAIN;            // (the epilogue's fp32 staging, 4 x 33 KB, fits inside)
};

// LDS-DMA from inline asm (outside the compiler's waitcnt bookkeeping, see conv_halo.hip): m0 = wave-uniform LDS byte address, lane i lands at + 16 i
// (destinations are LDS BYTE addresses: the kernel casts its LDS array to address space 3 once - a generic -> LDS cast per call makes hipcc
// emit a null check against src_shared_base per DMA, and in the stamped build an illegal VOPC operand)
__device__ __forceinline__ void wn_glds16_v(const void *g, unsigned lds_byte) {                 // per-lane 64-bit source address
  const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory");
}
// ... through a buffer resource: lanes whose offset lies outside the buffer (>= num_records: the 0x80000000 of the pad entries / zero rows)
// land ZEROS - the hardware's own border handling, no zero page, no 64-bit address arithmetic per lane
typedef int wn_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wn_glds16_b(wn_i32x4 srd, unsigned voff_bytes, unsigned lds_byte) {
  const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte);
  asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff_bytes), "s"(srd), "s"(la) : "memory");
}
__device__ __forceinline__ void wn_glds16_s(const f16 *sbase, unsigned voff_bytes, unsigned lds_byte) {   // scalar base + 32-bit lane offset
  const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(la) : "memory");
}

// a - b on eight packed halfs: v_pk_add_f16 with the negate modifiers on the second operand (hipcc scalarises a v2f16 fsub into
// v_sub_f16 + v_sub_f16_sdwa + v_pack_b32_f16: three instructions where one does)
typedef unsigned int wn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 wn_pk_sub(half8 a, half8 b) {
  const wn_u32x4 ua = __builtin_bit_cast(wn_u32x4, a), ub = __builtin_bit_cast(wn_u32x4, b);
  wn_u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r[k]) : "v"(ua[k]), "v"(ub[k]));
  return __builtin_bit_cast(half8, r);
}

template <int N>
__device__ __forceinline__ void wn_wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
  else static_assert(N < 0, "add the vmcnt immediate");
}

template <int W, int NW, bool RES, bool POST>
__global__ __launch_bounds__(NW * 64) void conv3x3_wino_kernel(ConvArgs p, const f16 *__restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  using C = WinoCfg<W>;
  static_assert(NW == 4 || NW == 8, "4 waves x 2 pair tiles or 8 waves x 1");
  constexpr int NPT = 8 / NW;                            // 32-pair tiles per wave
  constexpr int H = C::H, RL = C::RL, PE = C::PE, NI = C::NI;
  constexpr int HQ = (2 * NI + NW - 1) / NW;             // band DMA instructions per wave and chunk (12 | 6; 20x20 maps: 11 | 6)
  constexpr int WQ = 16 / NW;                            // weight DMA instructions per wave and group (4 | 2)
  constexpr int PXW = 64 * NPT;                          // pixels per wave
  const int n_ct = p.Cout / WN_BN;
  const int Ltile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (Ltile / n_ct) * WN_TM, ct64 = Ltile % n_ct, c0 = ct64 * WN_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nchunk = p.Cin / WN_CK, G = nchunk * 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;       // LDS byte address of the array
  WSTAMP(const unsigned long long t_entry = __builtin_amdgcn_s_memtime(); const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
         unsigned long long t_wait = 0, t_k0 = 0, t_k1 = 0;)
  // static priority for the second-dispatched half of an 8-wave workgroup (the arbitration loser of each SIMD pair, see conv_halo.hip)
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  // padded row of the tile's first pixel (ky = 0 tap): every band row index is relative to it
  const int n0 = m0 / (H * W), y0 = (m0 - n0 * H * W) / W;
  const int R0 = n0 * (H + 1) + y0;

  // ---- band DMA sources: instruction h of this wave = plane entry block q = wave * HQ + h (surplus slots repeat the last block).
  // (row, entry) of the lane's plane entry advance by 16 entries per instruction: one division for the wave's first block, none after ----
  unsigned hoff[HQ];          // byte offset of the lane's 16 bytes inside p.in for chunk 0; 0x80000000: outside the buffer -> zeros
  {
    int q = min(wave * HQ, 2 * NI - 1);
    int plane = q >= NI ? 1 : 0, instr = q - plane * NI;
    int L = instr * 16 + (lane >> 2);
    int r = L / RL, e = L - r * RL;
#pragma unroll
    for (int h = 0; h < HQ; ++h) {
      int yy = y0 + r, n = n0;               // padded row R0 + r = n (H + 1) + yy; r < NRP <= 2 (H + 1)
      if (yy >= H + 1) yy -= H + 1, ++n;
      if (yy >= H + 1) yy -= H + 1, ++n;
      const bool ok = yy != 0 && n < p.Nimg && (plane == 0 ? e < W / 2 : e >= 1);
      const int x = plane == 0 ? 2 * e : 2 * e - 1;
      const int cg = (lane & 3) ^ ((L >> 2) & 3);
      hoff[h] = ok ? (unsigned)((((n * H + yy - 1) * W + x) * p.Cin + cg * 8) * 2) : 0x80000000u;
      if (q < 2 * NI - 1) {                  // next block (the last one repeats)
        ++q;
        if (q == NI) {
          plane = 1, L = lane >> 2, r = 0, e = L;
        } else {
          L += 16, e += 16;
        }
        if (e >= RL) e -= RL, ++r;
        if (e >= RL) e -= RL, ++r;
      }
    }
  }
  // buffer resource over the input tensor (stride 0, num_records = its bytes); the chunk's 64-byte column offset goes into the base
  const unsigned long long in_addr = (unsigned long long)(size_t)p.in;
  const int in_bytes = (int)((long long)p.M * p.Cin * 2);
  auto band_dma = [&](int cc, auto hc) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    if constexpr (h < HQ) {
      const int q = min(wave * HQ + h, 2 * NI - 1);
      const unsigned long long base = in_addr + (unsigned)cc * (WN_CK * 2);
      wn_i32x4 srd;
      srd[0] = (int)(unsigned)base, srd[1] = (int)(unsigned)(base >> 32) & 0xffff, srd[2] = in_bytes, srd[3] = 0x00020000;
      wn_glds16_b(srd, hoff[h], lds0 + ((cc & 1) * C::BAND_HALFS + q * 512) * 2);
    }
  };
  auto band_dma_all = [&](int cc) __attribute__((always_inline)) {
    band_dma(cc, WIC<0>{}), band_dma(cc, WIC<1>{}), band_dma(cc, WIC<2>{}), band_dma(cc, WIC<3>{}), band_dma(cc, WIC<4>{}), band_dma(cc, WIC<5>{});
    band_dma(cc, WIC<6>{}), band_dma(cc, WIC<7>{}), band_dma(cc, WIC<8>{}), band_dma(cc, WIC<9>{}), band_dma(cc, WIC<10>{}), band_dma(cc, WIC<11>{});
    static_assert(HQ <= 12, "extend the list");
  };
  // ---- weight DMA: group g = (cc, ky) is one contiguous 16-KB block, 16 instructions, WQ per wave ----
  const f16 *wsrc = p.wwino + (size_t)ct64 * G * WN_WHALFS + wave * WQ * 512;
  auto w_dma_part = [&](int g, int u0, int u1) __attribute__((always_inline)) {
    const f16 *sb = wsrc + (size_t)g * WN_WHALFS;
    const unsigned dst = lds0 + (C::WOFF + (g % WN_RING) * WN_WHALFS + wave * WQ * 512) * 2;
#pragma unroll
    for (int u = u0; u < u1; ++u) wn_glds16_s(sb + u * 512, (unsigned)lane * 16u, dst + u * 1024);
  };

  // ---- per-lane read bases: entry of (ky = 0, d1) of this lane's pair in pair tile j ----
  int Lb[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int m = min(m0 + (wave * NPT + j) * 64 + 2 * lr, p.M - 2);
    const int n = m / (H * W), rem = m - n * H * W, y = rem / W, x0 = rem - y * W;
    Lb[j] = (n * (H + 1) + y - R0) * RL + (x0 >> 1);
  }
  // weight fragment: row co = ct * 32 + lr, channel group ks * 2 + lh at position ^ ((co >> 2) & 3)
  int wa[2];
  wa[0] = lr * 32 + ((lh ^ ((lr >> 2) & 3)) * 8);
  wa[1] = lr * 32 + (((2 + lh) ^ ((lr >> 2) & 3)) * 8);

  // prologue DMA first (its round trip runs under the accumulator set-up), in program order: band(0), weights(0), weights(1)
  band_dma_all(0);
  w_dma_part(0, 0, WQ);
  if (G > 1) w_dma_part(1, 0, WQ);

  // accumulators: m1 starts at the bias (y0 and y1 both take + m1), the others at 0
  floatx16 acc[4][2][NPT];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 bv = *reinterpret_cast<const float4 *>(p.bias + c0 + ct * 32 + rg * 8 + lh * 4);
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][ct][j][rg * 4 + e] = 0.f;
        acc[1][ct][j][rg * 4 + 0] = bv.x;
        acc[1][ct][j][rg * 4 + 1] = bv.y;
        acc[1][ct][j][rg * 4 + 2] = bv.z;
        acc[1][ct][j][rg * 4 + 3] = bv.w;
      }
    }

  // ---- main loop -----------------------------------------------------------------------------------------------------------------
  // Group g = (chunk cc, kernel row ky) = two k-steps of 16 channels; a k-step = four PHASES (one per m_i) of 2 NPT MFMAs.  Every operand is
  // produced a phase or a k-step AHEAD in program order (sched_barrier keeps that order), in the shadow of the MFMAs:
  //   phase i:  [2 ds_read: weight fragments of m_(i+1)]  then per MFMA a filler: [ds_read raw', 2 pk] ... [2 pk, DMA]
  //   raw' = the next k-step's raw fragments (d0, d2 first: v_0 of the next k-step is made in phase 3), pk = the v_pk_add_f16 that make
  //   v_(i+1) (phase 3: v_0 of the next k-step).
  // The one barrier of a group, B_g, sits BETWEEN its two k-steps: in front of it the wave waits for its own DMA of weights(g + 1) (and,
  // in a ky = 2 group, of the next chunk's band), behind it every wave's are visible - so the second k-step can already request the first
  // operands of group g + 1 - and the ring slot of weights(g - 1) and, in a ky = 0 group, the other band buffer are free: the DMA of
  // weights(g + 2) (WQ instructions) and band(cc + 1) (HQ) is issued from the eight phases behind B_g, weights first.
  // vmcnt in front of B_g: younger than weights(g + 1) is only the band that followed it behind B_(g-1), i.e. in a ky = 1 group.
  if (G > 1) wn_wait_vm<WQ>(); else wn_wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  struct Adr {            // LDS half offsets of one group's operands
    int a0[NPT][2], a1[NPT][2];   // [pair tile][ks]: entries L and L + 1 of the even plane (odd plane: + PE * 32)
    int woff;
  };
  // (entry, swizzled position) of every kernel row, once per tile; a group adds its band buffer (one v_add per address)
  int ab0[3][NPT][2], ab1[3][NPT][2];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int L = Lb[j] + ky * RL, s0 = (L >> 2) & 3, s1 = ((L + 1) >> 2) & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ab0[ky][j][ks] = L * 32 + (((ks * 2 + lh) ^ s0) * 8);
        ab1[ky][j][ks] = (L + 1) * 32 + (((ks * 2 + lh) ^ s1) * 8);
      }
    }
  auto make_adr = [&](int cc, int ky, int g) __attribute__((always_inline)) {
    Adr A;
    const int bandoff = (cc & 1) * C::BAND_HALFS;
    A.woff = C::WOFF + (g % WN_RING) * WN_WHALFS;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        A.a0[j][ks] = bandoff + ab0[ky][j][ks];
        A.a1[j][ks] = bandoff + ab1[ky][j][ks];
      }
    return A;
  };
  typedef wn_u32x4 raw_t[NPT][4];
  auto load_raw = [&](const Adr &A, int ks, int j, int k, raw_t &d) __attribute__((always_inline)) {
    // k = 0: d0 (odd plane, L), 1: d1 (even, L), 2: d2 (odd, L + 1), 3: d3 (even, L + 1)
    const int base = (k & 2) ? A.a1[j][ks] : A.a0[j][ks];
    d[j][k] = *reinterpret_cast<const wn_u32x4 *>(&lds[base + ((k & 1) ? 0 : PE * 32)]);
  };
  auto load_w = [&](const Adr &A, int ks, int i, half8 (&af)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) af[ct] = *reinterpret_cast<const half8 *>(&lds[A.woff + i * (WN_BN * WN_CK) + ct * (32 * WN_CK) + wa[ks]]);
  };
  // dwords [k0, k0 + 2) of v_i of pair tile j from its raw fragments
  auto vcalc = [&](int i, const raw_t &d, int j, int k0, wn_u32x4 &v) __attribute__((always_inline)) {
#pragma unroll
    for (int k = k0; k < k0 + 2; ++k) {
      if (i == 0) asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v[k]) : "v"(d[j][0][k]), "v"(d[j][2][k]));
      else if (i == 1) asm("v_pk_add_f16 %0, %1, %2" : "=v"(v[k]) : "v"(d[j][1][k]), "v"(d[j][2][k]));
      else if (i == 2) asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v[k]) : "v"(d[j][2][k]), "v"(d[j][1][k]));
      else asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v[k]) : "v"(d[j][1][k]), "v"(d[j][3][k]));
    }
  };
  // DMA of the phases behind B_g: slot = 0 .. 7 (phases of the second k-step of group g, then of the first k-step of group g + 1):
  // slots 0, 1 the weights (WQ / 2 instructions each), slots 2 .. 7 the band (HQ / 6 each)
  int dma_g = -1, dma_cc = -1;          // group whose barrier released the slots; -1: nothing to issue
  auto dma_slot = [&](int slot) __attribute__((always_inline)) {
    if (dma_g < 0) return;
    if (slot < 2) {
      if (WN_EXP != 6 && dma_g + 2 < G) w_dma_part(dma_g + 2, slot * (WQ / 2), (slot + 1) * (WQ / 2));
    } else if (WN_EXP != 5 && dma_cc >= 0) {
      constexpr int PER = (HQ + 5) / 6;
      auto some = [&](auto h0) __attribute__((always_inline)) {
        constexpr int h = decltype(h0)::value;
        band_dma(dma_cc, WIC<h>{});
        if constexpr (PER > 1) band_dma(dma_cc, WIC<h + 1>{});
      };
      if (slot == 2) some(WIC<0>{});
      else if (slot == 3) some(WIC<PER>{});
      else if (slot == 4) some(WIC<2 * PER>{});
      else if (slot == 5) some(WIC<3 * PER>{});
      else if (slot == 6) some(WIC<4 * PER>{});
      else some(WIC<5 * PER>{});
    }
  };
  // one k-step (A, ks); (An, ksn) is the next one.  In: dcur raw, w0 weight fragments of m_0, va = v_0.  Out: the same for the next step.
  auto kstep = [&](const Adr &A, int ks, const Adr &An, int ksn, raw_t &dcur, raw_t &dnxt, half8 (&w0)[2], half8 (&w1)[2], wn_u32x4 (&va)[NPT],
                   wn_u32x4 (&vb)[NPT], int slot0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      half8(&wi)[2] = (i & 1) ? w1 : w0;          // m_0, m_2 in w0's registers, m_1, m_3 in w1's
      half8(&wn)[2] = (i & 1) ? w0 : w1;
      wn_u32x4(&vi)[NPT] = (i & 1) ? vb : va;
      wn_u32x4(&vn)[NPT] = (i & 1) ? va : vb;
      const raw_t &dsrc = i == 3 ? dnxt : dcur;   // v_(i+1) of this step, or v_0 of the next
      const int in = (i + 1) & 3;
      if (WN_EXP != 3) {
        if (i < 3) load_w(A, ks, i + 1, wn);
        else load_w(An, ksn, 0, wn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 2 * NPT; ++f) {         // MFMA f = (pair tile f / 2, cout tile f % 2), then its filler
        const int j = f >> 1, ct = f & 1;
        acc[i][ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wi[ct], __builtin_bit_cast(half8, vi[j]), acc[i][ct][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (WN_EXP != 2) {
          // the next k-step's raw fragments, d0 and d2 in phases 0 and 1: NPT = 2: (j0: d0 d2 | j1: d0 d2 | d1 d1 | d3 d3), NPT = 1: (d0 | d2 | d1 | d3)
          if (NPT == 2 && f < 2) load_raw(An, ksn, i < 2 ? i : f, i == 0 || i == 1 ? f * 2 : i == 2 ? 1 : 3, dnxt);
          if (NPT == 1 && f == 0) load_raw(An, ksn, 0, i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 1 : 3, dnxt);
        }
        if (WN_EXP != 1) vcalc(in, dsrc, j, ct * 2, vn[j]);
        if (f == 2 * NPT - 2 && WN_EXP != 4 && slot0 >= 0) dma_slot(slot0 + i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  raw_t dA, dB;
  half8 w0[2], w1[2];
  wn_u32x4 va[NPT], vb[NPT];
  Adr cur = make_adr(0, 0, 0);
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) load_raw(cur, 0, j, k, dA);
  load_w(cur, 0, 0, w0);
#pragma unroll
  for (int j = 0; j < NPT; ++j) vcalc(0, dA, j, 0, va[j]), vcalc(0, dA, j, 2, va[j]);
  int g = 0;
  WSTAMP(const unsigned long long t_loop = __builtin_amdgcn_s_memtime();)
  for (int cc = 0; cc < nchunk; ++cc) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky, ++g) {
      WSTAMP(const unsigned long long ta = __builtin_amdgcn_s_memtime();)
      kstep(cur, 0, cur, 1, dA, dB, w0, w1, va, vb, dma_g >= 0 ? 4 : -1);
      WSTAMP(const unsigned long long tb = __builtin_amdgcn_s_memtime(); t_k0 += tb - ta;)
      dma_g = -1;
      if (g + 1 < G) {
        if (ky == 1 && cc + 1 < nchunk) wn_wait_vm<HQ>();
        else wn_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dma_g = g;
        dma_cc = (ky == 0 && cc + 1 < nchunk) ? cc + 1 : -1;
      }
      WSTAMP(const unsigned long long tc = __builtin_amdgcn_s_memtime(); t_wait += tc - tb;)
      const Adr nxt = ky == 2 ? make_adr(cc + 1, 0, g + 1) : make_adr(cc, ky + 1, g + 1);     // (past the last group: harmless reads inside the allocation)
      kstep(cur, 1, nxt, 0, dB, dA, w0, w1, va, vb, dma_g >= 0 ? 0 : -1);
      WSTAMP(t_k1 += __builtin_amdgcn_s_memtime() - tc;)
      cur = nxt;
    }
  }
  WSTAMP(const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();)

  // ---------------- epilogue, per wave: the wave owns PXW consecutive pixels x 64 couts ----------------
  // y = A^T m in fp32 goes through a PRIVATE fp32 staging area (PXW rows x 66 floats: the accumulator layout - lane = pair, 4 couts per
  // register quad - becomes rows; stride 66 keeps the 16-byte writes of 8 lanes on 8 different bank groups), is read back as rows of 8
  // couts, gets residual (loaded into registers BEFORE the transform, so its latency hides behind it) + ReLU (+ positional embedding) in
  // fp32 and is rounded once.  No second barrier: a wave's LDS operations execute in order.
  constexpr int NPC = PXW * 8 / 64;                 // 16-byte output pieces per lane
  constexpr int FLD = 66;
  static_assert(WN_TM * FLD * 4 <= C::LDS_BYTES, "staging");
  const int mw = m0 + wave * PXW;
  uint4 rv[NPC];
  if constexpr (RES) {
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
      const int idx = lane + 64 * u, px = idx >> 3, c8 = idx & 7;
      const int m = min(mw + px, p.M - 1);
      rv[u] = *reinterpret_cast<const uint4 *>(p.res + (size_t)m * p.Cout + c0 + c8 * 8);
    }
  }
  __syncthreads();                                  // every wave is done with the band / weight images
  float *st = reinterpret_cast<float *>(lds) + wave * (PXW * FLD);
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float4 y0v, y1v;
        {
          const floatx16 &a0 = acc[0][ct][j], &a1 = acc[1][ct][j], &a2 = acc[2][ct][j], &a3 = acc[3][ct][j];
          y0v = make_float4((a0[rg * 4 + 0] + a1[rg * 4 + 0]) + a2[rg * 4 + 0], (a0[rg * 4 + 1] + a1[rg * 4 + 1]) + a2[rg * 4 + 1],
                            (a0[rg * 4 + 2] + a1[rg * 4 + 2]) + a2[rg * 4 + 2], (a0[rg * 4 + 3] + a1[rg * 4 + 3]) + a2[rg * 4 + 3]);
          y1v = make_float4((a1[rg * 4 + 0] - a2[rg * 4 + 0]) - a3[rg * 4 + 0], (a1[rg * 4 + 1] - a2[rg * 4 + 1]) - a3[rg * 4 + 1],
                            (a1[rg * 4 + 2] - a2[rg * 4 + 2]) - a3[rg * 4 + 2], (a1[rg * 4 + 3] - a2[rg * 4 + 3]) - a3[rg * 4 + 3]);
        }
        float *o = st + (j * 64 + 2 * lr) * FLD + ct * 32 + rg * 8 + lh * 4;
        *reinterpret_cast<float4 *>(o) = y0v;
        *reinterpret_cast<float4 *>(o + FLD) = y1v;
      }
  const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
  for (int u = 0; u < NPC; ++u) {
    const int idx = lane + 64 * u, px = idx >> 3, c8 = idx & 7;
    const float4 f0 = *reinterpret_cast<const float4 *>(st + px * FLD + c8 * 8), f1 = *reinterpret_cast<const float4 *>(st + px * FLD + c8 * 8 + 4);
    float y[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    if constexpr (RES) {
      const half8 rq = __builtin_bit_cast(half8, rv[u]);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] += (float)rq[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], lo);
    const int m = mw + px;
    if constexpr (POST) {
      const float *pp = p.post_add + (size_t)(min(m, p.M - 1) % p.post_period) * p.Cout + c0 + c8 * 8;
      const float4 p0 = *reinterpret_cast<const float4 *>(pp), p1 = *reinterpret_cast<const float4 *>(pp + 4);
      y[0] += p0.x, y[1] += p0.y, y[2] += p0.z, y[3] += p0.w, y[4] += p1.x, y[5] += p1.y, y[6] += p1.z, y[7] += p1.w;
    }
    half8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) hv[e] = (f16)y[e];
    if (m < p.M) {
      const bool hi = m >= p.split_m;
      const long long orow = hi ? (long long)(m - p.split_m) : (long long)m;
      const int coff = hi ? p.coff_hi : 0;
      *reinterpret_cast<half8 *>((f16 *)p.out + orow * p.out_ld + coff + c0 + c8 * 8) = hv;
    }
  }
  WSTAMP(if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long *o = g_wino_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = t_wait; o[1] = t_k0; o[2] = t_k1; o[3] = t_loop_end - t_loop; o[4] = t_loop - t_entry; o[5] = __builtin_amdgcn_s_memtime() - t_loop_end;
    o[6] = r_entry; o[7] = __builtin_amdgcn_s_memrealtime();
  })
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// fp32 (BN-folded) weights (Cout, Cin, 3, 3) -> the transformed, fp16, DMA-ordered image the kernel streams:
// [Cout / 64][Cin / 32][ky 3][i 4][co 64][position 4][8], channel group cg of row co at position cg ^ ((co >> 2) & 3).
size_t wino_packed_halfs(int Cout, int Cin) { return (size_t)Cout * Cin * 12; }

void wino_pack_weights(const float *w, const float *scale, int Cout, int Cin, f16 *out) {
  const int nchunk = Cin / WN_CK;
  for (int co = 0; co < Cout; ++co) {
    const float sc = scale ? scale[co] : 1.f;
    const int ct64 = co / WN_BN, col = co % WN_BN;
    for (int ci = 0; ci < Cin; ++ci) {
      const int cc = ci / WN_CK, cil = ci % WN_CK, cg = cil >> 3, pos = cg ^ ((col >> 2) & 3);
      for (int ky = 0; ky < 3; ++ky) {
        const float *gp = w + (((size_t)co * Cin + ci) * 3 + ky) * 3;
        const float g0 = gp[0] * sc, g1 = gp[1] * sc, g2 = gp[2] * sc;
        const float u[4] = {g0, 0.5f * ((g0 + g2) + g1), 0.5f * ((g0 + g2) - g1), g2};
        f16 *blk = out + ((size_t)(ct64 * nchunk + cc) * 3 + ky) * WN_WHALFS;
        for (int i = 0; i < 4; ++i) blk[(size_t)i * (WN_BN * WN_CK) + col * WN_CK + pos * 8 + (cil & 7)] = (f16)u[i];
      }
    }
  }
}

bool conv_wino_supported(const ConvArgs &a) {
  return a.wwino && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.H == a.W && (a.W == 40 || a.W == 20) && a.Cin % WN_CK == 0 &&
         a.Cin >= 64 && a.Cout % WN_BN == 0 && a.out_mode == 0 && a.out_ld % 8 == 0 && a.coff_hi % 8 == 0 && a.M % 2 == 0 && !(a.splitk && a.ksplit > 1);
}

int fp_wino_waves() {
  static const int nw = getenv("FP_WINO_WAVES") ? atoi(getenv("FP_WINO_WAVES")) : 8;     // A/B knob: 4 = one wave per SIMD
  return nw == 4 ? 4 : 8;
}

template <int W, int NW>
static void wino_lds_all(std::vector<KernelLds> &v) {
  using C = WinoCfg<W>;
  v.push_back({(const void *)conv3x3_wino_kernel<W, NW, false, false>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_wino_kernel<W, NW, true, false>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_wino_kernel<W, NW, false, true>, C::LDS_BYTES});
  v.push_back({(const void *)conv3x3_wino_kernel<W, NW, true, true>, C::LDS_BYTES});
}
void conv_wino_kernel_lds(std::vector<KernelLds> &v) { wino_lds_all<40, 8>(v), wino_lds_all<20, 8>(v), wino_lds_all<40, 4>(v), wino_lds_all<20, 4>(v); }

template <int W, int NW, bool RES, bool POST>
static int launch_wino(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  using C = WinoCfg<W>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS");
  const int n_t = ((a.M + WN_TM - 1) / WN_TM) * (a.Cout / WN_BN);
  hipLaunchKernelGGL((conv3x3_wino_kernel<W, NW, RES, POST>), dim3(n_t), dim3(NW * 64), C::LDS_BYTES, s, a, (const f16 *)ctx->zero_page);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

template <int W, int NW>
static int launch_wino_flags(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  if (a.post_add) return a.res ? launch_wino<W, NW, true, true>(ctx, a, s) : launch_wino<W, NW, false, true>(ctx, a, s);
  return a.res ? launch_wino<W, NW, true, false>(ctx, a, s) : launch_wino<W, NW, false, false>(ctx, a, s);
}

int launch_conv_wino(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  FP_REQUIRE(conv_wino_supported(a), "conv3x3 (Winograd): unsupported layer");
  FP_REQUIRE((double)a.M * a.Cin * 2.0 < 2147483648.0, "conv3x3 (Winograd): input tensor of %.1f GB exceeds the 2 GB the kernel addresses", (double)a.M * a.Cin * 2e-9);
  if (fp_wino_waves() == 4) return a.W == 40 ? launch_wino_flags<40, 4>(ctx, a, s) : launch_wino_flags<20, 4>(ctx, a, s);
  return a.W == 40 ? launch_wino_flags<40, 8>(ctx, a, s) : launch_wino_flags<20, 8>(ctx, a, s);
}

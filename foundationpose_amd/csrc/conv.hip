// fp16 NHWC implicit-GEMM convolution / linear layer on MFMA (gfx950).
//
// Replaces the cuDNN convolutions + BN + ReLU + residual adds of learning/models/network_modules.py:37-111
// (BN folded into weights/bias at load) and, with KH=KW=1, every nn.Linear of the transformer heads.
//
//   out[m][co] = act( sum_{tap,ci} in[pix(m)+off(tap)][ci] * w[co][tap*Cin+ci] + bias[co] (+ res[m][co]) ) (+ post_add)
//
// GEMM view: M_gemm = Cout (weights are the MFMA A operand), N_gemm = pixels (activations are the B
// operand, gathered per row straight from the NHWC tensor - no im2col buffer), K = KH*KW*Cin.
// With this orientation a lane of the 32x32 accumulator owns ONE pixel and 4 consecutive output
// channels per register quad => 8-byte channel-contiguous NHWC stores, and the "V transposed"
// store used by the attention kernel ([b][head][d][token]) is lane-contiguous.
//
// Tile: 256 pixels x BM couts (BM = 128 | 64) per 256-thread workgroup, 2x2 waves, each wave (BM/2) x 128 via
// v_mfma_f32_32x32x16_f16; both operands staged by LDS-DMA through a 3-deep ring (below).  The 3x3 stride-1 layers
// have their own kernel (conv_halo.hip).
#include "common.h"
#include <cstdlib>
#include <type_traits>

#define CV_BK 32

// ------------------------------------------------------------------------------------------------
// 256 pixels x BM couts per workgroup, both operands staged by LDS-DMA (global_load_lds_dwordx4,
// per-lane gather address, out-of-image taps read a zero page), XOR-swizzled 64-byte rows
// (conflict-free ds_read_b128), 3-deep ring: two K-steps of DMA are in flight under the current
// step's 16 MFMAs per wave.  Used for the 1x1 (Linear) layers, the stride-2 3x3 and the 7x7 stem.
// ------------------------------------------------------------------------------------------------
#define C2_BN 256
#define C2_BK 32

#ifdef HALO_STAMP
// diagnostic build only (make -B EXTRA=-DHALO_STAMP): per-wave cycle counts of prologue / K loop / epilogue
__device__ unsigned long long g_igemm_stamps[4096 * 4 * 4];
extern "C" __attribute__((visibility("default"))) int fp_dbg_igemm_stamps(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_igemm_stamps), sizeof(g_igemm_stamps)) == hipSuccess ? 0 : -1;
}
#define ISTAMP(x) x
#else
#define ISTAMP(x)
#endif

// LDS-DMA from inline asm (see conv_halo.hip: through the builtin hipcc turns every later LDS-read wait into lgkmcnt(0) and
// every barrier into a full vmcnt(0) drain).  Completion is waited for by the explicit s_waitcnt vmcnt(n) before the barriers.
// scalar base + 32-bit lane offset form: no per-lane 64-bit address arithmetic in front of the DMA
__device__ __forceinline__ void glds16s(const f16 *sbase, unsigned voff_bytes, f16 *l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)l);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(la) : "memory");
}
__device__ __forceinline__ void glds16c(const f16 *g, f16 *l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)l);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory");
}

// One workgroup tile: 64 NT pixels x BM couts, 4 waves as 2 (cout halves) x 2 (pixel halves), each wave (BM/2) x 32 NT.
// NT = 4 -> the 256-pixel main tiles, NT = 1 -> 64-pixel tail tiles (conv_igemm2_kernel).  The accumulation order of an
// output element does not depend on NT.
template <int BM, int KW, bool CIN8, int NT, bool RES, bool SPLIT = false>
__device__ __forceinline__ void igemm2_tile(const ConvArgs &p, const f16 *__restrict__ zero_page, const int m0, const int c0, f16 *smem, const int ks = 0) {
  constexpr int TN = 64 * NT;                              // pixels per workgroup
  constexpr int XH = TN * C2_BK, WH = BM * C2_BK;          // halfs per stage
  constexpr int SLD = BM + 8;                              // halfs per staged output row
  ISTAMP(const unsigned long long t_entry = __builtin_amdgcn_s_memtime();)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int WM = BM / 2, MT = WM / 32;
  constexpr int WQ = BM / 64;                              // weight DMA instructions per wave per stage
  const int lr = lane & 31, lh = lane >> 5;
  const int HoWo = p.Ho * p.Wo;
  const int ntaps = p.KH * p.KW;

  // ---- DMA source bookkeeping: this lane feeds LDS chunk (px = (q*4+wave)*16 + lane/4, chp = lane&3) ----
  const int chp = lane & 3;
  long long xbase[NT];
  int iy0[NT], ix0[NT], xch[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int px = (q * 4 + wave) * 16 + (lane >> 2);
    const int m = m0 + px;
    const bool mv = m < p.M;
    const int mm = mv ? m : 0;
    const int n = mm / HoWo, rem = mm - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    iy0[q] = mv ? oy * p.stride - p.pad : -100000;          // invalid pixel -> every tap out of range
    ix0[q] = ox * p.stride - p.pad;
    xbase[q] = (long long)n * p.H * p.W * p.Cin;
    xch[q] = chp ^ ((px >> 2) & 3);                          // source chunk (swizzle on the SOURCE side)
  }
  // weights: instruction q of wave w covers couts (q*4 + w)*16 + lane/4; the swizzle term ((co>>2)&3) = (lane>>4)&3 does not
  // depend on q, so ONE 32-bit lane offset serves every q and K-step, the rest of the address is scalar
  const unsigned woff = (unsigned)(((wave * 16 + (lane >> 2)) * p.Kpad + ((chp ^ ((lane >> 4) & 3)) * 8)) * 2);
  const f16 *wbase = p.w + (size_t)c0 * p.Kpad;
  // DMA instruction d of a stage: d = 0..NT-1 the activation gather (pixel group d), d = NT.. the weights
  // 1x1 (Linear) layers: a pixel's K-row is contiguous, the gather address of step kt is a fixed 32-bit lane offset from the
  // scalar base p.in + kt*64 B - no per-step tap / bounds / 64-bit address arithmetic in front of the DMA
  // (pixels past M are clamped to the last pixel: their products only reach output rows that are never stored)
  unsigned xlin[NT];
  const bool lin = KW == 1 && !CIN8 && p.stride == 1 && p.pad == 0 && (double)p.M * p.Cin * 2.0 < 4294967296.0;   // output pixel m reads input pixel m
  if (lin) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int m = min(m0 + (q * 4 + wave) * 16 + (lane >> 2), p.M - 1);
      xlin[q] = (unsigned)(m * p.Cin + xch[q] * 8) * 2u;
    }
  }
  auto stage_one = [&](int kt, int buf, int d) __attribute__((always_inline)) {
    f16 *xs = smem + buf * (XH + WH), *ws = xs + XH;
    if (d < NT) {
      const int q = d;
#ifdef IGEMM_SKIP_A
      if (kt > 1) return;      // timing experiment only (wrong results): no activation DMA after the first two stages
#endif
      if (lin) {
        glds16s(p.in + kt * C2_BK, xlin[q], xs + (q * 4 + wave) * 512);
        return;
      }
      const int k = kt * C2_BK + xch[q] * 8;
      int tap, ci;
      if (CIN8) {
        tap = k >> 3;
        ci = 0;
      } else {
        tap = (kt * C2_BK) / p.Cin;
        ci = k - tap * p.Cin;
      }
      const int ky = tap / KW, kx = tap - ky * KW;
      const int iy = iy0[q] + ky, ix = ix0[q] + kx;
      const bool ok = tap < ntaps && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const f16 *src = ok ? p.in + xbase[q] + ((long long)iy * p.W + ix) * p.Cin + ci : zero_page;
      glds16c(src, xs + (q * 4 + wave) * 512);
    } else {
      const int q = d - NT;
#ifdef IGEMM_SKIP_W
      if (kt > 1) return;      // timing experiment only (wrong results): no weight DMA after the first two stages
#endif
      glds16s(wbase + (size_t)(q * 64) * p.Kpad + (size_t)kt * C2_BK, woff, ws + (q * 4 + wave) * 512);
    }
  };
  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < NT + WQ; ++d) stage_one(kt, buf, d);
  };

  // ---- fragment bases (swizzled rows: chunk ^ ((row>>2)&3); rows +32 keep the same swizzle) ----
  int wa[2], xa[NT][2];
  {
    const int co = wm * WM + lr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ks] = co * 32 + (((ks * 2 + lh) ^ ((co >> 2) & 3)) * 8);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int px = wn * (32 * NT) + j * 32 + lr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xa[j][ks] = px * 32 + (((ks * 2 + lh) ^ ((px >> 2) & 3)) * 8);
    }
  }

  // accumulators start at the bias (fp32): a lane owns channels wm*WM + i*32 + rg*8 + lh*4 + (0..3) of its pixels
  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float4 bv = *reinterpret_cast<const float4 *>(p.bias + c0 + wm * WM + i * 32 + rg * 8 + lh * 4);
      if (SPLIT && ks != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);          // (split-K: the bias rides in share 0)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j][rg * 4 + 0] = bv.x;
        acc[i][j][rg * 4 + 1] = bv.y;
        acc[i][j][rg * 4 + 2] = bv.z;
        acc[i][j][rg * 4 + 3] = bv.w;
      }
    }

  // split-K (SPLIT): share ks of p.ksplit walks K-steps [kb, kb + nk) and leaves fp32 sums for splitk_finish_kernel (conv_halo.hip)
  const int nk = SPLIT ? p.Kpad / C2_BK / p.ksplit : p.Kpad / C2_BK, kb = SPLIT ? ks * nk : 0;
  // 3-deep LDS-DMA ring: K-steps kt+1 and kt+2 stay in flight across the barrier (counted vmcnt + raw s_barrier;
  // a __syncthreads() would drain them).  Every wave issues exactly DMAW instructions per stage.
  constexpr int DMAW = NT + WQ;
  ISTAMP(const unsigned long long t_loop = __builtin_amdgcn_s_memtime();)
  stage(kb, 0);
  if (nk > 1) stage(kb + 1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt % 3;
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMAW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // stage kt landed for every wave; slot (kt+2)%3 was last read in step kt-1: free
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) stage(kb + kt + 2, (kt + 2) % 3);
    const f16 *xs = smem + cur * (XH + WH), *ws = xs + XH;
    // two k-steps per stage, software-pipelined like the halo kernel: pixel-tile-major MFMA order, the pixel fragment of
    // step 1 is requested as soon as step 0's MFMAs on that register have been issued (counted lgkmcnt waits).  (Issuing
    // the DMAs of stage kt+2 from between the MFMAs, as the halo kernel does, was measured 2-4 % slower here: these layers
    // are bound by the L2 -> LDS fill, and the fill wants its requests as early as possible.)
    half8 af[2][MT], bf[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const half8 *>(&ws[wa[0] + i * 32 * 32]);
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const half8 *>(&xs[xa[j][0]]);
#pragma unroll
    for (int i = 0; i < MT; ++i) af[1][i] = *reinterpret_cast<const half8 *>(&ws[wa[1] + i * 32 * 32]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][i], bf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) bf[j] = *reinterpret_cast<const half8 *>(&xs[xa[j][1]]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue, fp16 NHWC: residual staged through LDS and added in fp32, LDS transpose, 16-byte row stores.  The K loop
  // of a Linear layer is only 16 steps long, so the epilogue weighs a quarter of the tile; it is written to issue few vector
  // instructions: ReLU on the packed fp16 pair after conversion (rounding is monotonic and keeps 0: same result as before
  // it), no per-row output select when the whole tile lies on one side of split_m.  (Reading the residual as 8-byte
  // gathers in the accumulator layout behind the first DMA stages was measured: epilogue -6.8k cycles, K loop +16k.) ----
  ISTAMP(const unsigned long long t_epi = __builtin_amdgcn_s_memtime();)
  if constexpr (SPLIT) {      // fp32 sums of this share -> p.splitk[ks][m][Cout], 16-byte stores in the accumulator layout
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int m = m0 + wn * (32 * NT) + j * 32 + lr;
      if (m >= p.M) continue;
      float *o = p.splitk + ((size_t)ks * p.M + m) * p.Cout + c0 + wm * WM + lh * 4;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *reinterpret_cast<float4 *>(o + i * 32 + rg * 8) = make_float4(acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]);
    }
    return;
  }
  if (p.out_mode == 0) {
    f16 *stage = smem;
    constexpr int CPR = BM / 8;           // 16-byte chunks per staged row
    constexpr int NCH = TN * CPR / 256;
    constexpr int RB = NCH < 8 ? NCH : 8;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rv[NCH];
    if constexpr (RES) {                  // residual tile: ONE batch of coalesced 16-byte loads before the barrier, staged through LDS
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int idx = tid + 256 * u, px = idx / CPR, c16 = idx % CPR;
        const int m = min(m0 + px, p.M - 1);    // unconditional (clamped) load: a guarded one makes hipcc wait per element
        rv[u] = *reinterpret_cast<const u32x4 *>(p.res + (size_t)m * p.Cout + c0 + c16 * 8);
      }
    }
    __syncthreads();
    if constexpr (RES) {
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int idx = tid + 256 * u, px = idx / CPR, c16 = idx % CPR;
        *reinterpret_cast<u32x4 *>(&stage[px * SLD + c16 * 8]) = rv[u];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NT; ++j) {      // each lane reads back exactly the 8-byte slots it overwrites below: no barrier in between
        const int pxl = wn * (32 * NT) + j * 32 + lr;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const half4 rq = *reinterpret_cast<const half4 *>(&stage[pxl * SLD + wm * WM + i * 32 + rg * 8 + lh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][rg * 4 + e] += (float)rq[e];
          }
      }
    }
    if (p.post_add) {                     // fp32 add after the ReLU (positional embedding): one layer per forward
      const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int pxl = wn * (32 * NT) + j * 32 + lr;
        const int prow = min(m0 + pxl, p.M - 1) % p.post_period;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int col = wm * WM + i * 32 + rg * 8 + lh * 4;
            const float4 pv = *reinterpret_cast<const float4 *>(p.post_add + (size_t)prow * p.Cout + c0 + col);
            half4 hv;
            hv[0] = (f16)(fmaxf(acc[i][j][rg * 4 + 0], lo) + pv.x);
            hv[1] = (f16)(fmaxf(acc[i][j][rg * 4 + 1], lo) + pv.y);
            hv[2] = (f16)(fmaxf(acc[i][j][rg * 4 + 2], lo) + pv.z);
            hv[3] = (f16)(fmaxf(acc[i][j][rg * 4 + 3], lo) + pv.w);
            *reinterpret_cast<half4 *>(&stage[pxl * SLD + col]) = hv;
          }
      }
    } else {
      auto convert = [&](auto relu_c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int pxl = wn * (32 * NT) + j * 32 + lr;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              half4 hv;
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[e] = (f16)acc[i][j][rg * 4 + e];
              if constexpr (decltype(relu_c)::value) hv = __builtin_elementwise_max(hv, half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
              *reinterpret_cast<half4 *>(&stage[pxl * SLD + wm * WM + i * 32 + rg * 8 + lh * 4]) = hv;
            }
        }
      };
      if (p.relu) convert(std::true_type{});
      else convert(std::false_type{});
    }
    __syncthreads();
    const bool side_hi = m0 >= p.split_m;
    if (side_hi || m0 + TN <= p.split_m) {          // whole tile on one side: one scalar base, rows at a fixed stride
      f16 *obase = (f16 *)p.out + (long long)(side_hi ? m0 - p.split_m : m0) * p.out_ld + (side_hi ? p.coff_hi : 0) + c0;
#pragma unroll
      for (int i0 = 0; i0 < NCH; i0 += RB) {
        uint4 ov[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int idx = tid + 256 * (i0 + u), px = idx / CPR, c16 = idx % CPR;
          ov[u] = *reinterpret_cast<const uint4 *>(&stage[px * SLD + c16 * 8]);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int idx = tid + 256 * (i0 + u), px = idx / CPR, c16 = idx % CPR;
          if (m0 + px < p.M) *reinterpret_cast<uint4 *>(obase + (long long)px * p.out_ld + c16 * 8) = ov[u];
        }
      }
    } else {
#pragma unroll
      for (int i0 = 0; i0 < NCH; i0 += RB) {
        uint4 ov[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int idx = tid + 256 * (i0 + u), px = idx / CPR, c16 = idx % CPR;
          ov[u] = *reinterpret_cast<const uint4 *>(&stage[px * SLD + c16 * 8]);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int idx = tid + 256 * (i0 + u), px = idx / CPR, c16 = idx % CPR;
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
    ISTAMP(if (lane == 0 && blockIdx.x < 4096) {
      unsigned long long *o = g_igemm_stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
      const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
      if (wave == 1) {          // wave 1: absolute entry / exit times and the hardware slot (HW_ID | XCC_ID << 32): per-CU timelines
        o[0] = t_entry; o[1] = t_exit;
        o[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
      } else {
        o[0] = t_loop - t_entry; o[1] = t_epi - t_loop; o[2] = t_exit - t_epi;
      }
      o[3] = 1;
    })
    return;
  }

  // ---- epilogue, other output modes (fp32 NHWC, transposed V): direct stores ----
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int m = m0 + wn * (32 * NT) + j * 32 + lr;
    if (m >= p.M) continue;
    const bool hi = m >= p.split_m;
    const long long orow = hi ? (long long)(m - p.split_m) : (long long)m;
    const int coff = hi ? p.coff_hi : 0;
    int prow = 0;
    if (p.post_add) prow = m % p.post_period;
    int tb = 0, tt = 0;
    if (p.out_mode == 2) {
      tb = m / p.tokens;
      tt = vt_col(m - tb * p.tokens);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int co = c0 + wm * WM + i * 32 + rg * 8 + lh * 4;
        float v[4];
        v[0] = acc[i][j][rg * 4 + 0];      // the bias is already in the accumulator
        v[1] = acc[i][j][rg * 4 + 1];
        v[2] = acc[i][j][rg * 4 + 2];
        v[3] = acc[i][j][rg * 4 + 3];
        if (p.res) {
          half4 rv = *reinterpret_cast<const half4 *>(p.res + (size_t)m * p.Cout + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.post_add) {
          const float4 pv = *reinterpret_cast<const float4 *>(p.post_add + (size_t)prow * p.Cout + co);
          v[0] += pv.x;
          v[1] += pv.y;
          v[2] += pv.z;
          v[3] += pv.w;
        }
        if (p.out_mode == 0) {
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)v[e];
          *reinterpret_cast<half4 *>((f16 *)p.out + orow * p.out_ld + coff + co) = hv;
        } else if (p.out_mode == 1) {
          *reinterpret_cast<float4 *>((float *)p.out + orow * p.out_ld + coff + co) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = co + e, h = c >> 7, d = c & 127;
            ((f16 *)p.out)[(((size_t)tb * 4 + h) * 128 + d) * 416 + tt] = (f16)v[e];
          }
        }
      }
    }
  }
}

// Grid = n_main workgroups of 256 px x BM co, then n_tail4 workgroups of 64 px covering the LAST main-size tiles cut in four
// (see conv3x3_halo kernels: at N=252 these layers have 1576 or 3150 tiles for 512 slots - a last round that is 8-15 % full).
template <int BM, int KW, bool CIN8, bool RES>
__global__ __launch_bounds__(256, 2) void conv_igemm2_kernel(ConvArgs p, const f16 *__restrict__ zero_page, int n_main) {
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  const int n_ct = p.Cout / BM;
  if ((int)blockIdx.x < n_main) {
    const int L = xcd_remap(blockIdx.x, n_main);
    igemm2_tile<BM, KW, CIN8, 4, RES>(p, zero_page, (L / n_ct) * C2_BN, (L % n_ct) * BM, smem);
  } else {
    const int t = xcd_remap(blockIdx.x - n_main, gridDim.x - n_main);
    const int L = n_main + (t >> 2);
    const int m0 = (L / n_ct) * C2_BN + (t & 3) * (C2_BN / 4);
    if (m0 >= p.M) return;
    igemm2_tile<BM, KW, CIN8, 1, RES>(p, zero_page, m0, (L % n_ct) * BM, smem);
  }
}

// Split-K form of the 3x3 stride-2 layer with the long K (256 -> 512 channels: 72 K-steps) at 3 .. 4 hypotheses (1 .. 2: conv_small.hip; FP_SMALL=0: 1 .. 4), where the layer is
// 28 .. 100 workgroups that each walk all of K behind one DMA round trip per step (38 us at one hypothesis): 64-pixel tiles,
// p.ksplit workgroups per tile with an equal share of the K-steps each, fp32 sums to p.splitk, splitk_finish_kernel adds them in
// share order (the scheme of conv3x3_halo_splitk_kernel: these batch sizes are the size class with its own last bits).
template <int BM, int KW>
__global__ __launch_bounds__(256, 2) void conv_igemm2_splitk_kernel(ConvArgs p, const f16 *__restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  const int n_ct = p.Cout / BM;
  const int ks = blockIdx.x % p.ksplit, t = blockIdx.x / p.ksplit;
  const int m0 = (t / n_ct) * 64;
  if (m0 >= p.M) return;
  igemm2_tile<BM, KW, false, 1, false, true>(p, zero_page, m0, (t % n_ct) * BM, smem, ks);
}

void halo_split(int n_tiles, int slots, int *n_main, int *n_tail4);      // conv_halo.hip: main / quarter-tile split
int launch_splitk_finish(const ConvArgs &a, hipStream_t s);              // conv_halo.hip
int conv_halo_ksplit(const ConvArgs &a, int num_cu);                     // conv_halo.hip

// Split factor of a launch (0: none): the 3x3 stride-1 layers by conv_halo_ksplit; the 3x3 stride-2 layer from 36 K-steps on when
// its 64-pixel tiles fill at most a quarter of the workgroup slots (1 .. 4 hypotheses).  Decided by the caller that owns the scratch.
int conv_ksplit(const ConvArgs &a, int num_cu) {
  if (conv_small_use(a, num_cu)) return 0;           // conv_small.hip splits K inside its workgroups
  if (const int k = conv_halo_ksplit(a, num_cu)) return k;
  const int nk = a.Kpad / C2_BK;
  if (a.KH == 3 && a.KW == 3 && a.stride == 2 && a.out_mode == 0 && a.Cin % 32 == 0 && a.Cout % 128 == 0 && a.M < S2_MIN_PIXELS && nk >= 36 && nk % 4 == 0 &&
      ((a.M + 63) / 64) * (a.Cout / 128) * 4 <= 2 * num_cu)
    return 4;
  return 0;
}

template <int BM>
constexpr int igemm2_lds_bytes() {
  constexpr int main_b = 3 * (C2_BN * C2_BK + BM * C2_BK) * 2, epi_b = C2_BN * (BM + 8) * 2;
  return main_b > epi_b ? main_b : epi_b;
}

template <int BM, int KW, bool CIN8>
static void igemm2_lds_both(std::vector<KernelLds> &v) {
  v.push_back({(const void *)conv_igemm2_kernel<BM, KW, CIN8, true>, igemm2_lds_bytes<BM>()});
  v.push_back({(const void *)conv_igemm2_kernel<BM, KW, CIN8, false>, igemm2_lds_bytes<BM>()});
}

void conv_kernel_lds(std::vector<KernelLds> &v) {       // every instantiation launch_conv can reach
  v.push_back({(const void *)conv_igemm2_splitk_kernel<128, 3>, igemm2_lds_bytes<128>()});
  igemm2_lds_both<128, 7, true>(v), igemm2_lds_both<64, 7, true>(v);
  igemm2_lds_both<128, 3, false>(v), igemm2_lds_both<64, 3, false>(v);
  igemm2_lds_both<128, 1, false>(v), igemm2_lds_both<64, 1, false>(v);
}

template <int BM, int KW, bool CIN8, bool RES>
static int launch_two_r(fp_ctx *ctx, const ConvArgs &a, const f16 *zero_page, hipStream_t s) {
  const int n_tiles = ((a.M + C2_BN - 1) / C2_BN) * (a.Cout / BM);
  int n_main, n_tail4;
  halo_split(n_tiles, 2 * ctx->num_cu, &n_main, &n_tail4);
  constexpr int lds = igemm2_lds_bytes<BM>();
  FP_REQUIRE(a.out_mode != 0 || (a.out_ld % 8 == 0 && a.coff_hi % 8 == 0), "conv: out_ld/coff must be multiples of 8 for fp16 output");
  hipLaunchKernelGGL((conv_igemm2_kernel<BM, KW, CIN8, RES>), dim3(n_main + n_tail4), dim3(256), lds, s, a, zero_page, n_main);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

template <int BM, int KW, bool CIN8>
static int launch_two(fp_ctx *ctx, const ConvArgs &a, const f16 *zero_page, hipStream_t s) {
  // the residual is added in the fp16-NHWC epilogue only; the other output modes read it directly
  return (a.res && a.out_mode == 0) ? launch_two_r<BM, KW, CIN8, true>(ctx, a, zero_page, s) : launch_two_r<BM, KW, CIN8, false>(ctx, a, zero_page, s);
}

bool conv_halo_supported(const ConvArgs &a);
int launch_conv_halo(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);
bool stem_supported(const ConvArgs &a);                     // stem.hip
int launch_stem(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);

int fp_wino_mode() {
  static const int m = getenv("FP_WINO") ? atoi(getenv("FP_WINO")) : 0;
  return m;
}

int launch_conv(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  FP_REQUIRE(a.Cin == 8 || a.Cin % 32 == 0, "conv: Cin=%d must be 8 or a multiple of 32", a.Cin);
  FP_REQUIRE(a.Cout % 64 == 0, "conv: Cout=%d must be a multiple of 64", a.Cout);
  FP_REQUIRE(a.Kpad % CV_BK == 0 && a.Kpad >= a.KH * a.KW * a.Cin, "conv: bad Kpad=%d", a.Kpad);
  FP_REQUIRE(a.KH == a.KW && (a.KW == 1 || a.KW == 3 || a.KW == 7), "conv: kernel %dx%d unsupported (1,3,7)", a.KH, a.KW);
  FP_REQUIRE(a.M == a.Nimg * a.Ho * a.Wo, "conv: M mismatch");
  FP_REQUIRE(a.out_ld % 4 == 0 && a.coff_hi % 4 == 0, "conv: out_ld/coff must be multiples of 4");
  if (a.M == 0) return FP_OK;
  const double flops = 2.0 * (double)a.M * a.Cout * a.KH * a.KW * (a.Cin == 8 ? 6 : a.Cin);
  const bool halo = conv_halo_supported(a);
  const char *cls = halo ? "conv3x3_halo" : (a.KW == 3) ? "conv3x3_s2" : (a.KW == 7 ? "conv7x7" : "linear");
  ProfScope ps(ctx, s, cls, flops);
  if (!(a.splitk && a.ksplit > 1) && conv_small_use(a, ctx->num_cu)) return launch_conv_small(ctx, a, s);      // a few images: conv_small.hip
  if (halo && fp_wino_mode() != 0 && (fp_wino_mode() != 2 || a.Cin == 512) && (fp_wino_mode() != 3 || a.Cin >= 256) && conv_wino_supported(a))
    return launch_conv_wino(ctx, a, s);      // (2: the 512-channel layers only; 3: from 256 channels on)
  {
    static const int band = getenv("FP_C128_BAND") ? atoi(getenv("FP_C128_BAND")) : 1;   // conv_s1b.hip (bit-identical to the halo kernel); 0: off, 2: also the 256 -> 256 layers
    // ... from the batch size on at which the general kernel needs more than one round of its 512-pixel tiles (82 images on 256 CUs): below,
    // one round of those is faster (32 hypotheses per GPU: 6.67 against 6.79 ms per step; 63: 10.24 against 10.13)
    if (band != 0 && (a.Cin == 128 || band == 2) && (long long)a.M > (long long)ctx->num_cu * 512 && s1b_supported(a)) return launch_conv_s1b(ctx, a, s);
  }
  if (halo) return launch_conv_halo(ctx, a, s);
  if (a.splitk && a.ksplit > 1) {        // (conv_ksplit: the 3x3 stride-2 layer with the long K at 3 .. 4 hypotheses; 1 .. 2 run conv_small.hip)
    FP_REQUIRE(a.KW == 3 && a.stride == 2 && a.Cout % 128 == 0 && (a.Kpad / C2_BK) % a.ksplit == 0 && a.out_mode == 0, "conv split-K: unsupported layer");
    const int n_t = ((a.M + 63) / 64) * (a.Cout / 128);
    hipLaunchKernelGGL((conv_igemm2_splitk_kernel<128, 3>), dim3(n_t * a.ksplit), dim3(256), igemm2_lds_bytes<128>(), s, a, (const f16 *)ctx->zero_page);
    FP_CHECK_HIP(hipGetLastError());
    return launch_splitk_finish(a, s);
  }
  if (stem_supported(a)) return launch_stem(ctx, a, s);
  if (a.wpk && a.M >= S2_MIN_PIXELS && s2_supported(a)) return launch_conv_s2(ctx, a, s);
  const bool bm128 = (a.Cout % 128 == 0);
  const f16 *zp = (const f16 *)ctx->zero_page;
  if (a.Cin == 8) {
    FP_REQUIRE(a.KW == 7, "conv: Cin=8 path is the 7x7 stem");
    return bm128 ? launch_two<128, 7, true>(ctx, a, zp, s) : launch_two<64, 7, true>(ctx, a, zp, s);
  }
  if (a.KW == 3) return bm128 ? launch_two<128, 3, false>(ctx, a, zp, s) : launch_two<64, 3, false>(ctx, a, zp, s);
  if (a.KW == 1) return bm128 ? launch_two<128, 1, false>(ctx, a, zp, s) : launch_two<64, 1, false>(ctx, a, zp, s);
  FP_REQUIRE(false, "conv: unsupported configuration KW=%d Cin=%d", a.KW, a.Cin);
}
